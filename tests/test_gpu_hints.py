"""GPU parity of the hint-driven CandidateManager flow (cc_db_check_hints) against the oracle, through the C-ABI: per-hint
gate scores (ScoreConstellSim / ScorePairwiseSim bit-exact), the candidate chosen, correlation and pose within 1e-4.
Full-size scans (120k points) on a loop; hints in the reference demo's order and shuffled."""
import numpy as np
import pytest

from test_emu_hints import INT_FIELDS, _demo_hints

pytestmark = pytest.mark.gpu


def test_check_hints_matches_oracle(cc, oracle):
    import torch
    L = cc.L
    w = cc.synth.World(loop_len=100.0)
    n = 130
    xyzi, poses, ts = cc.synth.make_sequence(n, world=w, device="cuda")
    P = xyzi.shape[1]
    offs = np.arange(n + 1, dtype=np.int64) * P
    seeds = np.arange(n, dtype=np.int32)
    ctx = cc.Context(0, max_batch=128)
    desc = ctx.ingest(xyzi.reshape(-1, 4), offs)
    db = cc.Database(ctx, capacity=n)
    db.add_scans(desc, ts, seeds)
    hdesc = np.frombuffer(desc.cpu().numpy().tobytes(), dtype=L.scan_desc_dt)
    rng = np.random.default_rng(3)
    n_pass = n_res = 0
    for qi in (105, 112, 120, 129):
        # the place was visited one lap (100 m = 100 scans) earlier: true candidate qi-100, neighbours, and a far scan
        cands = [qi - 100, qi - 101, qi - 99, (qi - 50) % n]
        otgt = oracle.Scan.from_desc(hdesc[qi], int_id=qi)
        oscans = [oracle.Scan.from_desc(hdesc[g], int_id=int(g)) for g in cands]
        base = _demo_hints(L, hdesc, qi, cands)
        assert len(base) > 50
        for hints, mfo in ((base, 5), (base[rng.permutation(len(base))], 10), (base[::-1], 1)):
            eres, esc = oracle.check_hints(otgt, oscans, hints, max_fine_opt=mfo)
            h = np.zeros(len(hints), L.hint_dt)
            h["cand_gidx"] = np.array(cands)[hints[:, 0]]
            h["level"], h["seq_src"], h["seq_tgt"] = hints[:, 1], hints[:, 2], hints[:, 3]
            res, sc = db.check_hints(desc[qi], h, max_fine_opt=mfo)
            got = np.stack([sc[f] for f in ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim", "passed")], 1)
            bad = np.nonzero((got != esc).any(1))[0]
            assert len(bad) == 0, (qi, bad[:5], got[bad[:5]], esc[bad[:5]])
            for f in INT_FIELDS:
                exp = eres[f] if f != "cand_gidx" or eres["n_res"] == 0 else cands[int(eres[f])]
                assert exp == res[f], (qi, f, exp, res[f])
            if eres["n_res"]:
                assert abs(eres["correlation"] - res["correlation"]) < 1e-4
                assert np.abs(eres["tf"] - res["tf"]).max() < 1e-4
            n_pass += int(got[:, 5].sum())
            n_res += int(eres["n_res"])
    assert n_pass > 20 and n_res >= 6
    # the batched query path is untouched by the hint flow: same results before and after
    r1 = db.query(desc[100:], seeds[100:])
    db.check_hints(desc[129], np.zeros(0, L.hint_dt))
    r2 = db.query(desc[100:], seeds[100:])
    assert r1.tobytes() == r2.tobytes()
    torch.cuda.synchronize()


def test_umeyama_against_svd(cc):
    """getTFFromConstell (2-D umeyama without scaling, contour_mng.h:1246-1277) as the device computes it -- a closed
    form for SO(2) with parallel sums -- against an independent rigid fit: numpy SVD (Kabsch) over the very contour
    centres of the constellation that the device reports for each passing check.  The oracle is not involved."""
    import torch
    L = cc.L
    w = cc.synth.World(loop_len=100.0)
    n = 130
    xyzi, poses, ts = cc.synth.make_sequence(n, world=w, device="cuda")
    P = xyzi.shape[1]
    ctx = cc.Context(0, max_batch=128)
    desc = ctx.ingest(xyzi.reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P)
    db = cc.Database(ctx, capacity=n)
    db.add_scans(desc, ts, np.arange(n, dtype=np.int32))
    hdesc = np.frombuffer(desc.cpu().numpy().tobytes(), dtype=L.scan_desc_dt)
    n_checked = 0
    worst = 0.0
    for qi in (105, 118, 129):
        cands = [qi - 100, qi - 101]
        base = _demo_hints(L, hdesc, qi, cands)
        h = np.zeros(len(base), L.hint_dt)
        h["cand_gidx"] = np.array(cands)[base[:, 0]]
        h["level"], h["seq_src"], h["seq_tgt"] = base[:, 1], base[:, 2], base[:, 3]
        db.check_hints(desc[qi], h, max_fine_opt=5)
        for p in db.debug_passes():
            src, tgt = hdesc[int(h["cand_gidx"][p["hint"]])], hdesc[qi]
            S, T = [], []
            for wd in range(7):
                m = int(p["pairs"][wd])
                while m:
                    b = wd * 64 + (m & -m).bit_length() - 1
                    m &= m - 1
                    l, s_, t_ = b // 100 + 1, (b % 100) // 10, b % 10
                    S.append(src["cont"][l][s_]["pos_mean"])
                    T.append(tgt["cont"][l][t_]["pos_mean"])
            S, T = np.asarray(S, np.float64), np.asarray(T, np.float64)
            assert len(S) == p["n_pairs"] >= 4
            ms, mt = S.mean(0), T.mean(0)
            U, _, Vt = np.linalg.svd((T - mt).T @ (S - ms))
            D = np.diag([1.0, np.sign(np.linalg.det(U @ Vt))])
            R = U @ D @ Vt
            t = mt - R @ ms
            th = np.arctan2(R[1, 0], R[0, 0])
            err = max(abs(t[0] - p["tf"][0]), abs(t[1] - p["tf"][1]), abs(np.angle(np.exp(1j * (th - p["tf"][2])))))
            worst = max(worst, err)
            n_checked += 1
    assert n_checked > 20
    assert worst < 1e-9, worst
    db.close()
    ctx.close()
