"""The hint-driven CandidateManager flow (cc_db_check_hints; the reference's single-pair demo,
test/kitti_read_bin_test.cpp:226-291) on the CPU harness vs the oracle: per-hint gate scores, the candidate chosen and
its pose.  Hints are issued in the demo's order (level -> candidate key -> query key) and in a shuffled order, because
addProposal's greedy merge depends on it."""
import numpy as np

import emu_api

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy",
              "n_knn_hits"]


def _demo_hints(L, desc, q, cands, levels=(1, 2, 3, 4)):
    out = []
    for ci, c in enumerate(cands):
        for ll in levels:
            k1, k2 = desc["keys"][c][ll], desc["keys"][q][ll]
            for i1 in range(L.NPIV):
                for i2 in range(L.NPIV):
                    if k1[i1].sum() == 0 or k2[i2].sum() == 0:
                        continue
                    if float(((k1[i1] - k2[i2]) ** 2).sum()) > 1000.0:
                        continue
                    out.append((ci, ll, i1, i2))
    return np.array(out, np.int32).reshape(-1, 4)


def _run(cc, oracle, order_seed):
    L = oracle.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    P = x.shape[1]
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * P
    seeds = np.arange(n, dtype=np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=dcfg, want_desc=True)
    hit = np.nonzero(ores["n_res"] > 0)[0]
    assert len(hit) >= 2
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, odesc, ts, seeds)
    n_pass = 0
    for qi in hit[:2]:
        c = int(ores["cand_gidx"][qi])
        cands = [c, max(c - 1, 0), c + 1, 3]
        hints = _demo_hints(L, odesc, qi, cands)
        assert len(hints) > 30
        if order_seed is not None:
            hints = hints[np.random.default_rng(order_seed).permutation(len(hints))]
        oscans = [oracle.Scan.from_desc(odesc[g], int_id=int(g)) for g in cands]
        otgt = oracle.Scan.from_desc(odesc[qi], int_id=int(qi))
        for mfo in (5, 1):
            eres, esc = oracle.check_hints(otgt, oscans, hints, sim=dcfg.cont_sim, max_fine_opt=mfo)
            h = np.zeros(len(hints), L.hint_dt)
            h["cand_gidx"] = np.array(cands)[hints[:, 0]]
            h["level"], h["seq_src"], h["seq_tgt"] = hints[:, 1], hints[:, 2], hints[:, 3]
            res, sc = api.check_hints(db, odesc[qi:qi + 1], h, max_fine_opt=mfo)
            got = np.stack([sc[f] for f in ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim", "passed")], 1)
            bad = np.nonzero((got != esc).any(1))[0]
            assert len(bad) == 0, (qi, bad[:5], got[bad[:5]], esc[bad[:5]])
            for f in INT_FIELDS:
                exp = eres[f] if f != "cand_gidx" or eres["n_res"] == 0 else cands[int(eres[f])]
                assert exp == res[f], (qi, f, exp, res[f])
            if eres["n_res"]:
                assert abs(eres["correlation"] - res["correlation"]) < 1e-6
                assert np.abs(eres["tf"] - res["tf"]).max() < 1e-6
            n_pass += int(got[:, 5].sum())
    assert n_pass > 0, "no hint passed all four checks: the test would not exercise the merge"


def test_hints_demo_order(cc, oracle):
    _run(cc, oracle, None)


def test_hints_shuffled_order(cc, oracle):
    _run(cc, oracle, 7)


def test_hint_validation(cc, oracle):
    L = oracle.L
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    db = api.db_create(ctx, cap=4)
    q = np.zeros(1, L.scan_desc_dt)
    h = np.zeros(1, L.hint_dt)
    h["level"] = 1
    try:
        api.check_hints(db, q, h)  # no such candidate scan
    except RuntimeError as e:
        assert "not in the DB" in str(e)
    else:
        raise AssertionError("expected CC_EINVAL")


def test_hints_dense_world_long_pair_lists(cc, oracle):
    """The cluttered world (tens of contours per level): a candidate's correlation then selects hundreds of ellipse pairs, so
    the pair list of cc_k_gmm_init is flushed while it is being built and the refinement runs in its 64-lane instance
    (more than CC_GMM_G16_MAX_PAIRS = 96 pairs).  Descriptors come from the oracle; the candidates are the scans taken a
    moment before the query, the flow is the hint-driven one."""
    L = oracle.L
    dcfg = L.default_db_cfg()
    w = cc.synth.World(dense=True)
    n = 5
    x, poses, ts = cc.synth.make_sequence(n, world=w)
    P = x.shape[1]
    odesc = oracle.ingest_batch(x.numpy().reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P)
    assert odesc["n_cont"][:, 1:3].mean() > 25, "the dense world should give tens of contours on the low levels"
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, odesc, ts, np.arange(n, dtype=np.int32))
    qi, cands = 4, [3, 1]
    hints = _demo_hints(L, odesc, qi, cands)
    oscans = [oracle.Scan.from_desc(odesc[g], int_id=int(g)) for g in cands]
    otgt = oracle.Scan.from_desc(odesc[qi], int_id=int(qi))
    eres, esc = oracle.check_hints(otgt, oscans, hints, sim=dcfg.cont_sim, max_fine_opt=5)
    h = np.zeros(len(hints), L.hint_dt)
    h["cand_gidx"] = np.array(cands)[hints[:, 0]]
    h["level"], h["seq_src"], h["seq_tgt"] = hints[:, 1], hints[:, 2], hints[:, 3]
    res, sc = api.check_hints(db, odesc[qi:qi + 1], h, max_fine_opt=5)
    got = np.stack([sc[f] for f in ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim", "passed")], 1)
    assert np.array_equal(got, esc) and got[:, 5].sum() > 0
    assert eres["n_res"] == res["n_res"] == 1
    assert cands[int(eres["cand_gidx"])] == res["cand_gidx"]
    assert abs(eres["correlation"] - res["correlation"]) < 1e-6 and np.abs(eres["tf"] - res["tf"]).max() < 1e-6
