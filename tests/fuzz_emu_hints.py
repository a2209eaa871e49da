"""Randomised campaign of the hint-driven CandidateManager flow on the CPU harness (run by hand):
    python tests/fuzz_emu_hints.py <seed0> <n_iter>
cc_db_check_hints (the reference's single-pair flow, test/kitti_read_bin_test.cpp:226-291) against the oracle: a random
world and drive, random candidate sets (the true match, its neighbours, unrelated scans), the demo's hint list in a random
order and with random drops and duplicates, random max_fine_opt and contour-similarity settings; per-hint gate scores, the
candidate chosen, its correlation and pose."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import cc_amd  # noqa: E402
import emu_api  # noqa: E402
import oracle_py as oracle  # noqa: E402
from test_emu_hints import INT_FIELDS, _demo_hints  # noqa: E402


def one(cc, seed):
    L = oracle.L
    rng = np.random.default_rng(seed)
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    if rng.random() < 0.4:
        dcfg.cont_sim.ta_cell_cnt, dcfg.cont_sim.tp_cell_cnt = float(rng.uniform(3, 12)), float(rng.uniform(0.1, 0.4))
        dcfg.cont_sim.tp_eigval, dcfg.cont_sim.ta_h_bar = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.2, 0.8))
    kind = int(rng.integers(2))
    w = cc.synth.World(loop_len=float(rng.uniform(28, 44)), dense=(kind == 1), seed=int(rng.integers(1 << 20)))
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * x.shape[1]
    seeds = np.arange(n, dtype=np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=dcfg, want_desc=True)
    hit = np.nonzero(ores["n_res"] > 0)[0]
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, odesc, ts, seeds)
    bad = n_chk = n_pass = 0
    queries = list(hit[rng.permutation(len(hit))[:3]]) if len(hit) else []
    queries += [int(rng.integers(20, n))]
    for qi in queries:
        qi = int(qi)
        c = int(ores["cand_gidx"][qi]) if ores["n_res"][qi] else int(rng.integers(0, qi))
        cands = list(dict.fromkeys([c, max(c - 1, 0), min(c + 1, n - 1), int(rng.integers(0, n)), int(rng.integers(0, n))]))
        cands = [g for g in cands if g != qi][:int(rng.integers(1, 6))]
        hints = _demo_hints(L, odesc, qi, cands, levels=tuple(sorted(rng.choice([1, 2, 3, 4], int(rng.integers(1, 5)), replace=False).tolist())))
        if len(hints) == 0:
            continue
        hints = hints[rng.permutation(len(hints))]
        if rng.random() < 0.5:
            hints = hints[rng.random(len(hints)) < 0.7]
        if rng.random() < 0.3 and len(hints):
            hints = np.concatenate([hints, hints[rng.integers(0, len(hints), 5)]])   # repeated hints
        hints = hints[:L.HINT_MAX if hasattr(L, "HINT_MAX") else 4096]
        if len(hints) == 0:
            continue
        oscans = [oracle.Scan.from_desc(odesc[g], int_id=int(g)) for g in cands]
        otgt = oracle.Scan.from_desc(odesc[qi], int_id=qi)
        mfo = int(rng.choice([1, 2, 5, 10]))
        eres, esc = oracle.check_hints(otgt, oscans, hints, sim=dcfg.cont_sim, max_fine_opt=mfo)
        h = np.zeros(len(hints), L.hint_dt)
        h["cand_gidx"] = np.array(cands)[hints[:, 0]]
        h["level"], h["seq_src"], h["seq_tgt"] = hints[:, 1], hints[:, 2], hints[:, 3]
        res, sc = api.check_hints(db, odesc[qi:qi + 1], h, max_fine_opt=mfo)
        got = np.stack([sc[f] for f in ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim", "passed")], 1)
        n_chk += len(hints)
        n_pass += int(got[:, 5].sum())
        if (got != esc).any():
            k = np.nonzero((got != esc).any(1))[0][:3]
            print("  MISMATCH seed %d query %d: scores of hints %s: kernels %s oracle %s" % (seed, qi, k.tolist(), got[k].tolist(), esc[k].tolist()))
            bad += 1
        for f in INT_FIELDS:
            exp = eres[f] if f != "cand_gidx" or eres["n_res"] == 0 else cands[int(eres[f])]
            if exp != res[f]:
                print("  MISMATCH seed %d query %d field %s: oracle %s kernels %s" % (seed, qi, f, exp, res[f]))
                bad += 1
        if eres["n_res"] and res["n_res"]:
            e = max(abs(eres["correlation"] - res["correlation"]), float(np.abs(eres["tf"] - res["tf"]).max()))
            if e > 1e-6:
                print("  MISMATCH seed %d query %d float error %.3g" % (seed, qi, e))
                bad += 1
    print("seed %d kind %d queries %d hints %d passed %d: %s" % (seed, kind, len(queries), n_chk, n_pass, "ok" if not bad else "%d MISMATCHES" % bad), flush=True)
    return bad


if __name__ == "__main__":
    s0, it = int(sys.argv[1]), int(sys.argv[2])
    cc = cc_amd.load()
    tot = 0
    for s in range(s0, s0 + it):
        tot += one(cc, s)
    print("done: %d mismatches" % tot)
