"""Randomised QUERY parity campaign on the CPU harness (not collected by pytest; run by hand):
    python tests/fuzz_emu_query.py <seed0> <n_iter>
Every iteration draws a world, a short revisiting drive (reduced scans: 16 beams x 450 steps), a DB configuration and gate
thresholds, replays the reference loop with the oracle and queries EVERY scan of the drive through the emulated kernels at
its own epoch (walk and tiled K3 alternate); every integer of the result record must be equal, correlation and pose
within 1e-6."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import cc_amd  # noqa: E402
import emu_api  # noqa: E402
import oracle_py as oracle  # noqa: E402

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy", "n_knn_hits"]


def one(seed):
    cc = cc_amd.load()
    L = oracle.L
    rng = np.random.default_rng(seed)
    d = L.default_db_cfg()
    long_drive = bool(os.environ.get("CC_FUZZ_LONG"))   # 300-500 scans with the shipped 15 s / 25 s delays: many re-balances
    if not long_drive:
        d.min_elapse = float(rng.uniform(0.8, 2.0))
        d.max_elapse = d.min_elapse + float(rng.uniform(0.5, 1.5))
    d.nnk = int(rng.choice([10, 30, 50, 64]))
    d.max_fine_opt = int(rng.choice([2, 5, 10]))
    qlv = [(1, 2, 3), (2, 3), (2, 3, 4), (1, 2, 3)][int(rng.integers(4))]
    d.n_q_levels = len(qlv)
    for i, v in enumerate(qlv):
        d.q_levels[i] = v
    lb, ub = L.default_thresholds()
    if rng.random() < 0.5:
        lb.i_ovlp_sum, lb.i_ovlp_max_one, lb.i_in_ang_rng, lb.i_indiv_sim, lb.i_orie_sim = [int(v) for v in rng.integers(2, 5, 5)]
        lb.correlation = float(rng.uniform(0.1, 0.5))
    kind = int(rng.integers(3))
    world = cc.synth.World(loop_len=float(rng.uniform(160, 220) if long_drive else rng.uniform(24, 36)), dense=(kind == 1), seed=int(rng.integers(1 << 20))) if kind < 2 else \
        cc.synth.World(kitti=True, seed=int(rng.integers(1 << 20)), block=float(rng.uniform(36, 50)), tile=300.0)
    n = int(rng.integers(300, 500)) if long_drive else int(rng.integers(56, 84))
    x, poses, ts = cc.synth.make_sequence(n, world=world, beams=16, azim=450, step=(1.0 if kind < 2 else 3.0))
    P = x.shape[1]
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * P
    # the reference's drivers use a scan's sequence number as its id AND as the balance seed (batch_bin_test.cpp:131-237);
    # CandidateManager keys candidates by that id (contour_db.h:476), the C-ABI by DB index -- the same thing as long as ids
    # are unique, which the evaluator CHECKs: distinct values here (a duplicate merges two scans' candidates in the oracle)
    seeds = rng.choice(1 << 20, n, replace=False).astype(np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=d, lb=lb, ub=ub, want_desc=True)
    os.environ["CC_KNN_MODE"] = "2" if seed % 2 else "0"
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, d, cap=n)
    api.db_add(db, odesc, ts, seeds)
    qs = np.arange(n, dtype=np.int32)
    bad = 0
    for c0 in range(0, n, 16):
        q = qs[c0:c0 + 16]
        res = api.db_query(db, odesc[q], q, lb=lb, ub=ub)
        for k, qi in enumerate(q):
            for f in INT_FIELDS:
                if ores[f][qi] != res[f][k]:
                    print("  MISMATCH seed %d scan %d field %s: oracle %s kernels %s" % (seed, qi, f, ores[f][qi], res[f][k]))
                    bad += 1
            if ores["n_res"][qi] and res["n_res"][k]:
                e = max(abs(ores["correlation"][qi] - res["correlation"][k]), float(np.abs(ores["tf"][qi] - res["tf"][k]).max()))
                if e > 1e-6:
                    print("  MISMATCH seed %d scan %d float error %.3g" % (seed, qi, e))
                    bad += 1
    print("seed %d kind %d n %d nnk %d qlv %s hits %d knn-mode %s: %s" % (seed, kind, n, d.nnk, qlv, int((ores["n_res"] > 0).sum()),
                                                                     os.environ["CC_KNN_MODE"], "ok" if not bad else "%d MISMATCHES" % bad), flush=True)
    return bad


if __name__ == "__main__":
    s0, it = int(sys.argv[1]), int(sys.argv[2])
    tot = 0
    for s in range(s0, s0 + it):
        tot += one(s)
    print("done: %d mismatches" % tot)
