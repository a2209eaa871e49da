"""Randomised end-to-end parity campaign ON THE GPU (not collected by pytest; run by hand on the GPU box):
    python tests/fuzz_gpu_query.py <seed0> <n_iter>
The draws of tests/fuzz_emu_query.py (world, drive, DB configuration, gate thresholds), but the product's own path end to
end -- points -> cc_ingest_batch -> cc_db_add_scans -> one batched cc_db_query_batch of EVERY scan at its own epoch -- against
the oracle's replay of the reference loop on the same points: descriptors (integers, contour rows, BCIs bit-exact; keys to
the f64 exp's last bits), every integer of every result record, correlation and pose within 1e-4.  Walk and tiled K3
alternate; every third drive uses full-size scans."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import cc_amd  # noqa: E402
import oracle_py as oracle  # noqa: E402
from parity import compare_desc  # noqa: E402

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy", "n_knn_hits"]


def one(cc, seed):
    import torch
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
    full = seed % 3 == 0
    x, poses, ts = cc.synth.make_sequence(n, world=world, device="cuda", step=(1.0 if kind < 2 else 3.0),
                                          **({} if full else dict(beams=16, azim=450)))
    P = x.shape[1]
    offs = np.arange(n + 1, dtype=np.int64) * P
    # the reference's drivers use a scan's sequence number as its id AND as the balance seed (batch_bin_test.cpp:131-237);
    # CandidateManager keys candidates by that id (contour_db.h:476), the C-ABI by DB index -- the same thing as long as ids
    # are unique, which the evaluator CHECKs: distinct values here (a duplicate merges two scans' candidates in the oracle)
    seeds = rng.choice(1 << 20, n, replace=False).astype(np.int32)
    os.environ["CC_KNN_MODE"] = "2" if seed % 2 else "0"
    # every seventh drive: another ContourManagerConfig (MulRan levels, fewer anchors / neighbours, a coarser grid)
    mcfg = None
    if seed % 7 == 3:
        mcfg = L.default_manager_cfg(mulran=bool(rng.integers(2)))
        v = int(rng.integers(3))
        if v == 0:
            mcfg.piv_firsts, mcfg.dist_firsts, mcfg.roi_radius = 4, 8, 8.0
        elif v == 1:
            mcfg.reso_row = mcfg.reso_col = 2.0
            mcfg.n_row = mcfg.n_col = 74
        else:
            mcfg.min_cont_cell_cnt, mcfg.min_cont_key_cnt = 4, 12
    ctx = cc.Context(0, mcfg, max_batch=512)
    desc = ctx.ingest(x.reshape(-1, 4), offs)
    db = cc.Database(ctx, cfg=d, capacity=n)
    online = seed % 4 == 1
    if not online:
        db.add_scans(desc, ts, seeds)
        res = db.query(desc, np.arange(n, dtype=np.int32), lb=lb, ub=ub, allow_flagged=True)
    else:
        # the online loop: sub-batch after sub-batch is added and queried at its own epochs with nothing collected in
        # between (cc_db_add_scans[_prepare] / cc_db_query_submit, 1-4 lanes): appends run next to query chunks in flight
        sub = int(rng.choice([1, 5, 16, 37])) if not long_drive else int(rng.choice([16, 37, 128]))
        db.set_lanes(int(rng.choice([1, 2, 4])))
        prep = bool(rng.integers(2))
        parts = []
        if prep:
            db.add_scans_prepare(desc[0:min(sub, n)].contiguous())
        for a0 in range(0, n, sub):
            a1 = min(a0 + sub, n)
            blk = desc[a0:a1].contiguous()
            db.add_scans(blk, ts[a0:a1], seeds[a0:a1])
            if prep and a1 < n:
                db.add_scans_prepare(desc[a1:min(a1 + sub, n)].contiguous())
            try:
                parts.append(db.query_submit(blk, np.arange(a0, a1, dtype=np.int32), lb=lb, ub=ub))
            except cc.CCError as e:
                if e.rc != cc.CC_ECAPACITY:
                    raise
        try:
            db.query_wait()
        except cc.CCError as e:
            if e.rc != cc.CC_ECAPACITY:
                raise
        res = np.concatenate(parts)
    torch.cuda.synchronize()
    dn = cc.desc_to_numpy(desc)
    ores, _, odesc = oracle.run_sequence(x.cpu().numpy().reshape(-1, 4), offs, ts, seeds, mcfg=mcfg, dcfg=d, lb=lb, ub=ub, want_desc=True)
    bad = 0
    for i in range(n):
        if dn["flags"][i] or res["flags"][i]:
            continue  # a capacity was met and reported: not a parity case
        b = compare_desc(odesc[i], dn[i], float_exact=False)
        if b:
            print("  MISMATCH seed %d scan %d descriptor: %s" % (seed, i, b[:3]))
            bad += 1
        for f in INT_FIELDS:
            if ores[f][i] != res[f][i]:
                print("  MISMATCH seed %d scan %d field %s: oracle %s kernels %s" % (seed, i, f, ores[f][i], res[f][i]))
                bad += 1
        if ores["n_res"][i] and res["n_res"][i]:
            e = max(abs(ores["correlation"][i] - res["correlation"][i]), float(np.abs(ores["tf"][i] - res["tf"][i]).max()))
            if e > 1e-4:
                print("  MISMATCH seed %d scan %d float error %.3g" % (seed, i, e))
                bad += 1
    print("seed %d kind %d %s %s n %d nnk %d qlv %s hits %d flagged %d knn-mode %s: %s" % (
        seed, kind, "full" if full else "16x450", "online" if online else "batch", n, d.nnk, qlv, int((ores["n_res"] > 0).sum()), int((dn["flags"] != 0).sum() + (res["flags"] != 0).sum()),
        os.environ["CC_KNN_MODE"], "ok" if not bad else "%d MISMATCHES" % bad), flush=True)
    if seed % 5 == 2:
        # tiled against walk K3 on a DB made of R copies of the drive (thousands of scans, every key R times over: exact
        # distance ties in crowds, groups of sixteen searches, long windows): hit lists must be identical, entry for entry
        R = int(rng.choice([16, 40, 90]))
        big = desc.repeat(R, 1).contiguous()
        nb = big.shape[0]
        tsb = np.arange(nb, dtype=np.float64) / 10.0
        sb = rng.choice(1 << 22, nb, replace=False).astype(np.int32)
        ep = np.full(n, nb, np.int32)
        ep[::3] = nb // 2
        out = []
        for mode in ("0", "2"):
            os.environ["CC_KNN_MODE"] = mode
            dbb = cc.Database(ctx, cfg=d, capacity=nb)
            dbb.add_scans(big, tsb, sb)
            out.append(dbb.query(desc, ep, lb=lb, ub=ub, want_knn=True, allow_flagged=True))
            dbb.close()
        (r1, k1, c1), (r2, k2, c2) = out
        m = np.arange(k1.shape[-1])[None, None, None, :] < c1[..., None]
        same = np.array_equal(c1, c2) and all(np.array_equal(k1[f][m], k2[f][m]) for f in ("gidx", "level", "seq", "dist_sq"))
        same = same and all(np.array_equal(r1[f], r2[f]) for f in INT_FIELDS)
        if not same:
            print("  MISMATCH seed %d: tiled and walk K3 differ on the %d-scan DB of %d copies" % (seed, nb, R))
            bad += 1
    if bad and os.environ.get("CC_FUZZ_DUMP"):  # what a CPU-harness replay of the query side needs (tests/fuzz_emu_query.py --replay)
        os.makedirs(os.environ["CC_FUZZ_DUMP"], exist_ok=True)
        np.savez_compressed(os.path.join(os.environ["CC_FUZZ_DUMP"], "seed%d.npz" % seed), odesc=np.frombuffer(odesc.tobytes(), np.uint8),
                            gdesc=np.frombuffer(dn.tobytes(), np.uint8), ts=ts, seeds=seeds, ores=np.frombuffer(ores.tobytes(), np.uint8),
                            gres=np.frombuffer(res.tobytes(), np.uint8),
                            cfg=np.array([d.min_elapse, d.max_elapse, d.nnk, d.max_fine_opt, d.n_q_levels] + [d.q_levels[i] for i in range(3)], np.float64),
                            lb=np.array([lb.i_ovlp_sum, lb.i_ovlp_max_one, lb.i_in_ang_rng, lb.i_indiv_sim, lb.i_orie_sim, lb.correlation, lb.area_perc,
                                         lb.neg_est_dist], np.float64))
    db.close()
    ctx.close()
    return bad


if __name__ == "__main__":
    s0, it = int(sys.argv[1]), int(sys.argv[2])
    cc = cc_amd.load()
    tot = 0
    for s in range(s0, s0 + it):
        tot += one(cc, s)
    print("done: %d mismatches" % tot)
