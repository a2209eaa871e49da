#!/usr/bin/env python3
"""Headline benchmark: scans/s of ingest + query (120k-point scans against a 5k-scan DB).

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W
  * one "step" = one pass of the hot path over one batch of synthetic query scans that are already
    resident in HBM: cc_ingest_batch (BEV rasterise -> contours -> keys/BCI) + cc_db_query_batch
    (KNN preselect -> constellation checks -> GMM-L2 + L-BFGS) against a prebuilt 5 000-scan DB;
  * N > 1: launched by torch.distributed.run, one rank per GPU.  The DB build is scan-sharded
    (each rank ingests n_db/N scans) followed by ONE all-gather of the descriptors over RCCL (the path's only
    exchange: every replica needs every DB scan); in the timed step every rank ingests + queries its own
    batch against its replica (weak scaling, no data-path collective: queries never need another rank's scans).
    `--share-descriptors` additionally all-gathers each batch's raw descriptor blocks (169 KB/scan), which an
    online deployment that appends the queried scans to every replica would do;
  * rank 0 prints ONE JSON line.  `value` is whole-job scans/s.
Extra objects: `roofline` (dominant kernel, HIP-event timed inside the library) and `cpu_baseline`
(the CPU restatement of the reference under oracle/, single thread, bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks of one node); must precede torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--db-scans", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=1024, help="query scans per step per GPU")
    ap.add_argument("--cpu-sample", type=int, default=256, help="query scans timed by the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--share-descriptors", action="store_true",
                    help="N > 1: all-gather every batch's descriptors in the timed step (what appending them to all replicas needs)")
    ap.add_argument("--no-overlap", action="store_true", help="everything on one stream: ingest, then the query chunks one by one")
    ap.add_argument("--stats", action="store_true", help="print the per-query check funnel of the last step to stderr")
    ap.add_argument("--workload", choices=("sparse", "dense"), default="sparse",
                    help="sparse: SURVEY.md 8(d)'s world (1 object / 150 m2), the headline configuration; dense: the cluttered "
                         "world (vegetation, walls, relief, HDL-64E beam table) with several times the contours per level")
    ap.add_argument("--tune-sweep", default="",
                    help="tuning aid (library built with -DCC_TUNE): 'VAR=v1,v2;VAR2=...': after the timed run, rebuild the DB "
                         "handle under each setting and print the isolated per-kernel ms of two steps to stderr")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start N ranks of this same command under torch.distributed.run
    # (one process per GPU, RCCL over xGMI) and let rank 0's JSON line through.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import cc_amd
    cc = cc_amd.load()
    L = cc.L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("CC_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # n_gpus in the JSON line = the ranks the process group really has
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n_db, B, K, W = args.db_scans, args.batch, args.steps, args.warmup
    P = 64 * 1875
    wld = cc.synth.World(dense=(args.workload == "dense"))
    ctx = cc.Context(local_rank, max_batch=max(B, 256))

    # ---------------- DB build (untimed): scan-sharded ingest + one all-gather of descriptors ----------------
    t_setup = time.time()
    shard = (n_db + world - 1) // world
    lo, hi = min(rank * shard, n_db), min((rank + 1) * shard, n_db)
    desc_local = torch.empty((shard, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    CH = 128
    for c0 in range(lo, hi, CH):
        c1 = min(c0 + CH, hi)
        xyzi, _, _ = cc.synth.make_sequence(c1 - c0, world=wld, device=dev, start=c0)
        ctx.ingest(xyzi.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P, out=desc_local[c0 - lo:c1 - lo])
    torch.cuda.synchronize()
    if world > 1:
        desc_all = torch.empty((world * shard, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(desc_all, desc_local)
        desc_db = desc_all[:n_db]
    else:
        desc_db = desc_local[:n_db]
    db = cc.Database(ctx, capacity=n_db + 16)
    if args.no_overlap:
        db.set_lanes(1)
    ts_db = np.arange(n_db, dtype=np.float64) / 10.0
    db.add_scans(desc_db.contiguous(), ts_db, np.arange(n_db, dtype=np.int32))
    if not args.tune_sweep:
        del desc_db
    # ---------------- query batches (resident in HBM before the timed region) ----------------
    n_steps_total = W + K
    batches = []
    for s in range(n_steps_total):
        start = n_db + (s * world + rank) * B
        xyzi, _, _ = cc.synth.make_sequence(B, world=wld, device=dev, start=start)
        batches.append(xyzi.reshape(-1, 4).contiguous())
    offs = np.arange(B + 1, dtype=np.int64) * P
    epochs = np.full(B, n_db, np.int32)
    qdesc = torch.empty((B, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    share = world > 1 and args.share_descriptors
    gathered = torch.empty((world * B, cc.DESC_BYTES), dtype=torch.uint8, device=dev) if share else None
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    # Two HIP streams: while the query chain of batch s runs on the main stream, the ingest kernels of batch s+1 run on
    # a second one (double-buffered descriptors).  Every batch is ingested AND queried inside the timed region.
    qdesc2 = [qdesc, torch.empty_like(qdesc)]
    s_ing = torch.cuda.Stream(device=dev)
    s_main = torch.cuda.current_stream(dev)

    def ingest_async(x, slot):
        s_ing.wait_stream(s_main)  # the slot's previous query has been issued on the main stream
        with torch.cuda.stream(s_ing):
            ctx.ingest(x, offs, out=qdesc2[slot])
            ev = torch.cuda.Event()
            ev.record(s_ing)
        return ev

    def run_steps(first, count):
        """ingest + query of batches[first : first+count], software-pipelined; returns #loop closures found."""
        found = 0
        ev = ingest_async(batches[first], 0) if not args.no_overlap else None
        for k in range(count):
            slot = k & 1
            if args.no_overlap:
                ctx.ingest(batches[first + k], offs, out=qdesc2[slot])
            else:
                s_main.wait_event(ev)
                if k + 1 < count:
                    ev = ingest_async(batches[first + k + 1], slot ^ 1)
            q = qdesc2[slot]
            if share:
                dist.all_gather_into_tensor(gathered, q)
            res = db.query(q, epochs)
            found += int((res["n_res"] > 0).sum())
            run_steps.last = res
        return found

    run_steps(0, W)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    cc.lib().cc_profile_enable(ctx.h, 1)
    cc.lib().cc_db_profile_enable(db.h, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_found = run_steps(W, K)
    res = run_steps.last
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.stats and rank == 0:
        for k in ("n_knn_hits", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy"):
            print("funnel %-16s mean %8.1f  max %6d" % (k, float(res[k].mean()), int(res[k].max())), file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    import ctypes as C

    def read_kernel_ms():
        ms2 = (C.c_double * 2)()
        nl = C.c_int()
        cc.lib().cc_profile_read(ctx.h, ms2, C.byref(nl))
        ms5 = (C.c_double * 5)()
        nl2 = C.c_int()
        cc.lib().cc_db_profile_read(db.h, ms5, C.byref(nl2))
        a, b = max(nl.value, 1), max(nl2.value, 1) / float(B)   # ingest launches (one per step) | queries -> steps
        return {"cc_k_rasterize": ms2[0] / a, "cc_k_contours": ms2[1] / a, "cc_k_knn": ms5[0] / b, "cc_k_check": ms5[1] / b,
                "cc_k_merge": ms5[2] / b, "cc_k_gmm": ms5[3] / b, "cc_k_final": ms5[4] / b}

    # HIP events over the timed region: per step, the summed durations of the kernel's launches (one per
    # chunk of <= 512 queries on the query side).  Launches of different streams overlap each other, so these are durations of
    # kernels SHARING the GPU, and a chunk pair's durations add up although they ran side by side.
    kms = read_kernel_ms()
    kms_iso = None
    if not args.no_overlap:         # the same kernels strictly one after the other (2 extra, untimed steps): isolated durations
        args.no_overlap = True
        db.set_lanes(1)
        run_steps(W, min(2, K))
        torch.cuda.synchronize()
        kms_iso = read_kernel_ms()

    if args.tune_sweep and rank == 0:  # tuning aid: isolated kernel times under each setting of a -DCC_TUNE build's knobs
        print("tune-sweep base: " + json.dumps({k: round(v, 4) for k, v in (kms_iso or kms).items()}), file=sys.stderr)
        for spec in args.tune_sweep.split(";"):
            var, vals = spec.split("=")
            for v in vals.split(","):
                os.environ[var] = v
                db2 = cc.Database(ctx, capacity=n_db + 16)
                db2.set_lanes(1)
                db2.add_scans(desc_db.contiguous(), ts_db, np.arange(n_db, dtype=np.int32))
                cc.lib().cc_db_profile_enable(db2.h, 1)
                db_saved, db = db, db2
                run_steps(W, min(2, K))
                torch.cuda.synchronize()
                ms5 = (C.c_double * 5)()
                nl2 = C.c_int()
                cc.lib().cc_db_profile_read(db2.h, ms5, C.byref(nl2))
                bq = max(nl2.value, 1) / float(B)
                print("tune-sweep %s=%s: knn %.4f check %.4f merge %.4f gmm %.4f final %.4f (ms per %d queries, isolated)"
                      % (var, v, ms5[0] / bq, ms5[1] / bq, ms5[2] / bq, ms5[3] / bq, ms5[4] / bq, B), file=sys.stderr)
                db = db_saved
                db2.close()
            os.environ.pop(var, None)

    if rank == 0:
        total_scans = K * B * world
        value = total_scans / elapsed
        # ---- roofline of the dominant kernel (HIP-event timed on its launch stream inside the timed region) ----
        d = cc.desc_to_numpy(qdesc2[(K - 1) & 1][:64])
        n_pix = float(d["n_pix"].mean())
        wl_stats = {"occupied_cells_mean": round(n_pix, 1), "contours_per_level_mean": [round(float(v), 1) for v in d["n_cont"].mean(0)],
                    "cells_above_level_mean": [round(float(v), 1) for v in d["layer_cell_cnt"].mean(0)],
                    "inexact_descriptors": int((d["flags"] & 6).astype(bool).sum())}
        # ALGORITHMIC bytes per launch (DESIGN.md "Kernels"): what the step has to move, independent of how.
        #  K1 streams the xyzi records once (16 B/point) and emits the dense BEV + per-cell continuous positions;
        #  K2 reads those and emits the descriptor used downstream;
        #  K3 reads the layers' key matrices once and, per anchor key, the 40 B key and <= nnk 12-B hits;
        #  K4 reads, per KNN hit, the hit and two contour records, per anchor-similar pair the two 256-bit rings, per
        #     check that reaches the pairing the two 600-B BCIs, and writes a 104-B record per pass;
        #  merge reads those records; K5 reads two ellipse tables (32 B/ellipse) per problem and writes 64 B.
        desc_emit = float(np.mean(72 + 16 + 1440 + 36 * 600 + d["n_stored"].sum(1) * 76))
        f = {k: float(res[k].mean()) for k in ("n_knn_hits", "cand_aft_check1", "cand_aft_check3", "n_cand_tidy")}
        n_keys_db = 3 * 6 * n_db
        alg = {"cc_k_rasterize": B * (P * 16 + 22500 * 4 + n_pix * 8),
               "cc_k_contours": B * (22500 * 4 + n_pix * 8 + desc_emit),
               "cc_k_knn": n_keys_db * 44 + B * 18 * 40 + B * f["n_knn_hits"] * 12,
               "cc_k_check": B * (f["n_knn_hits"] * (12 + 2 * 76) + f["cand_aft_check1"] * (64 + 2 * 600) + f["cand_aft_check3"] * 104),
               "cc_k_merge": B * f["cand_aft_check3"] * 104,
               "cc_k_gmm": B * f["n_cand_tidy"] * (2 * 45 * 32 + 64),
               "cc_k_final": B * 64}
        dom = max(kms, key=lambda k: kms[k])
        dom_ms, dom_bytes = kms[dom], alg[dom]
        k1_ms, k1_bytes = kms["cc_k_rasterize"], alg["cc_k_rasterize"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": "scans/sec ingest+query (120k-pt scan vs 5k-scan DB); max-F1 parity",
            "value": value, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic Velodyne-64 scans (64x1875=120000 pts), %s world, %d-scan DB, %d query scans/step/GPU, "
                                   "queries revisit DB places (loop closures found: %d of %d on rank 0); a step = ingest + query of "
                                   "the batch, the DB update (addScan/pushAndBalance) is outside the timed step"
                                   % (args.workload, n_db, B, n_found, K * B),
                       "world": args.workload, "workload_stats": wl_stats,
                       "shape_limits": "6 levels, grid <= 150x150, nnk <= 64, dist_firsts <= 10, <= 320 contours/level (flagged otherwise)",
                       "db_scans": n_db, "batch": B, "points_per_scan": P, "parallelism": "scan-sharded x%d%s" % (world, ", batch descriptors all-gathered" if share else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, B)[0], "traffic_source": pmc_traffic(dom, B)[1],
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "kernels_ms_per_launch": kms, "kernels_ms_per_launch_isolated": kms_iso,
                         "streams": 1 if kms_iso is None else 3,
                         "rasterize_GBs": k1_bytes / ((kms_iso or kms)["cc_k_rasterize"] * 1e-3) / 1e9 if k1_ms > 0 else None},
            "setup_s": setup_s,
        }
        if world == 1 and not args.no_cpu and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cc, wld, n_db, batches[W], P, min(args.cpu_sample, B))
        print(json.dumps(out), flush=True)
    db.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def pmc_traffic(kernel, batch):
    """HBM bytes per STEP of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*_pmc_summary.json: FETCH_SIZE/WRITE_SIZE per launch, gfx950 x2 correction applied where it is calibrated).
    Query kernels are launched once per chunk (two chunks per step up to 1024 scans, 512-query chunks above)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    ks = d.get("kernels", {})
    parts = ["cc_k_check_a", "cc_k_check_b", "cc_k_check_c"] if kernel == "cc_k_check" else [kernel]
    if d.get("batch_scans") != batch or any(p not in ks or "hbm_bytes_per_launch" not in ks[p] for p in parts):
        return None, None
    per_step = 1 if kernel in ("cc_k_rasterize", "cc_k_contours") else max(2, (batch + 511) // 512)
    return sum(ks[p]["hbm_bytes_per_launch"] for p in parts) * per_step, os.path.relpath(files[-1], ROOT)


def cpu_baseline(cc, wld, n_db, batch0, P, n_q, max_db_seconds=150.0):
    """The reference's single-threaded code path on the CPU restatement (oracle/, kd-tree = the reference's vendored
    nanoflann when oracle/_ref is built) on the SAME workload: the same n_db-scan DB is built first (untimed:
    addScan + pushAndBalance per scan), then the first n_q scans of the first timed batch are ingested and queried
    (timed: ContourManager ctor + makeBEV + makeContoursRecurs + queryRangedKNN per scan, no DB update -- like a GPU step)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    kd = O.use_ref_kdtree(True)
    odb = O.DB()
    t_build = time.perf_counter()
    t_ingest_db = 0.0
    for c0 in range(0, n_db, 128):
        c1 = min(c0 + 128, n_db)
        x, _, _ = cc.synth.make_sequence(c1 - c0, world=wld, device="cuda", start=c0)
        xh = x.cpu().numpy()
        for i in range(c1 - c0):
            t0 = time.perf_counter()
            s = O.Scan(xh[i], int_id=c0 + i, keep_cells=False)
            t_ingest_db += time.perf_counter() - t0
            s.clear_image()
            odb.add_scan(s, (c0 + i) / 10.0)
            odb.push_and_balance(c0 + i, (c0 + i) / 10.0)
    t_build = time.perf_counter() - t_build
    xq = batch0[:n_q * P].cpu().numpy().reshape(n_q, P, 4)
    found = 0
    t_ing = t_qry = 0.0
    t0 = time.perf_counter()
    for i in range(n_q):
        ta = time.perf_counter()
        s = O.Scan(xq[i], int_id=n_db + i, keep_cells=False)
        s.clear_image()
        tb = time.perf_counter()
        r = odb.query(s)
        tc = time.perf_counter()
        t_ing += tb - ta
        t_qry += tc - tb
        found += int(r["n_res"] > 0)
    dt = time.perf_counter() - t0
    return {"value": n_q / dt, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": "first %d scans of the first timed batch: ingest + query against the same %d-scan DB "
                      "(DB build untimed, %.1f s incl. synthesis; kd-tree=%s); %d loop closures found"
                      % (n_q, n_db, t_build, "reference nanoflann (oracle/_ref)" if kd else "exact scan", found),
            "seconds_per_scan": {"ingest (make bev)": t_ing / n_q, "query (KNN+Constell+L2 opt)": t_qry / n_q,
                                 "ingest while building the DB": t_ingest_db / n_db},
            "host_cpus": os.cpu_count(), "host_cpu_model": _cpu_model(),
            # not a measurement: the reference is single-threaded; this is what host_cpus independent copies of it on
            # disjoint scans could reach at best (perfect scaling, no memory-bandwidth loss)
            "ideal_all_cores_upper_bound": n_q / dt * (os.cpu_count() or 1)}


if __name__ == "__main__":
    main()
