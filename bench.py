#!/usr/bin/env python3
"""Headline benchmark: scans/s of ingest + query (120k-point scans against a 5k-scan DB).

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W
  * one "step" = one pass of the hot path over one batch of synthetic query scans that are already
    resident in HBM: cc_ingest_batch (BEV rasterise -> contours -> keys/BCI) + the batched query
    (KNN preselect -> constellation checks -> GMM-L2 + L-BFGS) against a prebuilt 5 000-scan DB.  The steps are
    streamed: cc_db_query_submit per step (the lanes are not drained between batches) and one cc_db_query_wait before
    the timed region closes, so every result is on the host inside it (--sync-query: cc_db_query_batch per step);
  * N > 1: launched by torch.distributed.run, one rank per GPU.  The DB build is scan-sharded
    (each rank ingests n_db/N scans and packs them into the 59 KB per-scan records the DB keeps) followed by ONE
    all-gather of those records over RCCL (the path's only exchange: every replica needs every DB scan); in the timed step every rank ingests + queries its own
    batch against its replica (weak scaling, no data-path collective: queries never need another rank's scans).
    `--share-descriptors` additionally all-gathers each batch's compact records (59 KB/scan), which an online
    deployment that appends the queried scans to every replica would do;
  * rank 0 prints ONE JSON line.  `value` is whole-job scans/s.
Extra objects: `roofline` (dominant kernel, HIP-event timed inside the library) and `cpu_baseline`
(the CPU restatement of the reference under oracle/, single thread, bounded sample, rank 0, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks of one node); must precede torch
# The step runs on four streams (caller, ingest, two query lanes).  HIP hands a process' hardware queues (4 by default) to
# streams in creation order, and two busy streams on one queue serialise; with eight queues no
# creation order -- e.g. RCCL's own streams in a multi-GPU run -- can make two of them share (measured equal at N = 1:
# profiles/r2g_bench_line_lanes2_hwq8.json).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


PROF_EVERY = 3  # the library's stage events ride on every 3rd chunk launch of the timed region (they cost throughput)
ISO_STEPS = 4   # extra untimed steps, everything on one stream, for the isolated per-kernel times
# what the path computes in, as the reference does: f32 for the BEV, moments, keys' sums, kNN distances and gates; f64 for
# gaussPDF's exp, umeyama, the GMM-L2 cost / gradient and the L-BFGS refinement
DTYPE = "f32+f64"
DTYPE_NOTE = "as the reference: f32 raster / moments / key sums / kNN distances / gates, f64 gaussPDF exp, umeyama, GMM-L2 cost and L-BFGS"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; --workload seq: the whole sequence)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 5; --workload seq: 2)")
    ap.add_argument("--db-scans", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=1024, help="query scans per step per GPU")
    ap.add_argument("--cpu-sample", type=int, default=256, help="query scans timed by the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--share-descriptors", action="store_true",
                    help="N > 1: all-gather every batch's descriptors in the timed step (what appending them to all replicas needs)")
    ap.add_argument("--no-overlap", action="store_true", help="everything on one stream: ingest, then the query chunks one by one")
    ap.add_argument("--stats", action="store_true", help="print the per-query check funnel of the last step to stderr")
    ap.add_argument("--ab-env", default="",
                    help="tuning aid: 'VAR=v[,VAR2=v2];VAR=w...': after the timed run, for each setting rebuild the DB with those "
                         "environment variables and time the same K steps the same way (scans/s to stderr)")
    ap.add_argument("--sync-query", action="store_true",
                    help="cc_db_query_batch per step (collects every batch before the next one is queued) instead of cc_db_query_submit + one cc_db_query_wait")
    ap.add_argument("--lanes", type=int, default=0, help="query chunks in flight inside cc_db_query_batch (1..4; 0 = library default 2)")
    ap.add_argument("--workload", choices=("sparse", "dense", "kitti", "seq"), default="kitti",
                    help="kitti (the default, the workload BASELINE.json's target is stated on): the KITTI-shaped town (street grid, "
                         "porous vegetation, rough ground: SURVEY.md 8(d)'s value distributions -- 4-6 k occupied cells, ~100 "
                         "contours on the low levels, 18 valid keys per scan) driven as a random walk, so that ~10 % of the query "
                         "scans revisit a DB place and the others end without a candidate; "
                         "sparse: a world of 1 object / 150 m2 (2 k occupied cells, ~14 contours per level: lighter than 8(d) asks "
                         "for; rounds 1-5 quoted their headline on it, now in `extra`); dense: the cluttered "
                         "world (vegetation, walls, relief, HDL-64E beam table) with several times the contours per level; "
                         "seq: BASELINE config 2's shape -- the reference's ONLINE loop (test/batch_bin_test.cpp:131-237) over one "
                         "long sequence of the dense world from an empty DB: per sub-batch ingest -> addScan/pushAndBalance -> query "
                         "(scan i against epoch i), the DB update INSIDE the timed region; a step = one sub-batch of --seq-batch scans")
    ap.add_argument("--seq-scans", type=int, default=4096, help="--workload seq: scans of the sequence (KITTI-08 has 4071)")
    ap.add_argument("--seq-batch", type=int, default=512, help="--workload seq: scans per ingest/add/query sub-batch (with four query "
                    "lanes, the online loop's default, two sub-batches are in flight while the host books the next append)")
    ap.add_argument("--seq-repeats", type=int, default=5, help="--workload seq: the timed pass with the DB update is run this many times "
                    "(fresh DB each time); `value` is the MEDIAN, min / median / max are in the line")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` measurements of the default run (online replay)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank ingests + queries --batch scans per step (per-GPU work fixed, the default); "
                         "strong = the --batch scans of a step are split over the ranks (total work fixed)")
    ap.add_argument("--beams", type=int, default=64, help="rays of the synthetic sensor: beams x azim (the metric is quoted on 64 x 1875 = 120 000 points)")
    ap.add_argument("--azim", type=int, default=1875)
    ap.add_argument("--comm-owner", choices=("torch", "c"), default=os.environ.get("CC_BENCH_COMM_OWNER", "torch"),
                    help="who issues the path's collective (the all-gather of the packed records): torch.distributed (nccl = RCCL; the "
                         "default, covered by the gloo twin on CPU) or the library's own cc_comm_allgather_packed (ncclAllGather "
                         "through the C-ABI, what a C++ host uses: hostcpp/examples/batch_replay_mgpu.cpp); the barrier and the "
                         "max-over-ranks timing stay with torch.distributed either way.  'c' also runs at N = 1 (a world of one)")
    ap.add_argument("--ingest-cus", default=os.environ.get("CC_BENCH_INGEST_CUS", ""),
                    help="CU partition of the overlapped step (DESIGN.md section 3): 'a/b' = the ingest stream is created with "
                         "hipExtStreamCreateWithCUMask on a of every b CUs of each XCD, 'xa/b' = on a of every b XCDs; the query "
                         "lanes keep the whole chip, so the other CUs' LDS is never taken by K1 / K2.  Empty: no mask")
    ap.add_argument("--tune-sweep", default="",
                    help="tuning aid (library built with -DCC_TUNE): 'VAR=v1,v2;VAR2=...': after the timed run, rebuild the DB "
                         "handle under each setting and print the isolated per-kernel ms of two steps to stderr")
    args = ap.parse_args()
    steps_given = args.steps is not None
    if args.steps is None:
        args.steps = 20   # (the round-end driver passes --steps 20 --warmup 5; a timed region of 8 steps is a quarter pipeline fill and drain)
    args.steps_given = steps_given
    if args.warmup is None:
        args.warmup = 2 if args.workload == "seq" else 5

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
    if os.environ.get("CC_BENCH_LAUNCH_PROBE"):  # launcher self-test (tests/test_distributed_gloo.py): join the group, report, leave
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("CC_BENCH_BACKEND", "nccl"))
        if dist.get_rank() == 0:
            print(json.dumps({"launch_probe_world": dist.get_world_size()}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    import cc_amd
    cc = cc_amd.load()
    if os.environ.get("CC_BENCH_LIB"):  # tuning aid: time another build of the library (an ablation, an older kernel set)
        cc.LIB_PATH = os.path.abspath(os.environ["CC_BENCH_LIB"])
        print("bench.py: library %s" % cc.LIB_PATH, file=sys.stderr)
    L = cc.L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # CC_BENCH_HARNESS=emu (tests only, no GPU): the same launcher, sharding, DB exchange, step loop, timing protocol and JSON
    # assembly with the C-ABI's CPU-harness build (tests/emu) underneath and gloo as the backend, so that the whole N > 1
    # path has been executed before a multi-GPU node exists.  Its numbers mean nothing and the line says so.
    harness = os.environ.get("CC_BENCH_HARNESS") == "emu"
    dist = None
    backend = os.environ.get("CC_BENCH_BACKEND", "gloo" if harness else "nccl")  # "nccl" IS RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if harness:
            dist.init_process_group(backend)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # n_gpus in the JSON line = the ranks the process group really has
    dev = torch.device("cpu") if harness else torch.device("cuda", local_rank)
    if not harness:
        torch.cuda.set_device(dev)

    def sync():
        if not harness:
            torch.cuda.synchronize()

    n_db, B, K, W = args.db_scans, args.batch, args.steps, args.warmup
    if args.scaling == "strong":
        if B % world:
            raise SystemExit("bench.py --scaling strong: --batch %d is not divisible by %d ranks" % (B, world))
        B = B // world    # a step's --batch scans are split over the ranks
    P = args.beams * args.azim
    if harness:
        args.no_overlap = args.no_cpu = args.no_extra = True
    wld = cc.synth.World(kitti=True) if args.workload == "kitti" else cc.synth.World(dense=(args.workload in ("dense", "seq")))
    if args.workload == "seq":
        return bench_seq(cc, args, dev, local_rank, world, rank, dist)
    ctx = _HarnessCtx(cc, max(B, 8)) if harness else cc.Context(local_rank, max_batch=max(B, 256))
    mk = dict(beams=args.beams, azim=args.azim)

    # ---------------- DB build (untimed): scan-sharded ingest, pack, ONE all-gather of the compact records ----------------
    t_setup = time.time()
    SH = cc.sharding
    shard = SH.shard_len(n_db, world)
    mine = SH.my_scans(n_db, rank, world)              # interleaved: rank r takes DB scans r, r + world, ...
    HB, FB = cc.packed_sizes()
    rec_local = torch.zeros((shard, HB + FB), dtype=torch.uint8, device=dev)  # per scan: hot record | correlation inputs
    desc_tmp = torch.empty((128, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    keep_desc = args.tune_sweep or (world == 1 and not args.no_cpu and args.cpu_sample > 0)
    desc_keep = torch.empty((n_db, cc.DESC_BYTES), dtype=torch.uint8, device="cpu", pin_memory=True) if keep_desc else None
    CH = 128
    for c0 in range(0, len(mine), CH):
        c1 = min(c0 + CH, len(mine))
        xyzi, _, _ = cc.synth.make_sequence(0, world=wld, device=dev, indices=mine[c0:c1], **mk)
        d = ctx.ingest(xyzi.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P, out=desc_tmp[:c1 - c0])
        hot, feat = ctx.pack(d)
        rec_local[c0:c1, :HB] = hot
        rec_local[c0:c1, HB:] = feat
        if keep_desc:
            desc_keep[c0:c1].copy_(d)   # world == 1: mine == all scans in order
    sync()
    if world > 1:
        dist.barrier()
    t_x = time.perf_counter()
    c_comm = None
    if args.comm_owner == "c" and not harness:
        c_comm, c_rank, c_world = cc.comm_from_env()
        assert (c_rank, c_world) == (rank, world), (c_rank, c_world, rank, world)
        rec_db, exchange_bytes = SH.gather_records_c(cc, c_comm, rec_local, n_db, world)   # ncclAllGather issued by the library
    else:
        rec_db, exchange_bytes = SH.gather_records(rec_local, n_db, world, dist)   # RCCL over xGMI: 59 KB per scan (the descriptor is 169 KB)
    sync()
    exchange_ms = (time.perf_counter() - t_x) * 1e3 if (world > 1 or c_comm is not None) else 0.0
    db = _HarnessDb(ctx, n_db + 16) if harness else cc.Database(ctx, capacity=n_db + 16)
    if args.no_overlap:
        db.set_lanes(1)
    elif args.lanes:
        db.set_lanes(args.lanes)
    ts_db = np.arange(n_db, dtype=np.float64) / 10.0
    hot_db, feat_db = rec_db[:, :HB].contiguous(), rec_db[:, HB:].contiguous()
    db.add_packed(hot_db, feat_db, ts_db, np.arange(n_db, dtype=np.int32))
    if not args.tune_sweep and not args.ab_env:
        del hot_db, feat_db
    del rec_db
    # ---------------- query batches (resident in HBM before the timed region) ----------------
    # every step gets its own batch of scans up to 48 distinct batches (95 GB at the default shape); longer runs go round
    # the ring -- the work per step is the same, only the inputs repeat
    n_steps_total = min(W + K, 48)
    batches = []
    for s in range(n_steps_total):
        start = n_db + (s * world + rank) * B   # weak: rank r's own batch; strong: rank r's slice of the step's global batch
        xyzi, _, _ = cc.synth.make_sequence(B, world=wld, device=dev, start=start, **mk)
        batches.append(xyzi.reshape(-1, 4).contiguous())
    offs = np.arange(B + 1, dtype=np.int64) * P
    epochs = np.full(B, n_db, np.int32)
    qdesc = torch.empty((B, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    share = world > 1 and args.share_descriptors
    gathered = torch.empty((world * B, HB + FB), dtype=torch.uint8, device=dev) if share else None
    rec_q = torch.empty((B, HB + FB), dtype=torch.uint8, device=dev) if share else None
    share_ev = []  # (start, end) events around the per-step all-gather
    sync()
    setup_s = time.time() - t_setup

    # Two HIP streams: while the query chain of batch s runs on the main stream, the ingest kernels of batch s+1 run on
    # a second one (double-buffered descriptors).  Every batch is ingested AND queried inside the timed region.
    qdesc2 = [qdesc, torch.empty_like(qdesc)]
    s_ing = None if harness else (masked_stream(torch, dev, args.ingest_cus) if args.ingest_cus else torch.cuda.Stream(device=dev))
    s_main = None if harness else torch.cuda.current_stream(dev)

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
        pending = []
        if count <= 0:
            return 0
        ev = ingest_async(batches[first % len(batches)], 0) if not args.no_overlap else None
        for k in range(count):
            slot = k & 1
            if args.no_overlap:
                ctx.ingest(batches[(first + k) % len(batches)], offs, out=qdesc2[slot])
            else:
                s_main.wait_event(ev)
                if k + 1 < count:
                    ev = ingest_async(batches[(first + k + 1) % len(batches)], slot ^ 1)
            q = qdesc2[slot]
            if share:  # what appending the batch to every replica needs: its compact records on every rank
                hq, fq = ctx.pack(q)
                rec_q[:, :HB] = hq
                rec_q[:, HB:] = fq
                if c_comm is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    cc.comm_allgather(c_comm, rec_q.data_ptr(), gathered.data_ptr(), int(rec_q.numel()), torch.cuda.current_stream(dev).cuda_stream)
                    e1.record()
                    share_ev.append((e0, e1))
                elif harness:
                    t_s = time.perf_counter()
                    dist.all_gather_into_tensor(gathered, rec_q)
                    share_ev.append((time.perf_counter() - t_s) * 1e3)
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    dist.all_gather_into_tensor(gathered, rec_q)
                    e1.record()
                    share_ev.append((e0, e1))
            t_h = time.perf_counter()
            if args.sync_query or args.no_overlap:
                res = db.query(q, epochs)
            else:  # queue the batch; its chunks are collected when their lanes are needed again, the last ones below
                res = db.query_submit(q, epochs)
            run_steps.host_submit_s += time.perf_counter() - t_h
            pending.append(res)
        t_h = time.perf_counter()
        db.query_wait()
        run_steps.host_wait_s += time.perf_counter() - t_h
        for res in pending:
            found += int((res["n_res"] > 0).sum())
        run_steps.last = pending[-1]
        return found

    run_steps.host_submit_s = run_steps.host_wait_s = 0.0
    run_steps(0, W)
    sync()
    if world > 1:
        dist.barrier()
    if not harness:
        cc.lib().cc_profile_enable(ctx.h, 1)
        cc.lib().cc_db_profile_enable(db.h, PROF_EVERY)  # stage events on every 3rd chunk launch (alternating lanes)
    share_ev.clear()
    sync()
    run_steps.host_submit_s = run_steps.host_wait_s = 0.0
    t0 = time.perf_counter()
    n_found = run_steps(W, K)
    res = run_steps.last
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("CC_BENCH_HOSTPROF") and rank == 0:  # tuning aid: where the host thread spent the timed region
        print("host: %.3f ms per step inside cc_db_query_submit (lane collection included), %.3f ms in the final cc_db_query_wait, "
              "%.3f ms per step elapsed" % (run_steps.host_submit_s / K * 1e3, run_steps.host_wait_s * 1e3, elapsed / K * 1e3), file=sys.stderr)
    if args.stats and rank == 0:
        for k in ("n_knn_hits", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy"):
            print("funnel %-16s mean %8.1f  max %6d" % (k, float(res[k].mean()), int(res[k].max())), file=sys.stderr)
    per_rank = [K * B / elapsed]
    if world > 1:
        tl = torch.zeros(world, dtype=torch.float64, device=dev)
        tl[rank] = elapsed
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)    # every rank's own time (tiny; after the timed region)
        per_rank = [K * B / float(v) for v in tl.tolist()]
        elapsed = float(tl.max().item())

    import ctypes as C

    def read_kernel_ms():
        if harness:
            return {k_: 0.0 for k_ in ("cc_k_rasterize", "cc_k_contours", "cc_k_knn", "cc_k_check", "cc_k_merge", "cc_k_gmm", "cc_k_final")}
        return _read_kernel_ms(cc, ctx, db, B)

    # HIP events over the timed region: per step, the summed durations of the kernel's launches (one per
    # chunk of <= 512 queries on the query side).  Launches of different streams overlap each other, so these are durations of
    # kernels SHARING the GPU, and a chunk pair's durations add up although they ran side by side.
    kms = read_kernel_ms()
    if args.ab_env and rank == 0:
        for spec in [""] + args.ab_env.split(";"):
            kv = [a.split("=") for a in spec.split(",") if a]
            for k_, v_ in kv:
                if k_ not in ("LANES", "PROF", "INGEST_CUS"):
                    os.environ[k_] = v_
            if "INGEST_CUS" in dict(kv):  # pseudo-variable: the ingest stream's CU mask ('-' = none)
                v_ = dict(kv)["INGEST_CUS"]
                s_ing = torch.cuda.Stream(device=dev) if v_ == "-" else masked_stream(torch, dev, v_)
            db2 = cc.Database(ctx, capacity=n_db + 16)
            lanes_ = int(dict(kv).get("LANES", args.lanes))  # pseudo-variable: cc_db_set_lanes
            if lanes_:
                db2.set_lanes(lanes_)
            db2.add_packed(hot_db, feat_db, ts_db, np.arange(n_db, dtype=np.int32))
            if "PROF" in dict(kv):  # pseudo-variable: stage events on every n-th chunk launch
                cc.lib().cc_db_profile_enable(db2.h, int(dict(kv)["PROF"]))
            db_saved, db = db, db2
            run_steps(0, W)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(W, K)
            torch.cuda.synchronize()
            print("ab-env [%s]: %.0f scans/s" % (spec, K * B / (time.perf_counter() - t1)), file=sys.stderr)
            if "PROF" in dict(kv):
                print("ab-env [%s] kernel ms per step: %s" % (spec, json.dumps({k: round(v, 3) for k, v in read_kernel_ms().items()})),
                      file=sys.stderr)
            db = db_saved
            db2.close()
            for k_, _ in kv:
                os.environ.pop(k_, None)
        read_kernel_ms()  # drop what these runs added to the profiling sums
    share_ms = None
    if share_ev:  # per-step time of the batch-record all-gather (--share-descriptors), this rank
        share_ms = float(np.mean([e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in share_ev]))
    kms_iso = None
    if not args.no_overlap:         # the same kernels strictly one after the other (2 extra, untimed steps): isolated durations
        args.no_overlap = True
        db.set_lanes(1)
        cc.lib().cc_db_profile_enable(db.h, 1)
        run_steps(W, min(ISO_STEPS, K))
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
                db2.add_packed(hot_db, feat_db, ts_db, np.arange(n_db, dtype=np.int32))
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
        # ---- roofline (kernel groups HIP-event timed on their launch streams; dominant group by isolated time) ----
        d = cc.desc_to_numpy(qdesc2[(K - 1) & 1][:64])
        n_pix = float(d["n_pix"].mean())
        wl_stats = {"occupied_cells_mean": round(n_pix, 1), "contours_per_level_mean": [round(float(v), 1) for v in d["n_cont"].mean(0)],
                    "cells_above_level_mean": [round(float(v), 1) for v in d["layer_cell_cnt"].mean(0)],
                    "valid_db_keys_per_scan_mean": round(float((np.abs(d["keys"].reshape(len(d), 6, 6, 10)[:, 1:4]).sum(-1) > 0).sum((1, 2)).mean()), 2),
                    "inexact_descriptors": int((d["flags"] & 6).astype(bool).sum()),
                    "queries_with_a_loop_closure": round(n_found / float(K * B), 4),
                    "checks_per_query_mean": round(float(res["cand_aft_check1"].mean()), 1),
                    "correlation_problems_per_query_mean": round(float(res["n_cand_tidy"].mean()), 2)}
        alg, split = algorithmic_bytes(d, res, B, P, n_db)
        roof = roofline_object(alg, split, kms, kms_iso, B, n_db, args.workload, elapsed / K * 1e3, P)
        roof.update({"streams": 1 if kms_iso is None else 3,
                     "event_sampling": "query-side stage events on every %d. chunk launch of the timed region, ingest events on every step; "
                                       "isolated: %d extra untimed steps, one stream, every launch" % (PROF_EVERY, ISO_STEPS),
                     "query_protocol": "cc_db_query_batch per step" if (args.sync_query or kms_iso is None) else "cc_db_query_submit per step, one cc_db_query_wait inside the timed region",
                     "value_over_ingest_roofline": value / world / (HBM_PEAK_GBS * 1e9 / (P * 16)),
                     # scans/s over what the HBM could stream if ingest did nothing but read the points once (8 TB/s / 1.92 MB = 4.17 M scans/s)
                     "ingest_frac": value / world / (HBM_PEAK_GBS * 1e9 / (P * 16))})
        out = {
            "metric": "scans/sec ingest+query (120k-pt scan vs 5k-scan DB); max-F1 parity",
            "value": value, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "synthetic Velodyne-64 scans (%dx%d=%d pts), %s world, %d-scan DB, %d query scans/step/GPU, "
                                   "%s (loop closures found: %d of %d on rank 0); inputs resident in HBM before the timed region; a step = ingest + query of "
                                   "the batch, the DB update (addScan/pushAndBalance) is outside the timed step"
                                   % (args.beams, args.azim, P, args.workload, n_db, B, "the drive goes on through the town: about a tenth of the query scans revisit a DB place"
                                      if args.workload == "kitti" else "queries revisit DB places", n_found, K * B),
                       "world": args.workload, "occupied_cells_mean": wl_stats["occupied_cells_mean"],
                       "contours_per_level_mean": wl_stats["contours_per_level_mean"], "workload_stats": wl_stats, "dtype_note": DTYPE_NOTE,
                       "shape_limits": "6 levels, grid <= 150x150, nnk <= 64, dist_firsts <= 10; the 320 largest contours of a level are stored, a scan with more components on a level goes through the exact slow path (cc_k_contours_big)",
                       "db_scans": n_db, "batch": B, "global_batch": B * world, "points_per_scan": P,
                       "parallelism": "scan-sharded x%d%s" % (world, ", batch descriptors all-gathered" if share else "")},
            "roofline": roof,
            "setup_s": setup_s,
        }
        if harness:
            out["harness"] = "CPU harness build of the C-ABI (tests/emu), backend %s: a test of the launcher / sharding / JSON path, NOT a measurement" % backend
        if world > 1:
            # what the multi-GPU run did, as seen by the process group itself
            HBq, FBq = cc.packed_sizes() if not harness else ctx.api.packed_sizes()
            out["multi_gpu"] = {
                "ranks_seen": dist.get_world_size(), "backend": backend, "scaling": args.scaling,
                "collective_issued_by": "cc_comm_allgather_packed (the library: ncclAllGather)" if c_comm is not None else "torch.distributed.all_gather_into_tensor",
                "per_rank_scans_per_s": per_rank,
                "db_exchange": {"collective": "all_gather_into_tensor of the packed per-scan records (hot record + correlation inputs)",
                                "bytes_per_scan": HBq + FBq, "bytes_gathered_per_rank": exchange_bytes, "ms": exchange_ms,
                                "achieved_GBs": exchange_bytes * (world - 1) / world / (exchange_ms * 1e-3) / 1e9 if exchange_ms > 0 else None,
                                "note": "bytes each rank receives from its peers / wall time of the collective (includes its launch and the first-use set-up of the communicator)"},
                "per_step_exchange": ({"what": "--share-descriptors: all-gather of the step's packed records", "bytes_gathered_per_rank": int(gathered.numel()),
                                       "ms": share_ms,
                                       "achieved_GBs": gathered.numel() * (world - 1) / world / (share_ms * 1e-3) / 1e9 if share_ms else None}
                                      if share else None),
                "data_path_collectives_in_timed_step": 1 if share else 0}
        if world == 1 and c_comm is not None:  # --comm-owner c on one GPU: the library's RCCL call with a world of one
            out["comm_owner_c_world1"] = {"collective": "cc_comm_allgather_packed (ncclAllGather, world = 1)", "bytes": exchange_bytes, "ms": exchange_ms}
        # the CPU leg's sample: scans spread evenly over the TIMED batches (every (K * B / n)-th scan of the timed drive), so that it
        # sees the drive's mix of first passes and revisits -- the first scans of the first timed batch can be all of one kind
        # (with --warmup 5 every one of them revisits a DB place: 12 ms of L2 optimisation per scan, 64 scans/s instead of ~270)
        n_cpu = min(args.cpu_sample, B)
        if not args.no_cpu and n_cpu > 0 and world == 1:
            t_idx = [int(j) * (K * B) // n_cpu for j in range(n_cpu)]
            batch_cpu = torch.cat([batches[(W + t // B) % len(batches)][(t % B) * P:((t % B) + 1) * P] for t in t_idx])
        else:
            batch_cpu = batches[W % len(batches)]
        if world == 1 and not args.no_extra:
            # the reference's online loop on the scans already resident: from an empty DB, per 512-scan sub-batch
            # ingest -> add -> query at the scan's own epoch; with the DB update inside the timed region and without
            nrep = min(4, len(batches)) * B
            out["extra"] = {"online_replay": online_replay(cc, ctx, [b for b in batches[:min(4, len(batches))]], B, P, nrep, 512, dev)}
            try:
                out["extra"]["dropin_loop"] = dropin_loop(batches[0], P, min(B, 1024))
                # the same files four times over: a drive of KITTI 08's length whose laps 2-4 revisit the first (every query finds candidates)
                d4 = dropin_loop(batches[0], P, min(B, 1024), laps=4)
                d4.pop("what", None)
                out["extra"]["dropin_loop_4_laps"] = d4
            except Exception as e:  # the headline stands on its own
                out["extra"]["dropin_loop"] = {"error": repr(e)}
            # the other single-GPU configurations of BASELINE.json, 8 timed steps each (same pipeline as the headline)
            if args.db_scans == 5000 and args.workload == "kitti":
                try:
                    del batches
                    db.close()
                    db = None
                    torch.cuda.empty_cache()
                    sparse = cc.synth.World()
                    big, rec50 = measure_config(cc, ctx, dev, sparse, 50000, B, 8, 2, P, first_query=60000)
                    out["extra"]["db_50k_sparse"] = big
                    out["extra"]["db_20k_sparse"] = measure_config(cc, ctx, dev, sparse, 20000, B, 8, 2, P, first_query=60000, rec=rec50)[0]
                    del rec50
                    out["extra"]["db_5k_dense"] = measure_config(cc, ctx, dev, cc.synth.World(dense=True), 5000, B, 8, 2, P)[0]
                    # rounds 1-5's headline world (2 k occupied cells, ~14 contours per level, every query revisits a DB place)
                    out["extra"]["db_5k_sparse"] = measure_config(cc, ctx, dev, sparse, 5000, B, 16, 2, P, kernels=True, workload="sparse")[0]
                except Exception as e:
                    out["extra"]["other_configs_error"] = repr(e)
        if world == 1 and not args.no_cpu and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cc, desc_keep.numpy(), n_db, batch_cpu, P, min(args.cpu_sample, B))
        print(json.dumps(out), flush=True)
    if db is not None:
        db.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


# Query lanes of the online loop (sub-batches of a few hundred scans, an append between every two submits): with four
# lanes two sub-batches' chunks are in flight, so the GPU keeps working while the one host thread books the next append
# and collects the previous results (measured: 2 lanes 225-246 k, 3 lanes 257 k, 4 lanes + 512-scan sub-batches 275 k
# scans/s on the dense world).  The steady-state headline keeps the library default of two lanes (1 024-scan batches).
ONLINE_LANES = 4


_MASKED = []  # (handle, wrapper): the raw streams live as long as the process


def masked_stream(torch, dev, spec):
    """A HIP stream whose kernels are dispatched to a subset of the CUs (hipExtStreamCreateWithCUMask), wrapped for torch.
    spec 'a/b': a of every b CUs of each XCD; 'xa/b': a of every b XCDs.  Bit i of the mask is CU i // n_xcd of XCD
    i % n_xcd (the driver deals the bits out to the XCDs in turn)."""
    import ctypes as C
    by_xcd = spec.startswith("x")
    a, b = [int(v) for v in spec.lstrip("x").split("/")]
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    n_xcd = 8
    words = (C.c_uint32 * ((n_cu + 31) // 32))()
    on = 0
    for i in range(n_cu):
        unit = (i % n_xcd) if by_xcd else (i // n_xcd)
        if unit % b < a:
            words[i // 32] |= 1 << (i % 32)
            on += 1
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    w = torch.cuda.ExternalStream(st.value, device=dev)
    _MASKED.append((st, w))
    print("ingest stream on %d of %d CUs (%s)" % (on, n_cu, spec), file=sys.stderr)
    return w


def _replay_pass(cc, ctx, db, chunks, offs, ts, sub, dev, add):
    """One pass of the online loop over `chunks` (device tensors of sub*P points): ingest of sub-batch k+1 runs on its own
    stream while sub-batch k is added (add=True) and queried; scan i is queried against epoch i.  Returns the results."""
    import torch
    s_main = torch.cuda.current_stream(dev)
    s_ing = torch.cuda.Stream(device=dev)
    slots = [torch.empty((sub, cc.DESC_BYTES), dtype=torch.uint8, device=dev) for _ in range(3)]

    def ingest_async(k):
        s_ing.wait_stream(s_main)
        with torch.cuda.stream(s_ing):
            ctx.ingest(chunks[k], offs, out=slots[k % 3])
            if add:  # the device half of the append rides behind the ingest (cc_db_add_scans_prepare)
                db.add_scans_prepare(slots[k % 3])
            ev = torch.cuda.Event()
            ev.record(s_ing)
        return ev

    pending = []
    host = {"ingest_issue": 0.0, "add": 0.0, "submit": 0.0}   # host seconds inside the three calls (where the one thread waits)
    ev = ingest_async(0)
    for k in range(len(chunks)):
        s_main.wait_event(ev)
        ta = time.perf_counter()
        if k + 1 < len(chunks):
            ev = ingest_async(k + 1)
        q = slots[k % 3]
        i0 = k * sub
        tb = time.perf_counter()
        if add:
            db.add_scans(q, ts[i0:i0 + sub], np.arange(i0, i0 + sub, dtype=np.int32))
        tc = time.perf_counter()
        pending.append(db.query_submit(q, np.arange(i0, i0 + sub, dtype=np.int32)))
        td = time.perf_counter()
        host["ingest_issue"] += tb - ta
        host["add"] += tc - tb
        host["submit"] += td - tc
    db.query_wait()
    _replay_pass.host_ms = {k_: v * 1e3 / len(chunks) for k_, v in host.items()}
    return np.concatenate(pending)


def online_replay(cc, ctx, batches, B, P, n, sub, dev):
    """BASELINE config 2's loop shape on n scans that are already in HBM: DB empty at the start, then per sub-batch of
    `sub` scans ingest -> cc_db_add_scans (addScan + pushAndBalance per scan) -> query with scan i at epoch i, everything
    inside the timed region (`with_update`); and the same pass against a DB that already holds all n scans, the epochs
    giving every query the same view (`without_update`: ingest + query only).  Both passes must return identical results."""
    import torch
    chunks = []
    for b in batches:
        for j in range(0, B, sub):
            chunks.append(b[j * P:(j + sub) * P])
    chunks = chunks[:n // sub]
    n = len(chunks) * sub
    offs = np.arange(sub + 1, dtype=np.int64) * P
    ts = np.arange(n, dtype=np.float64) / 10.0
    out = {"scans": n, "sub_batch": sub, "what": "empty DB, then per sub-batch: ingest -> addScan/pushAndBalance -> query (scan i at epoch i); "
           "10 Hz stamps, shipped 15 s / 25 s delays (test/batch_bin_test.cpp:131-237, contour_db.h:814-843)"}
    res = {}
    for mode in ("warmup", "with_update", "without_update"):
        rates = []
        for rep in range(1 if mode == "warmup" else 3):  # a pass lasts ~14 ms: the median of three (each on a fresh DB) is what is quoted
            db = cc.Database(ctx, capacity=n + 16)
            db.set_lanes(ONLINE_LANES)
            if mode == "without_update":
                for k, c in enumerate(chunks):
                    d = ctx.ingest(c, offs)
                    db.add_scans(d, ts[k * sub:(k + 1) * sub], np.arange(k * sub, (k + 1) * sub, dtype=np.int32))
            torch.cuda.synchronize()
            gc.collect()
            gc.disable()
            t0 = time.perf_counter()
            r = _replay_pass(cc, ctx, db, chunks, offs, ts, sub, dev, add=(mode != "without_update"))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gc.enable()
            res[mode] = r
            rates.append((n / dt, dt, {k_: round(v, 4) for k_, v in _replay_pass.host_ms.items()}))
            db.close()
        if mode != "warmup":
            rates.sort(key=lambda t_: t_[0])
            med = rates[len(rates) // 2]
            out["scans_per_s_" + mode] = med[0]
            out["scans_per_s_" + mode + "_min_max"] = [rates[0][0], rates[-1][0]]
            out["ms_per_sub_batch_" + mode] = med[1] / len(chunks) * 1e3
            out["host_ms_per_sub_batch_" + mode] = med[2]
    out["loop_closures"] = int((res["with_update"]["n_res"] > 0).sum())
    out["identical_results"] = bool(res["with_update"].tobytes() == res["without_update"].tobytes())
    return out


def bench_seq(cc, args, dev, local_rank, world, rank, dist):
    """--workload seq: the online loop over one long dense-world sequence; a step = one sub-batch (ingest + DB update +
    query), W warm-up sub-batches from the start of the sequence, then exactly K timed ones (the DB keeps growing)."""
    import torch
    sub, P = args.seq_batch, 64 * 1875
    n = (args.seq_scans // sub) * sub
    nchunk = n // sub
    W = min(args.warmup, nchunk - 1)
    K = min(args.steps if args.steps_given else nchunk - W, nchunk - W)
    wld = cc.synth.World(dense=True)
    ctx = cc.Context(local_rank, max_batch=max(sub, 256))
    t_setup = time.time()
    chunks = []
    for k in range(W + K):   # every rank replays its own stretch of the trajectory (weak scaling, independent sequences)
        x, _, _ = cc.synth.make_sequence(sub, world=wld, device=dev, start=rank * n + k * sub)
        chunks.append(x.reshape(-1, 4).contiguous())
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    offs = np.arange(sub + 1, dtype=np.int64) * P
    ts = np.arange((W + K) * sub, dtype=np.float64) / 10.0
    out_modes = {}
    rates_with = []
    reps = max(1, args.seq_repeats)
    for mode in ["with_update"] * reps + ["without_update"]:
        db = cc.Database(ctx, capacity=(W + K) * sub + 16)
        db.set_lanes(args.lanes if args.lanes else ONLINE_LANES)
        if mode == "without_update":
            for k, c in enumerate(chunks):
                d = ctx.ingest(c, offs)
                db.add_scans(d, ts[k * sub:(k + 1) * sub], np.arange(k * sub, (k + 1) * sub, dtype=np.int32))
        # warm-up sub-batches (they also fill the DB in the with_update pass), then the K timed ones in ONE pipelined pass
        s_main = torch.cuda.current_stream(dev)
        s_ing = torch.cuda.Stream(device=dev)
        slots = [torch.empty((sub, cc.DESC_BYTES), dtype=torch.uint8, device=dev) for _ in range(3)]

        def ingest_async(k):
            s_ing.wait_stream(s_main)
            with torch.cuda.stream(s_ing):
                ctx.ingest(chunks[k], offs, out=slots[k % 3])
                if mode == "with_update":  # the device half of the append rides behind the ingest (cc_db_add_scans_prepare)
                    db.add_scans_prepare(slots[k % 3])
                ev = torch.cuda.Event()
                ev.record(s_ing)
            return ev

        def one(k, ev):
            s_main.wait_event(ev)
            nxt = ingest_async(k + 1) if k + 1 < W + K else None
            q = slots[k % 3]
            idx = np.arange(k * sub, (k + 1) * sub, dtype=np.int32)
            if mode == "with_update":
                db.add_scans(q, ts[k * sub:(k + 1) * sub], idx)
            return db.query_submit(q, idx), nxt

        res = []
        ev = ingest_async(0)
        for k in range(W):
            r, ev = one(k, ev)
            res.append(r)
        db.query_wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        gc.collect()
        gc.disable()  # a collector pause of a few ms would be a third of this timed region
        t0 = time.perf_counter()
        for k in range(W, W + K):
            r, ev = one(k, ev)
            res.append(r)
        db.query_wait()
        torch.cuda.synchronize()
        gc.enable()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if mode == "with_update":
            rates_with.append((K * sub * world / dt, dt))
            if "with_update" in out_modes and out_modes["with_update"][1].tobytes() != np.concatenate(res).tobytes():
                raise SystemExit("bench.py --workload seq: two passes of the same online loop returned different results")
        out_modes[mode] = (dt, np.concatenate(res))
        db.close()
    if rank == 0:
        _, r = out_modes["with_update"]
        dt2, r2 = out_modes["without_update"]
        rates_with.sort()
        value, dt = rates_with[len(rates_with) // 2]   # the median pass
        out = {"metric": "scans/sec ingest+query (120k-pt scan vs 5k-scan DB); max-F1 parity",
               "value": value, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
               "config": {"workload": "online replay of one synthetic Velodyne-64 sequence (64x1875=120000 pts, dense world, 10 Hz, "
                                      "%d scans per rank from an empty DB): per %d-scan sub-batch ingest -> addScan/pushAndBalance -> query, "
                                      "scan i at epoch i; the DB update is INSIDE the timed step (BASELINE config 2's loop, "
                                      "test/batch_bin_test.cpp:131-237)" % ((W + K) * sub, sub),
                          "world": "dense", "seq_scans": (W + K) * sub, "sub_batch": sub, "points_per_scan": P,
                          "parallelism": "independent sequences x%d" % world},
               "extra": {"online_replay": {"scans_per_s_with_update": value, "passes_with_update": len(rates_with),
                                           "scans_per_s_with_update_min_median_max": [rates_with[0][0], value, rates_with[-1][0]],
                                           "scans_per_s_without_update": K * sub * world / dt2,
                                           "ms_per_sub_batch_with_update": dt / K * 1e3, "ms_per_sub_batch_without_update": dt2 / K * 1e3,
                                           "loop_closures": int((r["n_res"] > 0).sum()),
                                           "identical_results": bool(r.tobytes() == r2.tobytes())}},
               "roofline": {"bound": "hbm", "kernel": "whole step", "achieved": value / world * P * 16 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": value / world * P * 16 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "note": "point stream bytes (16 B x 120000 per scan) over the whole online step; per-kernel figures: default run"},
               "setup_s": setup_s}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


class _HarnessCtx:
    """CC_BENCH_HARNESS=emu: cc.Context on the C-ABI's CPU-harness build (tests/emu_api.py), CPU torch tensors in and out."""

    def __init__(self, cc, max_batch):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_api
        self.L = cc.L
        self.api = emu_api.EmuApi(cc.L)
        self.h = self.api.create(max_batch=max_batch)

    def ingest(self, x, offs, out=None):
        import torch
        d = self.api.ingest(self.h, x.numpy().reshape(-1, 4), offs)
        t = torch.from_numpy(d.view(np.uint8).reshape(len(d), -1))
        if out is not None:
            out.copy_(t)
            return out
        return t

    def pack(self, d):
        import torch
        hot, feat = self.api.pack(self.h, d.numpy().view(self.L.scan_desc_dt).reshape(-1))
        return torch.from_numpy(hot), torch.from_numpy(feat)

    def close(self):
        pass


class _HarnessDb:
    def __init__(self, ctx, capacity):
        self.ctx, self.h = ctx, ctx.api.db_create(ctx.h, cap=capacity)

    def add_packed(self, hot, feat, ts, seeds):
        self.ctx.api.db_add_packed(self.h, hot.numpy(), feat.numpy(), ts, seeds)

    def query(self, q, epochs):
        return self.ctx.api.db_query(self.h, q.numpy().view(self.ctx.L.scan_desc_dt).reshape(-1), np.ascontiguousarray(epochs, np.int32))

    query_submit = query

    def query_wait(self):
        pass

    def set_lanes(self, n):
        pass

    def close(self):
        pass


def _read_kernel_ms(cc, ctx, db, B):
    """HIP-event sums of the library's profiling hooks since the last read, per step of B scans: ingest kernels per launch
    (one launch per step), query-side kernel groups per B queries (a group = the kernels of one stage of a chunk's chain)."""
    import ctypes as C
    ms2 = (C.c_double * 2)()
    nl = C.c_int()
    cc.lib().cc_profile_read(ctx.h, ms2, C.byref(nl))
    ms5 = (C.c_double * 5)()
    nl2 = C.c_int()
    cc.lib().cc_db_profile_read(db.h, ms5, C.byref(nl2))
    a, b = max(nl.value, 1), max(nl2.value, 1) / float(B)   # ingest launches (one per step) | queries -> steps
    return {"cc_k_rasterize": ms2[0] / a, "cc_k_contours": ms2[1] / a, "cc_k_knn": ms5[0] / b, "cc_k_check": ms5[1] / b,
            "cc_k_merge": ms5[2] / b, "cc_k_gmm": ms5[3] / b, "cc_k_final": ms5[4] / b}


def algorithmic_bytes(d, res, B, P, n_db):
    """ALGORITHMIC bytes per step and kernel group (DESIGN.md "Kernels"): what the step has to move, independent of how.
    d = descriptors of (a sample of) the step's scans, res = the step's query results.
      K1 streams the xyzi records once (16 B/point) and emits the list of the scan's active cells (the dense BEV + per-cell
         positions only on request or when the list overflows: neither happens in this workload);
      K2 reads that list and emits the descriptor used downstream;
      K3 reads the layers' key matrices once and, per anchor key, the 40-B key and <= nnk 12-B hits;
      K4 reads, per KNN hit, the hit and two contour records, per anchor-similar pair the two 256-bit rings, per check that
         reaches the pairing the two 600-B BCIs, and writes a 104-B record per pass;
      merge reads those records; K5 reads two ellipse tables (32 B/ellipse) per problem and writes 64 B.
    Returns (per-group bytes, split): the split sorts the same bytes into `compulsory` (point stream in, descriptors out, the
    key matrices and query keys once, results out), `intermediate` (the active-cell list that goes from K1 to K2 through HBM) and
    `logical_gather` (per-hit / per-check / per-problem reads of DB records, mostly served by L2 / Infinity Cache)."""
    n_pix = float(d["n_pix"].mean())
    desc_emit = float(np.mean(72 + 16 + 1440 + 36 * 600 + d["n_stored"].sum(1) * 76))
    f = {k: float(res[k].mean()) for k in ("n_knn_hits", "cand_aft_check1", "cand_aft_check3", "n_cand_tidy")}
    n_keys_db = 3 * 6 * n_db
    # round 6: K1 also lists the scan's active cells (above the lowest level) for K2 -- 15 B per entry ((row, col), level count,
    # height, continuous position) + a 16-B header -- and K2 starts from that list instead of re-reading the image.  The entry
    # count is not in the descriptor; the level-0 contours' cells (>= 3-cell components of the same level set) are a lower bound
    active_list = 16 + 15 * float(d["layer_cell_cnt"][:, 0].mean())
    alg = {"cc_k_rasterize": B * (P * 16 + active_list),   # (rounds 1-6a: + bev, the dense image and positions nobody read)
           "cc_k_contours": B * (active_list + desc_emit),
           "cc_k_knn": n_keys_db * 44 + B * 18 * 40 + B * f["n_knn_hits"] * 12,
           "cc_k_check": B * (f["n_knn_hits"] * (12 + 2 * 76) + f["cand_aft_check1"] * (64 + 2 * 600) + f["cand_aft_check3"] * 104),
           "cc_k_merge": B * f["cand_aft_check3"] * 104,
           "cc_k_gmm": B * f["n_cand_tidy"] * (2 * 45 * 32 + 64),
           "cc_k_final": B * 64}
    compulsory = B * P * 16 + B * desc_emit + n_keys_db * 44 + B * 18 * 40 + B * 64
    intermediate = 2 * B * active_list   # the list, both ways (K1 writes the dense image only on request / list overflow since round 6)
    split = {"compulsory": compulsory, "intermediate": intermediate, "logical_gather": sum(alg.values()) - compulsory - intermediate}
    return alg, split


def roofline_object(alg, split, kms, kms_iso, B, n_db, workload, ms_per_step, P):
    """The `roofline` object of a JSON line.  Dominant kernel group = largest ISOLATED time per step (stable from run to run;
    in the timed region kernels of three streams share the GPU and their event times move with the overlap).  `achieved` /
    `frac` = algorithmic bytes / the group's HIP-event time in the timed region (the contract's definition);
    `*_isolated` = the same bytes over the group's time with the GPU to itself; `kernels` lists every group both ways next
    to its PMC traffic (profiles/*_pmc_summary.json of this same configuration, null otherwise)."""
    iso = kms_iso or kms
    dom = max(iso, key=lambda k: iso[k])
    rows = []
    for k in kms:
        tr, _src = pmc_traffic(k, B, n_db, workload)
        rows.append({"kernel": k, "algorithmic_bytes": alg[k], "ms_isolated": iso[k], "ms_in_step": kms[k],
                     "frac_isolated": alg[k] / (iso[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if iso[k] > 0 else None,
                     "frac_in_step": alg[k] / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if kms[k] > 0 else None,
                     "pmc_bytes": tr, "traffic_ratio": (tr / alg[k]) if tr else None})
    if not kms[dom] > 0 or not iso[dom] > 0:   # no device timers (the CPU-harness launcher test)
        return {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernels": rows, "step_algorithmic_bytes": sum(alg.values()), "step_bytes_split": split}
    ach = alg[dom] / (kms[dom] * 1e-3) / 1e9
    ach_iso = alg[dom] / (iso[dom] * 1e-3) / 1e9
    tr, src = pmc_traffic(dom, B, n_db, workload)
    k1 = "cc_k_rasterize"
    total = sum(alg.values())
    return {"bound": "hbm", "kernel": dom, "dominant_by": "isolated HIP-event time per step",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "achieved_isolated": ach_iso, "frac_isolated": ach_iso / HBM_PEAK_GBS,
            "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": alg[dom],
            "kernels": rows, "kernels_ms_per_launch": kms, "kernels_ms_per_launch_isolated": kms_iso,
            "isolated_sum_ms": sum(iso.values()),
            "rasterize_GBs": alg[k1] / (iso[k1] * 1e-3) / 1e9 if iso[k1] > 0 else None,
            "rasterize_frac_isolated": alg[k1] / (iso[k1] * 1e-3) / 1e9 / HBM_PEAK_GBS if iso[k1] > 0 else None,
            "step_algorithmic_bytes": total, "step_bytes_split": split,
            "step_hbm_frac": total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_hbm_frac_compulsory": split["compulsory"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_hbm_frac_logical_gather": split["logical_gather"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "ingest_roofline_scans_per_s": HBM_PEAK_GBS * 1e9 / (P * 16)}


def measure_config(cc, ctx, dev, wld, n_db, B, K, W, P, first_query=None, rec=None, kernels=False, workload=None, cpu_sample=0):
    """A short run of the headline step on another configuration (world, DB size), same pipeline as the headline: ingest of
    batch s + 1 on its own stream while batch s is queried (cc_db_query_submit), every result collected inside the timed
    region.  `rec` = packed records of a DB built earlier whose first n_db scans are used.  Returns (figures, records)."""
    import torch
    HB, FB = cc.packed_sizes()
    desc_keep = None
    if rec is None or rec.shape[0] < n_db:
        rec = torch.empty((n_db, HB + FB), dtype=torch.uint8, device=dev)
        tmp = torch.empty((256, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
        if cpu_sample:   # the CPU leg rebuilds its DB from the scans' descriptors
            desc_keep = torch.empty((n_db, cc.DESC_BYTES), dtype=torch.uint8, device="cpu")
        for c0 in range(0, n_db, 256):
            c1 = min(c0 + 256, n_db)
            x, _, _ = cc.synth.make_sequence(c1 - c0, world=wld, device=dev, start=c0)
            d = ctx.ingest(x.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P, out=tmp[:c1 - c0])
            hot, feat = ctx.pack(d)
            rec[c0:c1, :HB] = hot
            rec[c0:c1, HB:] = feat
            if desc_keep is not None:
                desc_keep[c0:c1].copy_(d)
    db = cc.Database(ctx, capacity=n_db + 16)
    db.add_packed(rec[:n_db, :HB].contiguous(), rec[:n_db, HB:].contiguous(), np.arange(n_db) / 10.0, np.arange(n_db, dtype=np.int32))
    q0 = n_db if first_query is None else first_query
    batches = []
    for s_ in range(W + K):
        x, _, _ = cc.synth.make_sequence(B, world=wld, device=dev, start=q0 + s_ * B)
        batches.append(x.reshape(-1, 4).contiguous())
    offs = np.arange(B + 1, dtype=np.int64) * P
    epochs = np.full(B, n_db, np.int32)
    slots = [torch.empty((B, cc.DESC_BYTES), dtype=torch.uint8, device=dev) for _ in range(2)]
    s_main, s_ing = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)

    def ingest_async(k):
        s_ing.wait_stream(s_main)
        with torch.cuda.stream(s_ing):
            ctx.ingest(batches[k], offs, out=slots[k & 1])
            ev = torch.cuda.Event()
            ev.record(s_ing)
        return ev

    def run(first, count):
        res = []
        ev = ingest_async(first)
        for k in range(first, first + count):
            s_main.wait_event(ev)
            if k + 1 < first + count:
                ev = ingest_async(k + 1)
            res.append(db.query_submit(slots[k & 1], epochs))
        db.query_wait()
        return res
    run(0, W)
    torch.cuda.synchronize()
    if kernels:
        cc.lib().cc_profile_enable(ctx.h, 1)
        cc.lib().cc_db_profile_enable(db.h, PROF_EVERY)
        _read_kernel_ms(cc, ctx, db, B)
    t0 = time.perf_counter()
    res = run(W, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    found = int(sum((r["n_res"] > 0).sum() for r in res))
    flags = int(sum((r["flags"] != 0).sum() for r in res))
    d = cc.desc_to_numpy(slots[(W + K - 1) & 1][:64])
    out = {"scans_per_s": K * B / dt, "ms_per_step": dt / K * 1e3, "steps": K, "warmup": W, "db_scans": n_db, "batch": B,
           "loop_closures": found, "queries": K * B, "flagged_queries": flags,
           "occupied_cells_mean": round(float(d["n_pix"].mean()), 1), "contours_per_level_mean": [round(float(v), 1) for v in d["n_cont"].mean(0)],
           "valid_db_keys_per_scan_mean": round(float((np.abs(d["keys"].reshape(len(d), 6, 6, 10)[:, 1:4]).sum(-1) > 0).sum((1, 2)).mean()), 2)}
    if kernels:   # per-kernel-group times: in the timed region and with the GPU to itself (ISO_STEPS extra steps on one stream)
        kms = _read_kernel_ms(cc, ctx, db, B)
        db.set_lanes(1)
        cc.lib().cc_db_profile_enable(db.h, 1)
        s_main.synchronize()
        for k in range(W, W + min(ISO_STEPS, K)):
            ctx.ingest(batches[k], offs, out=slots[0])
            last = db.query(slots[0], epochs)
        torch.cuda.synchronize()
        kms_iso = _read_kernel_ms(cc, ctx, db, B)
        cc.lib().cc_profile_enable(ctx.h, 0)
        alg, split = algorithmic_bytes(d, res[-1], B, P, n_db)
        roof = roofline_object(alg, split, kms, kms_iso, B, n_db, workload, dt / K * 1e3, P)
        out["roofline"] = {k_: roof[k_] for k_ in ("kernel", "frac", "frac_isolated", "kernels", "isolated_sum_ms", "rasterize_frac_isolated",
                                                   "step_hbm_frac", "step_hbm_frac_compulsory", "step_bytes_split")}
        out["queries_with_a_loop_closure"] = round(found / float(K * B), 4)
        out["checks_per_query_mean"] = round(float(np.mean([r["cand_aft_check1"].mean() for r in res])), 1)
        out["knn_hits_per_query_mean"] = round(float(np.mean([r["n_knn_hits"].mean() for r in res])), 1)
        out["correlation_problems_per_query_mean"] = round(float(np.mean([r["n_cand_tidy"].mean() for r in res])), 2)
    db.close()
    if cpu_sample and desc_keep is not None:
        # the oracle on the SAME workload (same DB, the first cpu_sample scans of the first timed batch), one thread like the
        # reference: what north_star's "x times the CPU path on KITTI-08-shaped input" is measured against
        try:
            cb = cpu_baseline(cc, desc_keep.numpy(), n_db, batches[W], P, cpu_sample, all_cores=False)
            out["cpu_baseline"] = cb
            out["gpu_over_one_cpu_core"] = out["scans_per_s"] / cb["value"]
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out, rec


def dropin_loop(batch0, P, n, laps=1, reps=3):
    """The reference's per-scan driver loop through the C++ class mirror (hostcpp/examples/batch_bin_test.cpp: the
    reference's test/batch_bin_test.cpp without ROS, same ContourManager / ContourDB calls): n KITTI-format .bin files are
    read one by one, makeBEV + makeContoursRecurs -> queryRangedKNN -> addScan + pushAndBalance per scan, DB empty at the
    start.  Wall-clock seconds per call from the driver's own stage timers (tools/bm_util.h).  laps > 1: the list names the n
    files `laps` times over (time stamps, ids and poses keep counting): a drive of laps * n scans whose later laps revisit
    the first one -- KITTI 08 has 4 071 scans."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "contour-context_amd", "hostcpp", "bin", "batch_bin_test")
    if not os.path.exists(exe):
        import __graft_entry__
        exe = __graft_entry__.build_dropin_driver()
    tmp = tempfile.mkdtemp(prefix="cc_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        xs = batch0[:n * P].cpu().numpy().reshape(n, P, 4)
        with open(os.path.join(tmp, "scans.txt"), "w") as f, open(os.path.join(tmp, "poses.txt"), "w") as g:
            for i in range(n):
                xs[i].tofile(os.path.join(tmp, "%06d.bin" % i))
            for i in range(n * laps):
                p = os.path.join(tmp, "%06d.bin" % (i % n))
                f.write("%.6f %d %s\n" % (i / 10.0, i, p))
                g.write("%.6f 1 0 0 %.3f 0 1 0 0 0 0 1 0\n" % (i / 10.0, float(i)))
        # The files are read once before the driver starts: the first read(2) of a freshly WRITTEN tmpfs page activates it under
        # the kernel's LRU lock, which serialises parallel readers (measured on the MI355X host: 4 threads 163 us per file on that
        # first pass, 28 us on any later one; profiles/r5/read_pinned_bench.cpp) -- an artefact of producing the input right
        # here, not a property of reading scans.  (CC_DROPIN_COLD_FILES=1 skips this.)
        if not os.environ.get("CC_DROPIN_COLD_FILES"):
            for i in range(n):   # (n: files, here)
                with open(os.path.join(tmp, "%06d.bin" % i), "rb", buffering=0) as f:
                    while f.read(1 << 22):
                        pass
        cfg = open(os.path.join(ROOT, "contour-context_amd", "hostcpp", "examples", "batch_bin_test_config.yaml")).read()
        cfg = cfg.replace("/path/to/ts-sens_pose-kitti08.txt", os.path.join(tmp, "poses.txt"))
        cfg = cfg.replace("/path/to/ts-lidar_bins-kitti08.txt", os.path.join(tmp, "scans.txt"))
        cfg = cfg.replace("/path/to/outcome-kitti08.txt", os.path.join(tmp, "outcome.txt"))
        open(os.path.join(tmp, "cfg.yaml"), "w").write(cfg)
        def one_run():
            t0 = time.perf_counter()
            prefix = os.environ.get("CC_DROPIN_PREFIX", "").split()   # tuning aid: e.g. a rocprofv3 command line in front of the driver
            r = subprocess.run(prefix + [exe, os.path.join(tmp, "cfg.yaml")], capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "driver exit code %d: %s" % (r.returncode, r.stderr[-300:])}
            n_files, n_scans = n, n * laps
            out = {"scans": n_scans, "files": n_files, "what": "hostcpp/examples/batch_bin_test (the reference driver's loop through the class mirror): per scan "
                   ".bin file (tmpfs, in the page cache) -> makeBEV + makeContoursRecurs -> queryRangedKNN -> addScan + pushAndBalance, DB empty at the start; "
                   "seconds per call = wall clock inside the driver; the run with the median loop rate of `scans_per_s_runs`", "process_wall_s": wall}
            for line in r.stdout.splitlines():
                p = line.split()
                for name in ("make bev", "queryRangedKNN (wall)", "Update database"):
                    if name in line and len(p) >= 6:
                        try:
                            out.setdefault("seconds_per_call", {})[name] = float(p[-5])
                        except ValueError:
                            pass
                if line.startswith("Loop wall time:"):
                    out["loop_wall_s"] = float(p[3])
                    out["scans_per_s"] = n_scans / float(p[3])
                if line.startswith("Construction time:"):   # evaluator + ContourDB constructors: lists, device runtime, stream pool
                    out["construction_s"] = float(p[2])
            try:   # what the driver decided, scan by scan (outcome file without its two path columns): equal for every read-ahead setting
                import hashlib
                h = hashlib.sha1()
                for line in open(os.path.join(tmp, "outcome.txt")):
                    h.update("\t".join(line.rstrip("\n").split("\t")[:6]).encode() + b"\n")
                out["outcome_sha1"] = h.hexdigest()
            except OSError:
                pass
            if "construction_s" in out and "loop_wall_s" in out:
                out["scans_per_s_with_construction"] = n_scans / (out["loop_wall_s"] + out["construction_s"])
            if "seconds_per_call" in out:
                out["ms_per_scan_three_calls"] = 1e3 * sum(out["seconds_per_call"].values())
            for line in r.stderr.splitlines():
                if line.startswith("[evaluator helper"):     # CC_EVAL_TIMERS=1: what the prefetch thread spent per scan
                    out["helper_thread"] = line
                if line.startswith("[ContourDB read-ahead") or line.startswith("[cc_db appends"):  # CC_EVAL_TIMERS=1 / CC_ADD_TIMERS=1
                    out.setdefault("db_read_ahead", []).append(line)
            return out

        # the driver is run `reps` times on the same files (a process that starts a device runtime, creates streams and pins memory is
        # at the mercy of whatever else the host is doing for a few hundred ms: one run in eight of this loop came out at half the rate
        # of the others); the run with the median loop rate is reported, every run's rate is listed
        runs = [one_run() for _ in range(max(1, reps))]
        bad = [r_ for r_ in runs if "error" in r_]
        if bad:
            return bad[0]
        runs.sort(key=lambda r_: r_.get("scans_per_s", 0.0))
        out = runs[len(runs) // 2]
        out["scans_per_s_runs"] = [round(r_.get("scans_per_s", 0.0), 1) for r_ in runs]
        out["outcome_identical_across_runs"] = len({r_.get("outcome_sha1") for r_ in runs}) == 1
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container often shows all of
    the host's CPUs but is only scheduled on a few), one process per physical core (half of the hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    n = max(1, n // 2)
    if quota is not None:
        n = max(1, min(n, int(round(quota))))
    return min(n, 128)


def _git_head():
    try:
        import subprocess
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


def pmc_traffic(kernel, batch, db_scans, workload):
    """HBM bytes per STEP of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*_pmc_summary.json: FETCH_SIZE/WRITE_SIZE per launch, gfx950 x2 correction applied where it is calibrated).
    A summary is only used if it was taken on the SAME configuration (batch, DB size, world) -- otherwise null.
    Query kernels are launched once per chunk: a streamed step of 1024 queries or more goes out in chunks of 1024, a smaller
    one is cut over the two lanes (cc_db_query_submit)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))  # round tags sort by name
    parts = {"cc_k_check": ["cc_k_check_a", "cc_k_check_b1", "cc_k_compact_cstl", "cc_k_check_b2", "cc_k_check_c"],
             "cc_k_gmm": ["cc_k_gmm_init", "cc_k_select", "cc_k_gmm_refine"]}.get(kernel, [kernel])
    optional = {"cc_k_contours": ["cc_k_contours_mid", "cc_k_contours_big"]}.get(kernel, [])  # (the launches behind the list kernel, usually with empty queues)
    d = ks = None
    for f in reversed(files):   # the newest summary taken on this configuration
        try:
            cand = json.load(open(f))
        except (OSError, ValueError):
            continue
        if cand.get("batch_scans") == batch and cand.get("db_scans") == db_scans and cand.get("workload", "sparse") == workload:
            d, ks, files = cand, cand.get("kernels", {}), [f]
            break
    if d is None:
        return None, None
    if any(p not in ks or "hbm_bytes_per_launch" not in ks[p] for p in parts):
        return None, None
    per_step = 1 if kernel in ("cc_k_rasterize", "cc_k_contours") else ((batch + 1023) // 1024 if batch >= 1024 else 2)
    src = os.path.relpath(files[-1], ROOT) + (" (taken at %s)" % d["git_head"] if d.get("git_head") else "")
    parts = parts + [p for p in optional if p in ks and "hbm_bytes_per_launch" in ks[p]]
    return sum(ks[p]["hbm_bytes_per_launch"] for p in parts) * per_step, src


def _cpu_worker(shm_dir, wid, n_workers, n_db, P, n_q_total, repeats, barrier, out_q):
    """One of the N processes of the all-cores CPU measurement: builds its own copy of the DB from the shared descriptor
    file, then ingests + queries its disjoint slice of the sample scans `repeats` times."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    L = O.L
    O.use_ref_kdtree(True)
    desc = np.load(os.path.join(shm_dir, "db_desc.npy"), mmap_mode="r").view(L.scan_desc_dt).reshape(-1)
    xq = np.load(os.path.join(shm_dir, "q_xyzi.npy"), mmap_mode="r")
    odb = O.DB()
    for i in range(n_db):
        sc = O.Scan.from_desc(desc[i], int_id=i)
        odb.add_scan(sc, i / 10.0)
        odb.push_and_balance(i, i / 10.0)
    mine = list(range(wid, n_q_total, n_workers))
    barrier.wait()
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < repeats:  # `repeats` = seconds of work per process
        i = mine[done % len(mine)]
        sc = O.Scan(np.ascontiguousarray(xq[i]), int_id=n_db + i, keep_cells=False)
        sc.clear_image()
        odb.query(sc)
        done += 1
    out_q.put((wid, done, t0, time.perf_counter()))


def cpu_baseline(cc, db_desc, n_db, batch0, P, n_q, all_cores=True):
    """The reference's single-threaded code path on the CPU restatement (oracle/, kd-tree = the reference's vendored
    nanoflann when oracle/_ref is built) on the SAME workload.  The DB is rebuilt on the CPU side from the descriptors of
    the DB scans (untimed: addScan + pushAndBalance per scan); then the first n_q scans of the first timed batch go
    through the reference's per-scan loop with its five stage timers (tools/bm_util.h names, contour_db.h:729-787):
    make bev (ContourManager ctor + makeBEV + makeContoursRecurs), KNN search, Constell, L2 opt, and -- after the
    queries, so that every query sees the same DB as a GPU step does -- Update database (addScan + pushAndBalance).
    `value` covers what a GPU step covers: ingest + query.  Besides the single thread (the reference has no threading),
    an all-cores figure is MEASURED: N processes, each with its own copy of the DB, on disjoint scans."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    L = O.L
    kd = O.use_ref_kdtree(True)
    desc = db_desc.view(L.scan_desc_dt).reshape(-1)[:n_db]
    odb = O.DB()
    t_build = time.perf_counter()
    for i in range(n_db):
        s = O.Scan.from_desc(desc[i], int_id=i)
        odb.add_scan(s, i / 10.0)
        odb.push_and_balance(i, i / 10.0)
    t_build = time.perf_counter() - t_build
    xq = batch0[:n_q * P].cpu().numpy().reshape(n_q, P, 4)
    odb.timers(reset=True)
    found = 0
    t_ing = 0.0
    scans = []
    t0 = time.perf_counter()
    for i in range(n_q):
        ta = time.perf_counter()
        s = O.Scan(xq[i], int_id=n_db + i, keep_cells=False)
        s.clear_image()
        t_ing += time.perf_counter() - ta
        r = odb.query(s)
        scans.append(s)
        found += int(r["n_res"] > 0)
    dt = time.perf_counter() - t0
    tq = odb.timers()
    t_upd = time.perf_counter()
    for i, s in enumerate(scans):
        odb.add_scan(s, (n_db + i) / 10.0)
        odb.push_and_balance(n_db + i, (n_db + i) / 10.0)
    t_upd = time.perf_counter() - t_upd
    out = {"value": n_q / dt, "unit": "scans/s", "cores": 1, "kind": "port",
           "sample": "%d scans of the timed batches (spread evenly over the timed drive in the default run): ingest + query against the same %d-scan DB (CPU-side DB rebuilt "
                     "from the scans' descriptors, untimed, %.1f s; kd-tree=%s); %d loop closures found"
                     % (n_q, n_db, t_build, "reference nanoflann (oracle/_ref)" if kd else "exact scan", found),
           "seconds_per_scan": {"make bev": t_ing / n_q, "KNN search": tq["KNN search"] / n_q, "Constell": tq["Constell"] / n_q,
                                "L2 opt": tq["L2 opt"] / n_q, "Update database (outside `value`, like the GPU step)": t_upd / n_q},
           "online_loop_scans_per_s": n_q / (dt + t_upd),
           "host_cpus": os.cpu_count(), "host_cpu_model": _cpu_model(),
           "ratios_this_supports": "GPU value / cpu_baseline.value is a ratio to ONE host core (the reference is single-threaded per "
                                   "scan); GPU value / all_cores_measured.value is the ratio to N independent single-threaded copies "
                                   "(N = the cores this container may use, not a socket: the 'single-socket' figure of north_star is "
                                   "not measurable here)"}
    if not all_cores:
        return out
    # ---- measured all-cores figure: N independent single-threaded copies on disjoint scans
    try:
        import multiprocessing as mp
        import shutil
        import tempfile
        n_workers = _usable_cores()
        shm = tempfile.mkdtemp(prefix="cc_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            np.save(os.path.join(shm, "db_desc.npy"), db_desc[:n_db])
            n_all = min(batch0.shape[0] // P, max(n_q, 2 * n_workers))
            np.save(os.path.join(shm, "q_xyzi.npy"), batch0[:n_all * P].cpu().numpy().reshape(n_all, P, 4))
            ctxm = mp.get_context("spawn")
            barrier = ctxm.Barrier(n_workers + 1)
            out_q = ctxm.Queue()
            repeats = 4.0  # seconds of work per process
            procs = [ctxm.Process(target=_cpu_worker, args=(shm, w, n_workers, n_db, P, n_all, repeats, barrier, out_q))
                     for w in range(n_workers)]
            for p in procs:
                p.start()
            barrier.wait(timeout=300)
            res = [out_q.get(timeout=600) for _ in procs]
            for p in procs:
                p.join(timeout=60)
            t_first, t_last = min(r[2] for r in res), max(r[3] for r in res)
            total = sum(r[1] for r in res)
            out["all_cores_measured"] = {"value": total / (t_last - t_first), "unit": "scans/s", "processes": n_workers,
                                         "usable_cores": "affinity mask / 2, capped by the cgroup CPU quota",
                                         "scans": total, "seconds": t_last - t_first,
                                         "what": "N single-threaded copies of the reference path (each with its own copy of the "
                                                 "DB) on disjoint scans of the batch, ingest + query, wall clock from the first "
                                                 "start to the last finish"}
        finally:
            shutil.rmtree(shm, ignore_errors=True)
    except Exception as e:  # the single-thread figure stands on its own
        out["all_cores_measured"] = {"value": None, "error": repr(e)}
    return out


if __name__ == "__main__":
    main()
