"""Multi-GPU path on CPU (gloo, world_size 2): scan-sharded ingest -> ONE all-gather of the descriptor blocks ->
replicated DB -> query-sharded scoring.  Compute runs through the product's C-ABI in its CPU build (tests/emu);
the collective is torch.distributed exactly as bench.py uses it (nccl = RCCL on the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cc_amd
    import emu_api
    cc = cc_amd.load()
    L = cc.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 48
    shard = n // world
    lo = rank * shard
    x, _, _ = cc.synth.make_sequence(shard, world=w, beams=16, azim=450, start=lo)
    P = x.shape[1]
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    local = api.ingest(ctx, x.numpy().reshape(-1, 4), np.arange(shard + 1, dtype=np.int64) * P)
    t_local = torch.from_numpy(local.view(np.uint8).reshape(shard, -1).copy())
    t_all = torch.empty((n, t_local.shape[1]), dtype=torch.uint8)
    dist.all_gather_into_tensor(t_all, t_local)          # the path's only exchange
    desc_all = t_all.numpy().view(L.scan_desc_dt).reshape(-1)
    ts = np.arange(n) / 10.0
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, desc_all, ts, np.arange(n, dtype=np.int32))
    sizes, ranges = api.bucket_state(db)
    qs = np.arange(40 + rank, 48, world).astype(np.int32)     # query-sharded
    res = api.db_query(db, desc_all[qs], qs)
    np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), sizes=sizes, ranges=ranges, qs=qs, res=res.view(np.uint8),
             desc=t_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ingest_allgather_query(tmp_path, cc, oracle):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["desc"], r1["desc"]), "all-gathered DB must be identical on every rank"
    assert np.array_equal(r0["sizes"], r1["sizes"]) and np.array_equal(r0["ranges"], r1["ranges"])
    # single-process oracle replay of the same sequence
    L = cc.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    x, _, ts = cc.synth.make_sequence(48, world=w, beams=16, azim=450)
    P = x.shape[1]
    ores, _, odesc = oracle.run_sequence(x.numpy().reshape(-1, 4), np.arange(49, dtype=np.int64) * P, ts,
                                         np.arange(48, dtype=np.int32), dcfg=dcfg, want_desc=True)
    from parity import compare_desc
    got = r0["desc"].view(L.scan_desc_dt).reshape(-1)
    for i in range(48):
        assert not compare_desc(odesc[i], got[i], float_exact=True)
    for r in (r0, r1):
        res = r["res"].view(L.query_result_dt).reshape(-1)
        for k, qi in enumerate(r["qs"]):
            for f in ["n_res", "cand_gidx", "cand_aft_check3", "n_knn_hits"]:
                assert ores[f][qi] == res[f][k], (qi, f)
