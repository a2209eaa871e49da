"""The library-owned collective (include/cont2_amd.h: cc_comm_*) on the hardware that exists: ONE GPU, i.e.
RCCL with a world of one -- dlopen of librccl, ncclGetUniqueId, ncclCommInitRank, ncclAllGather and the C++ driver on top
(hostcpp/examples/batch_replay_mgpu.cpp) really run; what more than one rank adds (the id file under /dev/shm, xGMI traffic)
cannot be exercised here and is said to be unmeasured wherever it is described.  (One test: an RCCL communicator takes the better
part of a minute to come up on the build boxes.  The ctypes route to the same calls -- contour-context_amd/sharding.py:
gather_records_c, `bench.py --comm-owner c` -- was run by hand: profiles/r5/bench_comm_owner_c_world1.txt.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_multi_gpu_driver_world_of_one(cc, tmp_path):
    """batch_replay_mgpu (C++, C-ABI only): scan-sharded ingest -> pack -> cc_comm_allgather_packed -> cc_db_add_packed ->
    query-sharded scoring at each scan's own epoch, against the same flow through the Python host."""
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    exe = G.build_mgpu_driver()
    L = cc.L
    w = cc.synth.World(loop_len=40.0)
    n = 72
    xyzi, _, ts = cc.synth.make_sequence(n, world=w, device="cuda", beams=32, azim=900)
    P = xyzi.shape[1]
    xh = xyzi.cpu().numpy()
    ts = np.asarray(ts, np.float64) * 10.0   # the shipped DB delays (15 / 25 s) against a 40-m loop driven in a few seconds
    lst = tmp_path / "scans.txt"
    with open(lst, "w") as f:
        for i in range(n):
            p = tmp_path / ("%06d.bin" % i)
            xh[i].astype(np.float32).tofile(p)
            f.write("%.6f %s\n" % (ts[i], p))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # PyTorch's own librccl is in the page cache already (this process imported torch); the system's copy under /opt/rocm is
    # several hundred MB that a fresh box pages in for the better part of a minute
    torch_rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(torch_rccl):
        env["CC_RCCL_LIB"] = torch_rccl
    r = subprocess.run([exe, str(lst), str(tmp_path / "out")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0 of 1" in r.stdout
    got = np.loadtxt(tmp_path / "out.rank0.txt").reshape(n, 6)
    ctx = cc.Context(0, max_batch=n)
    desc = ctx.ingest(xyzi.reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P)
    db = cc.Database(ctx, capacity=n + 16)
    db.add_scans(desc, ts, np.arange(n, dtype=np.int32))
    res = db.query(desc, np.arange(n, dtype=np.int32))
    exp = np.where(res["n_res"] > 0, res["cand_gidx"], -1)
    assert np.array_equal(got[:, 0].astype(int), np.arange(n)) and np.array_equal(got[:, 1].astype(int), exp)
    assert (exp >= 0).sum() > 0, "the sequence should close loops"
    m = exp >= 0
    assert np.abs(got[m, 2] - res["correlation"][m]).max() < 1e-5 and np.abs(got[m, 3:6] - res["tf"][m]).max() < 1e-4
    db.close()
    ctx.close()
