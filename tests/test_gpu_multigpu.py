"""The RCCL path on real GPUs: runs only where at least two devices are visible (the single-GPU test boxes skip it; the
first multi-GPU node runs it).  `bench.py --gpus 2` launches its own ranks under torch.distributed.run, builds the replicated
DB through the sharding helpers (interleaved shards, one all_gather_into_tensor of the packed records over xGMI) and times
the weak-scaling step; with --share-descriptors every step also all-gathers the batch's packed records.  The CPU twin of
this test is tests/test_distributed_gloo.py (same helpers, gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--db-scans", "1664", "--batch", "256", "--steps", "2",
           "--warmup", "1", "--no-cpu", "--no-extra", "--workload", "sparse"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_nccl_bench_path():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    d = _run([])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cc_amd
    from mgpu_checks import check_multi_gpu_line
    cc = cc_amd.load()
    # record sizes from the library, the same expression the gloo twin asserts (tests/test_distributed_gloo.py)
    check_multi_gpu_line(d, 2, 1664, 256, cc.packed_sizes(), False, "nccl", cc.sharding.shard_len)
    # the DB covers the whole 1.5 km loop, so every query revisits a DB place: rank 0's replica must close (nearly) all loops
    found = int(d["config"]["workload"].split("loop closures found: ")[1].split(" of")[0])
    assert found >= 480, d["config"]["workload"]
    d2 = _run(["--share-descriptors"])
    check_multi_gpu_line(d2, 2, 1664, 256, cc.packed_sizes(), True, "nccl", cc.sharding.shard_len)
