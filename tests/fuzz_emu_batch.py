"""Randomised campaign of the batched per-scan calls on the CPU harness (run by hand):
    python tests/fuzz_emu_batch.py <seed0> <n_iter> [gpu]     ("gpu": through libcont2_amd.so on the device, full-size scans)
tests/scan_batch_check.cpp (cc_scan_ingest_batch / cc_db_add_scan_batch / cc_db_query_scan_batch_submit against the same
calls made one by one: descriptors and query results byte for byte) on random worlds and drives: sparse / dense world, lap
length, ragged scans (a random share of every scan's points dropped), a contour-rich scan now and then (the slow path of
K2 inside a batch), the time step between scans (how long keys wait in the buffers before they are searchable)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]
import cc_amd  # noqa: E402
import emu_api  # noqa: E402
from test_emu_ingest import _blob_scene  # noqa: E402


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    cc = cc_amd.load()
    tmp = tempfile.mkdtemp(prefix="cc_fuzz_batch_")
    exe = os.path.join(tmp, "scan_batch_check")
    if gpu:
        pkg = os.path.join(ROOT, "contour-context_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "scan_batch_check.cpp"), "-I", os.path.join(ROOT, "include"), "-L", pkg,
                               "-lcont2_amd", "-Wl,-rpath," + pkg, "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
        env = dict(os.environ)
    else:
        emu_so = emu_api.build()
        subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(HERE, "scan_batch_check.cpp"), "-I", os.path.join(ROOT, "include"), "-L",
                               os.path.dirname(emu_so), "-lcc_emu", "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
        env = dict(os.environ, CC_B1_GRID="6", CC_B2_GRID="6", CC_GMM_GRID="6")
    bad = 0
    for it in range(n_iter):
        seed = seed0 + it
        rng = np.random.default_rng(seed)
        dense = bool(rng.integers(2))
        w = cc.synth.World(loop_len=float(rng.uniform(24, 44)), dense=dense, seed=int(rng.integers(1 << 20)))
        n = int(rng.integers(120, 220)) if gpu else int(rng.integers(40, 72))
        if gpu:
            x, poses, ts = cc.synth.make_sequence(n, world=w, beams=64, azim=1875, device="cuda")
        else:
            x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
        xs = x.cpu().numpy()
        paths = []
        for i in range(n):
            p = os.path.join(tmp, "%06d.bin" % i)
            if rng.random() < 0.04:
                _blob_scene(int(rng.integers(1 << 20))).tofile(p)
            else:
                keep = np.sort(rng.choice(xs.shape[1], size=int(rng.uniform(0.5, 1.0) * xs.shape[1]), replace=False))
                np.ascontiguousarray(xs[i][keep]).astype(np.float32).tofile(p)
            paths.append(p)
        dt = float(rng.choice([0.4, 1.0, 4.0]))
        r = subprocess.run([exe, "%g" % dt] + paths, env=env, capture_output=True, text=True, timeout=3000)
        ok = r.returncode == 0 and r.stdout.split()[-3:-2] == ["ok"]
        bad += 0 if ok else 1
        print("seed %d dense %d scans %d dt %g: %s" % (seed, dense, n, dt, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
    print("done: %d of %d drives differ" % (bad, n_iter))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
