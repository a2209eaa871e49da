"""Randomised campaign of the class mirror's read-ahead on the CPU harness (run by hand):
    python tests/fuzz_emu_read_ahead.py <seed0> <n_iter> [gpu]     ("gpu": through libcont2_amd.so on the device, full-size scans)
tests/db_read_ahead_check.cpp with a RANDOM driver (mode = 100 + seed: second queries with other thresholds, scans that are
not added, adds with an unexpected seed or time stamp, jumps back in the scan list) on a random drive: the answers with the
read-ahead in steps of several scans, one scan per step, and off must be the same lines."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "contour-context_amd")
sys.path[:0] = [ROOT, HERE]
import cc_amd  # noqa: E402
import emu_api  # noqa: E402


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    cc = cc_amd.load()
    tmp = tempfile.mkdtemp(prefix="cc_fuzz_ra_")
    exe = os.path.join(tmp, "db_read_ahead_check")
    if gpu:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "db_read_ahead_check.cpp"), "-I", os.path.join(PKG, "hostcpp"),
                               "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lcont2_amd", "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    else:
        emu_so = emu_api.build()
        subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(HERE, "db_read_ahead_check.cpp"), "-I", os.path.join(PKG, "hostcpp"), "-I",
                               os.path.join(ROOT, "include"), "-L", os.path.dirname(emu_so), "-lcc_emu", "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread",
                               "-o", exe])
    bad = 0
    for it in range(n_iter):
        seed = seed0 + it
        rng = np.random.default_rng(seed)
        n = int(rng.integers(120, 200)) if gpu else int(rng.integers(36, 60))
        w = cc.synth.World(loop_len=float(rng.uniform(24, 44)), dense=bool(rng.integers(2)), seed=int(rng.integers(1 << 20)))
        if gpu:
            x, poses, ts = cc.synth.make_sequence(n, world=w, beams=64, azim=1875, device="cuda")
        else:
            x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
        ts = ts * float(rng.choice([1.0, 4.0]))
        xs = x.cpu().numpy()
        lst, pos = os.path.join(tmp, "scans.txt"), os.path.join(tmp, "poses.txt")
        with open(lst, "w") as f, open(pos, "w") as g:
            for i in range(n):
                p = os.path.join(tmp, "%06d.bin" % i)
                xs[i].astype(np.float32).tofile(p)
                f.write("%.6f %d %s\n" % (ts[i], i, p))
                g.write("%.6f 1 0 0 %.9f 0 1 0 %.9f 0 0 1 0\n" % (ts[i], poses[i, 0], poses[i, 1]))
        outs = []
        for env_ra in ({}, {"CC_DB_READ_AHEAD": "0", "CC_EVAL_AHEAD": "4", "CC_EVAL_INGEST_BATCH": "1"}, {"CC_DB_READ_AHEAD": "3", "CC_EVAL_AHEAD": "6", "CC_EVAL_INGEST_BATCH": "3"}):
            env = dict(os.environ, CC_EVAL_TIMERS="1", **env_ra)
            if not gpu:
                env.update(CC_B1_GRID="6", CC_B2_GRID="6", CC_GMM_GRID="6")
            r = subprocess.run([exe, pos, lst, str(100 + seed)], env=env, capture_output=True, text=True, timeout=3000)
            if r.returncode != 0:
                outs.append(["crash %d: %s" % (r.returncode, r.stderr[-400:])])
            else:
                outs.append([l for l in r.stdout.splitlines() if l[:1] in "qtd" and not l.startswith("===")])
            last = [l for l in r.stderr.splitlines() if l.startswith("[ContourDB read-ahead]")]
        ok = outs[0] == outs[1] == outs[2] and outs[0] and outs[0][-1].startswith("done")
        bad += 0 if ok else 1
        print("seed %d scans %d: %s  (%d answers; %s)" % (seed, n, "ok" if ok else "DIFFER", len(outs[0]), last[-1][23:] if last else ""), flush=True)
        if not ok:
            for a, b in zip(outs[0], outs[1]):
                if a != b:
                    print("   first difference: %r | %r" % (a, b))
                    break
    print("done: %d of %d drives differ" % (bad, n_iter))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
