"""The C++ class mirror (contour-context_amd/hostcpp: ContourManager / ContourDB with the reference's signatures)
driven by a batch_bin_test-shaped program on KITTI-format .bin files, vs the oracle's replay of the same loop."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_bin_demo_matches_oracle(cc, oracle, tmp_path):
    pkg = os.path.join(ROOT, "contour-context_amd")
    exe = str(tmp_path / "batch_bin_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(pkg, "hostcpp", "examples", "batch_bin_demo.cpp"),
                           "-I", os.path.join(pkg, "hostcpp"), "-L", pkg, "-lcont2_amd", "-Wl,-rpath," + pkg,
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    w = cc.synth.World(loop_len=40.0)
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=32, azim=900, device="cuda")
    xs = x.cpu().numpy()
    lst = tmp_path / "bins.txt"
    with open(lst, "w") as f:
        for i in range(n):
            p = tmp_path / ("%06d.bin" % i)
            xs[i].astype(np.float32).tofile(p)
            f.write("%.6f %d %s\n" % (ts[i], i, p))
    out = subprocess.check_output([exe, str(lst), "1.5", "2.5"], text=True)
    rows = [l.split() for l in out.strip().split("\n")]
    assert len(rows) == n
    dcfg = cc.L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    P = xs.shape[1]
    ores, _, _ = oracle.run_sequence(xs.reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P, ts, np.arange(n, dtype=np.int32),
                                     dcfg=dcfg)
    assert (ores["n_res"] > 0).sum() > 5
    for i, r in enumerate(rows):
        assert int(r[0]) == i
        assert int(r[1]) == ores["cand_gidx"][i], (i, r, ores[i])
        if ores["n_res"][i]:
            assert abs(float(r[2]) - ores["correlation"][i]) < 1e-4
            assert np.abs(np.array([float(v) for v in r[3:6]]) - ores["tf"][i]).max() < 1e-4
