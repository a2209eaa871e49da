"""The C++ CandidateManager mirror (hostcpp/cont2/contour_db.h) in the single-pair flow (hostcpp/examples/pair_demo.cpp,
the reference's test/kitti_read_bin_test.cpp:226-291) vs the oracle.  CPU variant: the program is linked against the CPU
execution harness of the product TU (tests/emu, same C-ABI); the GPU variant links the product library."""
import os
import subprocess

import numpy as np
import pytest

import emu_api
from test_emu_hints import _demo_hints

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contour-context_amd")


def _pair(cc, oracle, device=None):
    L = oracle.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 64
    kw = {"device": device} if device else {}
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450, **kw)
    xs = x.cpu().numpy()
    P = xs.shape[1]
    ores, _, odesc = oracle.run_sequence(xs.reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P, ts, np.arange(n, dtype=np.int32),
                                         dcfg=dcfg, want_desc=True)
    qi = int(np.nonzero(ores["n_res"] > 0)[0][0])
    return xs, odesc, qi, int(ores["cand_gidx"][qi]), dcfg


def _run_and_compare(cc, oracle, exe, tmp_path, device=None):
    L = oracle.L
    xs, odesc, qi, c, dcfg = _pair(cc, oracle, device)
    old, new = tmp_path / "old.bin", tmp_path / "new.bin"
    xs[c].astype(np.float32).tofile(old)
    xs[qi].astype(np.float32).tofile(new)
    out = subprocess.check_output([exe, str(old), str(new), "5"], text=True)
    hl = [[int(v) for v in l.split()[1:]] for l in out.split("\n") if l.startswith("H ")]
    rl = [l.split()[1:] for l in out.split("\n") if l.startswith("R ")]
    assert len(rl) == 1
    hints = _demo_hints(L, odesc, qi, [c])
    eres, esc = oracle.check_hints(oracle.Scan.from_desc(odesc[qi], int_id=1), [oracle.Scan.from_desc(odesc[c], int_id=0)], hints,
                                   sim=dcfg.cont_sim, max_fine_opt=5)
    assert len(hl) == len(hints)
    for got, h, s in zip(hl, hints, esc):
        assert got[:3] == list(h[1:]) and got[3:] == list(s[:5]), (got, h, s)
    r = rl[0]
    assert int(r[0]) == eres["n_res"] == 1
    assert abs(float(r[1]) - eres["correlation"]) < 1e-6
    assert np.abs(np.array([float(v) for v in r[2:5]]) - eres["tf"]).max() < 1e-5
    # sensor-frame pose: ConstellCorrelation::getEstSensTF of the BEV-frame result
    th = eres["tf"][2]
    ox = oy = 74.5
    sens = [np.cos(th) * ox - np.sin(th) * oy + eres["tf"][0] - ox, np.sin(th) * ox + np.cos(th) * oy + eres["tf"][1] - oy, th]
    assert np.abs(np.array([float(v) for v in r[5:8]]) - sens).max() < 1e-5


def test_pair_demo_on_cpu_harness(cc, oracle, tmp_path):
    emu_so = emu_api.build()
    exe = str(tmp_path / "pair_demo_emu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(PKG, "hostcpp", "examples", "pair_demo.cpp"),
                           "-I", os.path.join(PKG, "hostcpp"), "-L", os.path.dirname(emu_so), "-lcc_emu",
                           "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
    _run_and_compare(cc, oracle, exe, tmp_path)


@pytest.mark.gpu
def test_pair_demo_on_gpu(cc, oracle, tmp_path):
    exe = str(tmp_path / "pair_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(PKG, "hostcpp", "examples", "pair_demo.cpp"),
                           "-I", os.path.join(PKG, "hostcpp"), "-L", PKG, "-lcont2_amd", "-Wl,-rpath," + PKG,
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    _run_and_compare(cc, oracle, exe, tmp_path, device="cuda")
