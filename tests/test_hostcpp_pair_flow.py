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


def _read_png_gray8(path):
    """decode the 8-bit grey, non-interlaced PNG the mirror writes (any zlib stream, filter type 0 rows)"""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        body = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0], "chunk CRC"
        if typ == b"IHDR":
            w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, comp, flt, inter) == (8, 0, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:]


def _contour_image(bev, thr):
    """ContourManager::getContourImage (contour_mng.h:1041-1049) in numpy: THRESH_TOZERO, then NORM_MINMAX to 0..255 as
    convertTo does it for a float source (f32 multiply-add, round half to even, saturate)"""
    m = np.where(bev > np.float32(thr), bev, np.float32(0)).astype(np.float32)
    mn, mx = float(m.min()), float(m.max())
    scale = 255.0 * (1.0 / (mx - mn) if mx - mn > np.finfo(np.float64).eps else 0.0)
    a, b = np.float32(scale), np.float32(0.0 - mn * scale)
    return np.clip(np.rint(m * a + b), 0, 255).astype(np.uint8)


def _run_and_compare(cc, oracle, exe, tmp_path, device=None, env=None):
    L = oracle.L
    xs, odesc, qi, c, dcfg = _pair(cc, oracle, device)
    old, new = tmp_path / "old.bin", tmp_path / "new.bin"
    xs[c].astype(np.float32).tofile(old)
    xs[qi].astype(np.float32).tofile(new)
    out = subprocess.check_output([exe, str(old), str(new), "5", str(tmp_path / "img")], text=True, env=env)
    # the SAVE_MID_FILE artefacts: level images of both scans side by side (saveMatchedPairImg, contour_mng.h:1286-1311)
    lv = [1.5, 2.0, 2.5, 3.0, 3.5, 4.0]
    b_old, b_new = oracle.Scan(xs[c]).bev()[0].reshape(150, 150), oracle.Scan(xs[qi]).bev()[0].reshape(150, 150)
    exp = np.full((150 * 2 + 1, 151 * 6), 255, np.uint8)
    for i, t in enumerate(lv):
        exp[0:150, i * 151:i * 151 + 150] = _contour_image(b_old, t)
        exp[151:301, i * 151:i * 151 + 150] = _contour_image(b_new, t)
    assert np.array_equal(_read_png_gray8(tmp_path / "img_pair.png"), exp)
    assert np.array_equal(_read_png_gray8(tmp_path / "img_lv2.png"), _contour_image(b_new, 2.5))
    assert exp[:150, :150].max() == 255 and (exp[:150, :150] > 0).sum() > 50
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
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(PKG, "hostcpp", "examples", "pair_demo.cpp"),
                           "-I", os.path.join(PKG, "hostcpp"), "-L", os.path.dirname(emu_so), "-lcc_emu",
                           "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
    _run_and_compare(cc, oracle, exe, tmp_path, env=dict(os.environ, **emu_api.SMALL_GRIDS))


@pytest.mark.gpu
def test_pair_demo_on_gpu(cc, oracle, tmp_path):
    exe = str(tmp_path / "pair_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(PKG, "hostcpp", "examples", "pair_demo.cpp"),
                           "-I", os.path.join(PKG, "hostcpp"), "-L", PKG, "-lcont2_amd", "-Wl,-rpath," + PKG,
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    _run_and_compare(cc, oracle, exe, tmp_path, device="cuda")
