"""cc_scan_ingest_batch / cc_db_add_scan_batch / cc_db_query_scan_batch_submit (the per-scan loop's calls, a few scans at a
time: include/cont2_amd.h) against the same calls made one by one -- tests/scan_batch_check.cpp, on the CPU harness here and
on the GPU through libcont2_amd.so.  Descriptors and query results must agree byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import emu_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contour-context_amd")
SRC = os.path.join(ROOT, "tests", "scan_batch_check.cpp")


def _files(cc, tmp_path, n, beams, azim, device=None):
    kw = {"device": device} if device else {}
    x, poses, ts = cc.synth.make_sequence(n, world=cc.synth.World(loop_len=40.0), beams=beams, azim=azim, **kw)
    xs = x.cpu().numpy()
    paths = []
    for i in range(n):
        p = tmp_path / ("%06d.bin" % i)
        if i in (7, 8, 30):   # scans with more than CC_MAXC components on a level: the batch's slow path (cc_k_contours_big) runs for them
            from test_emu_ingest import _blob_scene
            _blob_scene(i).tofile(p)
        else:
            xs[i].astype(np.float32).tofile(p)
        paths.append(str(p))
    return paths


def test_batched_scan_calls_on_the_cpu_harness(cc, tmp_path):
    emu_so = emu_api.build()
    exe = str(tmp_path / "scan_batch_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", SRC, "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(emu_so), "-lcc_emu",
                           "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
    paths = _files(cc, tmp_path, 56, 16, 450)
    env = dict(os.environ, CC_B1_GRID="6", CC_B2_GRID="6", CC_GMM_GRID="6")
    r = subprocess.run([exe, "4.0"] + paths, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    tag, n, hits = r.stdout.split()[-3:]
    assert tag == "ok" and int(n) == 56 and int(hits) > 0, r.stdout[-300:]   # the drive closes loops: the answers are not all empty


@pytest.mark.gpu
def test_batched_scan_calls_on_the_gpu(cc, tmp_path):
    exe = str(tmp_path / "scan_batch_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", SRC, "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lcont2_amd",
                           "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    paths = _files(cc, tmp_path, 160, 64, 1875, device="cuda")
    r = subprocess.run([exe, "1.0"] + paths, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    tag, n, hits = r.stdout.split()[-3:]
    assert tag == "ok" and int(n) == 160 and int(hits) > 0, r.stdout[-300:]
