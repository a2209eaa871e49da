"""The class mirror's two read-ahead mechanisms (mirror-only; the reference's loop is strictly sequential):
  * hostcpp/eval/evaluator.h reads and ingests the next scans on helper threads, a batch at a time (cc_scan_ingest_batch);
  * hostcpp/cont2/contour_db.h appends the published scans and queues their queries ahead of the driver, a batch per step
    (cc_db_add_scan_batch / cc_db_query_scan_batch_submit), and validates the driver's calls against that work.
Whatever the driver does, every descriptor and every answer must be the one the sequential path gives.  On the CPU harness
here, and on the GPU through libcont2_amd.so."""
import os
import re
import subprocess

import numpy as np
import pytest

import emu_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contour-context_amd")


def _build(tmp_path, src, name, gpu):
    exe = str(tmp_path / name)
    if gpu:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", src), "-I", os.path.join(PKG, "hostcpp"),
                               "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lcont2_amd", "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-lamdhip64",
                               "-o", exe])
    else:
        emu_so = emu_api.build()
        subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", src), "-I", os.path.join(PKG, "hostcpp"),
                               "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(emu_so), "-lcc_emu", "-Wl,-rpath," + os.path.dirname(emu_so),
                               "-pthread", "-o", exe])
    return exe


def _lists(cc, tmp_path, n, beams, azim, ts_scale, device=None):
    kw = {"device": device} if device else {}
    x, poses, ts = cc.synth.make_sequence(n, world=cc.synth.World(loop_len=40.0), beams=beams, azim=azim, **kw)
    ts = ts * ts_scale
    xs = x.cpu().numpy()
    lst, pos = tmp_path / "scans.txt", tmp_path / "poses.txt"
    with open(lst, "w") as f, open(pos, "w") as g:
        for i in range(n):
            p = tmp_path / ("%06d.bin" % i)
            xs[i].astype(np.float32).tofile(p)
            f.write("%.6f %d %s\n" % (ts[i], i, p))
            g.write("%.6f 1 0 0 %.9f 0 1 0 %.9f 0 0 1 0\n" % (ts[i], poses[i, 0], poses[i, 1]))
    return lst, pos


def _read_ahead_any_driver(cc, tmp_path, mode, gpu):
    """tests/db_read_ahead_check.cpp: the reference's loop, repeated queries with other thresholds, scans that are never added, a
    jump in the scan list, two evaluators feeding two databases in turn -- with the mirror's read-ahead in steps of several scans, one scan per step, and off
    (CC_DB_READ_AHEAD=0) every answer is the same."""
    exe = _build(tmp_path, "db_read_ahead_check.cpp", "db_read_ahead_check", gpu)
    n = 96 if gpu else 48
    lst, pos = _lists(cc, tmp_path, n, 64 if gpu else 16, 1875 if gpu else 450, 4.0, "cuda" if gpu else None)
    outs = []
    # default: 16 deep, steps of eight (evaluator: 32 ahead, ingest batches of eight) | off | depth 8: steps of four | depth 3: one scan per step
    cfgs = [{}, {"CC_DB_READ_AHEAD": "0", "CC_EVAL_AHEAD": "4", "CC_EVAL_INGEST_BATCH": "1"}, {"CC_DB_READ_AHEAD": "8", "CC_EVAL_AHEAD": "12"},
            {"CC_DB_READ_AHEAD": "3", "CC_EVAL_AHEAD": "4"}]
    if gpu:
        cfgs = cfgs[:3]   # (the one-scan-per-step depth is covered on the CPU harness)
    for env_ra in cfgs:
        env = dict(os.environ, CC_EVAL_TIMERS="1", **env_ra)
        if not gpu:
            env.update(CC_B1_GRID="6", CC_B2_GRID="6", CC_GMM_GRID="6")
        r = subprocess.run([exe, str(pos), str(lst), str(mode)], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
        outs.append(([l for l in r.stdout.splitlines() if l[:1] in "qtd" and not l.startswith("===")], r.stderr))
    for o in outs[1:]:
        assert o[0] == outs[0][0]
    assert outs[0][0][-1].startswith("done")
    assert any(l.split()[1] != "-1" for l in outs[0][0] if l.startswith("q")), "the sequence should close loops"
    for k in (0, 2):
        steps = [l for l in outs[k][1].splitlines() if l.startswith("[ContourDB read-ahead, mean")]
        # some steps took several scans (how many depends on how far the helper threads got on this machine)
        assert steps and max(float(re.search(r"steps of ([0-9.]+) scans", l).group(1)) for l in steps) > 1.0, steps
        ra = [l for l in outs[k][1].splitlines() if l.startswith("[ContourDB read-ahead]")]
        hit, miss, rebuilds = [int(v) for v in re.findall(r"(\d+)", ra[-1])][-3:]
        assert hit > 0, ra[-1]
        if mode in (2, 3):
            assert rebuilds > 0, ra[-1]   # the driver left the predicted sequence: the device database was rebuilt
        if mode == 4:   # two sources, two databases: each follows its own source, nothing has to be undone (both databases print a line)
            assert all(int(re.findall(r"(\d+)", l)[-1]) == 0 for l in ra), ra


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_database_read_ahead_survives_any_driver(cc, tmp_path, mode):
    _read_ahead_any_driver(cc, tmp_path, mode, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_database_read_ahead_survives_any_driver_on_the_gpu(cc, tmp_path, mode):
    _read_ahead_any_driver(cc, tmp_path, mode, gpu=True)


def _prefetch_paths(cc, tmp_path, gpu):
    """hostcpp/eval/evaluator.h reads and ingests scans ahead of the driver in batches; a scan asked for twice, or with the image
    switch flipped in between, goes the direct way -- same descriptors either way (tests/evaluator_prefetch_check.cpp)."""
    exe = _build(tmp_path, "evaluator_prefetch_check.cpp", "prefetch_check", gpu)
    n = 40 if gpu else 9
    lst, pos = _lists(cc, tmp_path, n, 64 if gpu else 16, 1875 if gpu else 450, 1.0, "cuda" if gpu else None)
    out = subprocess.run([exe, str(pos), str(lst)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and ("OK %d" % n) in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_evaluator_read_ahead_paths_give_the_same_descriptors(cc, tmp_path):
    _prefetch_paths(cc, tmp_path, gpu=False)


@pytest.mark.gpu
def test_evaluator_read_ahead_paths_give_the_same_descriptors_on_the_gpu(cc, tmp_path):
    _prefetch_paths(cc, tmp_path, gpu=True)
