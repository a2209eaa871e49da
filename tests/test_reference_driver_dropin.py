"""the reference's own offline driver, test/batch_bin_test.cpp, read IN PLACE from /root/reference
(never copied), compiles UNCHANGED against the class mirror (hostcpp/) with PUB_ROS_MSG=0 -- and, linked against the CPU
execution harness of the product's translation unit (same C-ABI as libcont2_amd.so), runs a short sequence end to end:
YAML -> ContLCDEvaluator -> ContourManager / ContourDB -> outcome file, equal to the oracle's replay of the same loop.
Build-container test: the reference tree does not exist on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

import emu_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "contour-context_amd")
REF_DRIVER = "/root/reference/test/batch_bin_test.cpp"

pytestmark = pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="reference tree not present")


@pytest.mark.parametrize("save_mid_file", [0, 1])  # 1: the driver also calls saveContourImage / saveMatchedPairImg
def test_reference_driver_compiles_unchanged(save_mid_file):
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DPUB_ROS_MSG=0", "-DSAVE_MID_FILE=%d" % save_mid_file, '-DPJSRCDIR="/tmp"',
                        "-I", os.path.join(PKG, "hostcpp"), "-I", os.path.join(ROOT, "include"), REF_DRIVER],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_reference_driver_runs_on_the_cpu_harness(cc, oracle, tmp_path):
    emu_so = emu_api.build()
    proj = tmp_path / "proj"
    (proj / "config").mkdir(parents=True)
    (proj / "log").mkdir()
    exe = str(tmp_path / "ref_batch_bin_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DPUB_ROS_MSG=0", "-DSAVE_MID_FILE=0", '-DPJSRCDIR="%s"' % proj,
                           REF_DRIVER, "-I", os.path.join(PKG, "hostcpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.dirname(emu_so), "-lcc_emu", "-Wl,-rpath," + os.path.dirname(emu_so), "-pthread", "-o", exe])
    w = cc.synth.World(loop_len=40.0)
    n = 52
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    ts = ts * 4.0  # 0.4 s per scan: a 40-scan lap takes 16 s, past the evaluator's 15 s exclusion window
    xs = x.numpy()
    # ragged scans: the first ones hold 55 % of their points, later ones up to all of them -- the evaluator mirror sizes its pinned
    # staging buffers by the files, so they have to GROW while batches of scans are in flight (hostcpp/eval/evaluator.h)
    rng = np.random.default_rng(5)
    scans = []
    for i in range(n):
        frac = 0.55 if i < 6 else (1.0 if i in (20, 37) else rng.uniform(0.7, 0.95))
        keep = np.sort(rng.choice(xs.shape[1], size=int(frac * xs.shape[1]), replace=False))
        scans.append(np.ascontiguousarray(xs[i][keep]).astype(np.float32))
    lst, pos = tmp_path / "scans.txt", tmp_path / "poses.txt"
    with open(lst, "w") as f, open(pos, "w") as g:
        for i in range(n):
            p = tmp_path / ("%06d.bin" % i)
            scans[i].tofile(p)
            f.write("%.6f %d %s\n" % (ts[i], i, p))
            c, s_ = np.cos(poses[i, 2]), np.sin(poses[i, 2])
            g.write("%.6f %.9f %.9f 0 %.9f %.9f %.9f 0 %.9f 0 0 1 0\n" % (ts[i], c, -s_, poses[i, 0], s_, c, poses[i, 1]))
    cfg = open(os.path.join(PKG, "hostcpp", "examples", "batch_bin_test_config.yaml")).read()
    cfg = cfg.replace("/path/to/ts-sens_pose-kitti08.txt", str(pos)).replace("/path/to/ts-lidar_bins-kitti08.txt", str(lst))
    cfg = cfg.replace("/path/to/outcome-kitti08.txt", str(tmp_path / "outcome.txt"))
    cfg = cfg.replace("max_elapse_: 25.0", "max_elapse_: 10.0").replace("min_elapse_: 15.0", "min_elapse_: 6.0")
    (proj / "config" / "batch_bin_test_config.yaml").write_text(cfg)
    env = dict(os.environ, CC_B1_GRID="6", CC_B2_GRID="6", CC_GMM_GRID="6", CC_STP_DEVICE_TIMERS="1", CC_EVAL_TIMERS="1")  # device stage timers: on request; the database's read-ahead: on by default
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    # with CC_DB_READ_AHEAD the database mirror works ahead of this unchanged driver: answers were queued before the driver
    # asked, none of it had to be undone (hostcpp/cont2/contour_db.h "read-ahead of the database")
    ra = [l for l in out.stderr.splitlines() if l.startswith("[ContourDB read-ahead]")]
    assert ra, out.stderr[-1500:]
    hit, miss, rebuilds = [int(v) for v in __import__("re").findall(r"(\d+)", ra[-1])][-3:]
    assert hit + miss == n and hit > 0 and rebuilds == 0, ra[-1]   # (how many: depends on how far ahead the read-ahead thread gets on this machine)
    rows = [l.rstrip("\n").split("\t") for l in open(tmp_path / "outcome.txt")]
    assert len(rows) == n
    dcfg = cc.L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 10.0, 6.0
    offs = np.concatenate([[0], np.cumsum([len(a) for a in scans])]).astype(np.int64)
    ores, _, _ = oracle.run_sequence(np.concatenate(scans), offs, ts, np.arange(n, dtype=np.int32), dcfg=dcfg)
    assert (ores["n_res"] > 0).sum() >= 3
    for i, r in enumerate(rows):
        a, b = r[1].split("-")
        assert int(a) == i
        assert (b == "x") == (ores["n_res"][i] == 0), (i, r)
        if b != "x":
            assert int(b) == ores["cand_gidx"][i]
            assert abs(float(r[2]) - ores["correlation"][i]) < 1e-5
    # the library's stage timers went through the executable's `stp` (the five reference stage names)
    timing = (proj / "log" / "timing_cont2.txt").read_text()
    for name in ("make bev", "KNN search", "Constell", "L2 opt", "Update database", "queryRangedKNN (wall)"):
        assert name in timing, timing
