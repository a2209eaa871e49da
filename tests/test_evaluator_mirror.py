"""ContLCDEvaluator mirror (contour-context_amd/hostcpp/eval/evaluator.h) pinned on the result files the reference
ships: replaying the (tgt, src, correlation) columns of results/outcome-kitti08.txt against the KITTI-08 ground-truth
poses with the shipped correlation_thres (config/batch_bin_test_config.yaml:66) must reproduce the file's TP/FP/TN/FN
column, its pair and correlation columns and its row format; evalMetricEst is checked against an independent numpy
restatement of the formula.  Host-only code: runs on CPU (the library is linked for its two host helpers)."""
import gzip
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SIM_THRES = 0.64928


def _build(cc, tmp_path):
    pkg = os.path.join(ROOT, "contour-context_amd")
    cc.build()
    exe = str(tmp_path / "eval_replay")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(pkg, "hostcpp", "examples", "eval_replay.cpp"),
                           "-I", os.path.join(pkg, "hostcpp"), "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lcont2_amd",
                           "-Wl,-rpath," + pkg, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def _yaw_only_err(T_est, src, tgt):
    """numpy restatement of correlation.h:241-280 for poses given as 3x4 matrices; T_est = (x, y, theta) in BEV pixels"""
    def hom(p):
        m = np.eye(4)
        m[:3, :4] = p
        return m
    rel = np.linalg.inv(hom(tgt)) @ hom(src)
    z1 = rel[:3, 2]
    ax = np.cross([0, 0, 1.0], z1)
    n = np.linalg.norm(ax)
    ax = ax / n if n > 0 else ax
    ang = -np.arccos(z1[2])
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    D = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    Rr = D @ rel[:3, :3]
    yaw = np.arctan2(Rr[1, 0], Rr[0, 0])
    gt = np.array([[np.cos(yaw), -np.sin(yaw), rel[0, 3]], [np.sin(yaw), np.cos(yaw), rel[1, 3]], [0, 0, 1]])
    o = np.array([74.5, 74.5])   # n_row/2 - 0.5
    c, s = np.cos(T_est[2]), np.sin(T_est[2])
    Rd = np.array([[c, -s], [s, c]])
    t = Rd @ o + np.array(T_est[:2]) - o
    est = np.eye(3)
    est[:2, :2] = Rd
    est[:2, 2] = t
    e = np.linalg.inv(gt) @ est
    return e[0, 2], e[1, 2], np.arctan2(e[1, 0], e[0, 0])


def test_replay_reproduces_shipped_outcome(cc, tmp_path):
    exe = _build(cc, tmp_path)
    poses = tmp_path / "poses.txt"
    with gzip.open(os.path.join(GOLD, "ts-sens_pose-kitti08.txt.gz"), "rt") as f:
        pose_lines = [l for l in f.read().split("\n") if l.strip()]
    poses.write_text("\n".join(pose_lines) + "\n")
    with gzip.open(os.path.join(GOLD, "outcome-kitti08.txt.gz"), "rt") as f:
        rows = [l.rstrip("\n").split("\t") for l in f if l.strip()]
    # scan list: KITTI scan i has the i-th pose's stamp (scripts/gen_batch_bin_configs.py writes both from times.txt)
    scans = tmp_path / "scans.txt"
    with open(scans, "w") as f:
        for i, l in enumerate(pose_lines):
            f.write("%s %d /data/kitti/2_dataset/08/velodyne/%06d.bin\n" % (l.split()[0], i, i))
    rng = np.random.default_rng(0)
    preds = tmp_path / "pred.txt"
    est = []
    with open(preds, "w") as f:
        for r in rows:
            a, b = r[1].split("-")
            T = (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-0.5, 0.5))
            est.append(T)
            f.write("%s %s %s %.17g %.17g %.17g\n" % (a, "-1" if b == "x" else b, r[2], *T))
    out = tmp_path / "outcome.txt"
    log = subprocess.check_output([exe, str(poses), str(scans), repr(SIM_THRES), str(preds), str(out)], text=True)
    got = [l.rstrip("\n").split("\t") for l in open(out) if l.strip()]
    assert len(got) == len(rows) == 4071
    P = np.array([[float(v) for v in l.split()[1:]] for l in pose_lines]).reshape(-1, 3, 4)
    n_pos = 0
    for k, (g, r) in enumerate(zip(got, rows)):
        assert g[0] == r[0], "row %d: tfpn %s, the reference wrote %s (%s)" % (k, g[0], r[0], r[1])
        assert g[1] == r[1] and g[6] == r[6] and g[7] == r[7], (k, g, r)
        assert abs(float(g[2]) - float(r[2])) <= 1e-6 * max(1.0, abs(float(r[2])))
        a, b = r[1].split("-")
        if b != "x" and n_pos < 60:
            n_pos += 1
            ex, ey, et = _yaw_only_err(est[k], P[int(b)], P[int(a)])
            # the file carries 6 significant digits
            assert np.allclose([float(v) for v in g[3:6]], [ex, ey, et], rtol=6e-5, atol=1e-6), (k, g[3:6], (ex, ey, et))
        if b == "x":
            assert g[3:6] == ["0", "0", "0"]
    assert "Found 4071 laser scans with gt poses." in log
    assert {r[0] for r in got} == {"0", "1", "2", "3"}


def test_python_evaluator_reproduces_shipped_outcome(cc, tmp_path):
    """the numpy twin (contour-context_amd/evaluator.py) on the same replay: same labels, same rows"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "contour-context_amd"))
    import evaluator as E
    poses = tmp_path / "poses.txt"
    with gzip.open(os.path.join(GOLD, "ts-sens_pose-kitti08.txt.gz"), "rt") as f:
        pose_lines = [l for l in f.read().split("\n") if l.strip()]
    poses.write_text("\n".join(pose_lines) + "\n")
    with gzip.open(os.path.join(GOLD, "outcome-kitti08.txt.gz"), "rt") as f:
        rows = [l.rstrip("\n").split("\t") for l in f if l.strip()]
    scans = tmp_path / "scans.txt"
    with open(scans, "w") as f:
        for i, l in enumerate(pose_lines):
            f.write("%s %d /data/kitti/2_dataset/08/velodyne/%06d.bin\n" % (l.split()[0], i, i))
    ev = E.ContLCDEvaluator(str(poses), str(scans), SIM_THRES)
    assert len(ev.scans) == 4071
    P = np.array([[float(v) for v in l.split()[1:]] for l in pose_lines]).reshape(-1, 3, 4)
    rng = np.random.default_rng(1)
    for k, r in enumerate(rows):
        a, b = r[1].split("-")
        T = (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-0.5, 0.5))
        rec = ev.add_prediction(int(a), float(r[2]), None if b == "x" else int(b), T)
        assert rec["tfpn"] == int(r[0]), (k, r[:3], rec)
        if b != "x" and k % 7 == 0:
            # the evaluator re-orthonormalises the 6-digit pose matrices through a quaternion, the check does not
            assert np.allclose(rec["err"], _yaw_only_err(T, P[int(b)], P[int(a)]), rtol=2e-5, atol=3e-4)
    out = tmp_path / "outcome_py.txt"
    ev.save_prediction_results(str(out))
    got = [l.rstrip("\n").split("\t") for l in open(out)]
    for g, r in zip(got, rows):
        assert g[0] == r[0] and g[1] == r[1] and g[6] == r[6] and g[7] == r[7]
        assert abs(float(g[2]) - float(r[2])) <= 1e-6 * max(1.0, abs(float(r[2])))


def test_load_check_thres_reads_the_reference_threshold_files(cc, tmp_path):
    """ContLCDEvaluator::loadCheckThres (evaluator.h:436, src/eval/evaluator.cpp:7-64): `name lower upper` per line, `#`
    comments, blank lines and unknown names skipped -- the format of the reference's config/score_thres_*.cfg."""
    import emu_api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "contour-context_amd")
    emu_so = emu_api.build()
    exe = str(tmp_path / "thr")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tests", "load_check_thres_check.cpp"), "-I", os.path.join(pkg, "hostcpp"),
                           "-I", os.path.join(root, "include"), "-L", os.path.dirname(emu_so), "-lcc_emu", "-Wl,-rpath," + os.path.dirname(emu_so),
                           "-pthread", "-o", exe])
    cfg = tmp_path / "thres.cfg"
    cfg.write_text("i_ovlp_sum          4       6\ni_ovlp_max_one      3       7\ni_in_ang_rng        4       6\n\ni_indiv_sim         5       6\n"
                   "i_orie_sim          4       8\n# f_area_perc         5       10\nunknown_name 1 2\ncorrelation         0.40    0.75\n"
                   "area_perc           0.03    0.15\nneg_est_dist        -5.01    -5.0\n")
    out = subprocess.check_output([exe, str(cfg)], text=True)
    assert "RES 4 3 4 5 4 0.4000 0.0300 -5.0100 | 6 7 6 6 8 0.7500 0.1500 -5.0000" in out
    assert out.splitlines()[0] == "i_ovlp_sum" and "#" in out          # names are echoed as they are read, comments too
