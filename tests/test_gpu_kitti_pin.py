"""PIN ON ARRIVAL.  The only artefacts of the reference that pin OpenCV's component order, Eigen's solvers and Ceres'
L-BFGS together are end-to-end ones and need the KITTI-08 point clouds, which this image does not have:
  * results/outcome_txt/outcome-kitti08.txt (committed as tests/golden/outcome-kitti08.txt.gz): per scan the matched
    scan, the correlation to 6 significant digits and the pose error, e.g. `1648-237 0.789965 0.0154641 0.210156 -0.000339971`;
  * scripts/plot_contours.py:154-156: the BEV transform of the pair 0237 -> 1648;
  * the aggregate numbers of scripts/pr_mpe.py on that file (max-F1 0.955621 at index 1802, 323 true positives).
This test does the whole comparison the moment a KITTI odometry directory is mounted:

    KITTI08_DIR=/data/kitti/dataset python -m pytest tests/test_gpu_kitti_pin.py -m gpu -s

Accepted layouts: <dir>/sequences/08/{velodyne/*.bin,times.txt,calib.txt} (+ <dir>/poses/08.txt), or a flat
<dir>/{velodyne/*.bin,times.txt,calib.txt} (+ poses.txt or 08.txt).  The ground-truth pose file is the one the reference
ships (sample_data/ts-sens_pose-kitti08.txt, committed under tests/golden/); when the dataset's own poses are present
the list generator's output is checked against it too (row P2 of SURVEY.md 8(a)).
Without $KITTI08_DIR the test is skipped -- and `parity` stays "partial" for the unpinned pieces."""
import gzip
import math
import os
import shutil
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_OUT = os.path.join(ROOT, "tests", "golden", "outcome-kitti08.txt.gz")
GOLD_POSE = os.path.join(ROOT, "tests", "golden", "ts-sens_pose-kitti08.txt.gz")
# scripts/plot_contours.py:154-156, T_delta of the BEV frames of scans 0237 (src) and 1648 (tgt)
PLOT_T_DELTA = ((-0.999999, -0.00140861, 145.041), (0.00140861, -0.999999, 149.279))


def _find_layout(d):
    for seq, poses in ((os.path.join(d, "sequences", "08"), os.path.join(d, "poses", "08.txt")), (d, os.path.join(d, "poses.txt")),
                       (d, os.path.join(d, "08.txt"))):
        if os.path.isdir(os.path.join(seq, "velodyne")) and os.path.isfile(os.path.join(seq, "times.txt")):
            return seq, (poses if os.path.isfile(poses) else None)
    return None, None


def test_kitti08_outcome_equals_the_reference(tmp_path):
    d = os.environ.get("KITTI08_DIR")
    if not d or not os.path.isdir(d):
        pytest.skip("KITTI08_DIR is not set: the KITTI-08 point clouds are not in this image (see the module docstring)")
    seq, poses = _find_layout(d)
    assert seq, "KITTI08_DIR has none of the accepted layouts"
    sys.path.insert(0, os.path.join(ROOT, "contour-context_amd", "tools"))
    import batch_eval
    import compare_outcome
    import gen_lists
    pose_txt = str(tmp_path / "ts-sens_pose-kitti08.txt")
    with gzip.open(GOLD_POSE, "rt") as f, open(pose_txt, "w") as g:
        shutil.copyfileobj(f, g)
    lst_txt = str(tmp_path / "ts-lidar_bins-kitti08.txt")
    calib = os.path.join(seq, "calib.txt")
    if poses and os.path.isfile(calib):  # the dataset's own poses through the list generator == the shipped pose file
        own = str(tmp_path / "own_pose.txt")
        gen_lists.gen_kitti(os.path.join(seq, "velodyne"), poses, os.path.join(seq, "times.txt"), calib, own, lst_txt)
        a = np.loadtxt(own)
        b = np.loadtxt(pose_txt)
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-6, "gen_lists.gen_kitti differs from the reference's pose file"
    else:
        times = [float(l) for l in open(os.path.join(seq, "times.txt")) if l.strip()]
        bins = sorted(f for f in os.listdir(os.path.join(seq, "velodyne")) if f.endswith(".bin"))
        with open(lst_txt, "w") as f:
            f.write("\n".join("%.6f %d %s" % (t, i, os.path.join(seq, "velodyne", bins[i])) for i, t in enumerate(times)))
    cfg = open(os.path.join(ROOT, "contour-context_amd", "hostcpp", "examples", "batch_bin_test_config.yaml")).read()
    out_txt = str(tmp_path / "outcome-kitti08.txt")
    cfg = cfg.replace("/path/to/ts-sens_pose-kitti08.txt", pose_txt).replace("/path/to/ts-lidar_bins-kitti08.txt", lst_txt)
    cfg = cfg.replace("/path/to/outcome-kitti08.txt", out_txt)
    (tmp_path / "cfg.yaml").write_text(cfg)
    ev, res, summary = batch_eval.run(str(tmp_path / "cfg.yaml"), chunk=256, verbose=True)
    # ---- the outcome file, row by row, and the aggregate numbers
    r = compare_outcome.compare(out_txt, GOLD_OUT, poses=pose_txt)
    print("\n".join(r["details"]))
    print({k: v for k, v in r.items() if k != "details"})
    assert r["n_rows_ours"] == r["n_rows_ref"] == 4071
    assert r["n_match_diff"] == 0, "a scan is matched with a different scan than in the reference's outcome file"
    assert r["n_label_diff"] == 0 and r["n_corr_diff"] == 0 and r["n_pose_diff"] == 0
    assert abs(r["max_f1_ours"] - 0.955621) < 5e-7 and r["max_f1_idx_ours"] == 1802 and r["tp_count_ours"] == 323
    assert r["ok"]
    # ---- the pair the reference's scripts single out: 0237 -> 1648
    q = res[1648]
    assert q["n_res"] == 1 and q["cand_gidx"] == 237
    th = math.atan2(PLOT_T_DELTA[1][0], PLOT_T_DELTA[0][0])
    assert abs(math.cos(q["tf"][2]) - PLOT_T_DELTA[0][0]) < 2e-6 and abs(math.sin(q["tf"][2]) - PLOT_T_DELTA[1][0]) < 2e-6, (q["tf"], th)
    assert abs(q["tf"][0] - PLOT_T_DELTA[0][2]) < 1e-3 and abs(q["tf"][1] - PLOT_T_DELTA[1][2]) < 1e-3, q["tf"]
    assert abs(q["correlation"] - 0.789965) < 1e-5
