"""P1: max-F1 / PR / pose-error evaluation pinned by numbers produced with the reference's own scripts/pr_mpe.py
(tests/golden/make_pr_golden.py) on the two result files the reference ships."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_max_f1_and_pose_errors(cc):
    import pr_eval
    g = json.load(open(os.path.join(G, "pr_mpe_kitti08.json")))
    r = pr_eval.evaluate(pr_eval.load_gt_poses(os.path.join(G, "ts-sens_pose-kitti08.txt.gz")),
                         pr_eval.load_outcome(os.path.join(G, "outcome-kitti08.txt.gz")))
    assert abs(r["max_f1"] - g["max_f1"]) < 1e-6
    assert r["max_f1_idx"] == g["max_f1_idx"] and r["tp_count"] == g["tp_count"]
    assert abs(r["sim_thres"] - g["sim_thres"]) < 1e-9
    for k in ["rot_mean_deg", "rot_rmse_deg", "trans_mean", "trans_rmse"]:
        assert abs(r[k] - g[k]) < 1e-12, k
    assert len(r["pr_points"]) == g["n_pr_points"]
    assert np.allclose(r["pr_points"][:5], g["pr_points_head"])
    # PR sweep over the predictions with correlation > 0 (the zero-correlation tail is ordered by an unstable
    # argsort in the reference and carries no information)
    gp = np.asarray(g["pr_points_poscorr"])
    mine = r["pr_points"]
    for p in gp[:: max(1, len(gp) // 50)]:
        assert np.isclose(mine, p, atol=1e-12).all(1).any(), p


def test_outcome_roundtrip(cc, tmp_path):
    import pr_eval
    recs = [(2, 0, -1, 0.0, 0, 0, 0), (0, 200, 3, 0.81, 0.01, -0.2, 0.001)]
    p = tmp_path / "o.txt"
    pr_eval.write_outcome(p, recs)
    o = pr_eval.load_outcome(p)
    assert o[0]["idx_best"] is None and o[1]["idx_best"] == 3 and abs(o[1]["corr"] - 0.81) < 1e-9
