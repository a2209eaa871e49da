#!/usr/bin/env python3
"""Row-by-row comparison of two outcome files (the 8-column format of ContLCDEvaluator::savePredictionResults,
include/eval/evaluator.h:370-425) -- the "pin on arrival": the day a KITTI-08 directory is mounted, the outcome file this
build produces on it is diffed against the one the reference ships (results/outcome_txt/outcome-kitti08.txt, committed as
tests/golden/outcome-kitti08.txt.gz), which pins OpenCV's component order, Eigen's solvers and Ceres' L-BFGS together.

    python contour-context_amd/tools/compare_outcome.py OURS REFERENCE [--poses POSE_FILE] [--tol-corr 1e-5] [--tol-pose 1e-4]

What "green" means (tests/test_gpu_kitti_pin.py asserts exactly this):
  * the same number of rows, row i about the same target scan;
  * the same matched scan in every row (`tgt-src`, `x` = none) and the same TP/FP/TN/FN label;
  * correlation equal to the 6 significant digits the reference prints (|d| <= tol_corr * max(1, |c|));
  * the three pose-error columns within tol_pose;
  * with a ground-truth pose file: identical max-F1, arg-max and true-positive count (pr_eval, the counterpart of
    scripts/pr_mpe.py).
Exit code 0 iff all of it holds.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pr_eval  # noqa: E402


def load_rows(path):
    rows = []
    with pr_eval._open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) < 6:
                p = line.split()
            if len(p) < 6:
                continue
            a, b = p[1].split("-")
            rows.append({"tfpn": int(p[0]), "tgt": int(a), "src": None if b == "x" else int(b), "corr": float(p[2]),
                         "err": (float(p[3]), float(p[4]), float(p[5]))})
    return rows


def compare(path_ours, path_ref, poses=None, tol_corr=1e-5, tol_pose=1e-4, max_report=20):
    """-> dict(ok, n_rows, n_match_diff, n_label_diff, n_corr_diff, n_pose_diff, max_corr_diff, max_pose_diff, details,
    and with `poses`: max_f1_ours / max_f1_ref / ... )"""
    ours, ref = load_rows(path_ours), load_rows(path_ref)
    out = {"n_rows_ours": len(ours), "n_rows_ref": len(ref), "n_match_diff": 0, "n_label_diff": 0, "n_corr_diff": 0, "n_pose_diff": 0,
           "max_corr_diff": 0.0, "max_pose_diff": 0.0, "details": []}

    def note(s):
        if len(out["details"]) < max_report:
            out["details"].append(s)

    for i, (a, b) in enumerate(zip(ours, ref)):
        if a["tgt"] != b["tgt"]:
            out["n_match_diff"] += 1
            note("row %d: target %d vs %d" % (i, a["tgt"], b["tgt"]))
            continue
        if a["src"] != b["src"]:
            out["n_match_diff"] += 1
            note("row %d (scan %d): matched %s vs reference %s (corr %g vs %g)" % (i, a["tgt"], a["src"], b["src"], a["corr"], b["corr"]))
            continue
        if a["tfpn"] != b["tfpn"]:
            out["n_label_diff"] += 1
            note("row %d (scan %d): label %d vs %d" % (i, a["tgt"], a["tfpn"], b["tfpn"]))
        dc = abs(a["corr"] - b["corr"])
        out["max_corr_diff"] = max(out["max_corr_diff"], dc)
        if dc > tol_corr * max(1.0, abs(b["corr"])):
            out["n_corr_diff"] += 1
            note("row %d (scan %d-%s): correlation %g vs %g" % (i, a["tgt"], a["src"], a["corr"], b["corr"]))
        dp = max(abs(x - y) for x, y in zip(a["err"], b["err"]))
        out["max_pose_diff"] = max(out["max_pose_diff"], dp)
        if dp > tol_pose:
            out["n_pose_diff"] += 1
            note("row %d (scan %d-%s): pose error columns %s vs %s" % (i, a["tgt"], a["src"], a["err"], b["err"]))
    out["ok"] = (len(ours) == len(ref) and not (out["n_match_diff"] or out["n_label_diff"] or out["n_corr_diff"] or out["n_pose_diff"]))
    if poses:
        gt = pr_eval.load_gt_poses(poses)
        eo = pr_eval.evaluate(gt, pr_eval.load_outcome(path_ours))
        er = pr_eval.evaluate(gt, pr_eval.load_outcome(path_ref))
        for k in ("max_f1", "max_f1_idx", "sim_thres", "tp_count"):
            out[k + "_ours"], out[k + "_ref"] = eo[k], er[k]
        out["ok"] = out["ok"] and eo["max_f1"] == er["max_f1"] and eo["max_f1_idx"] == er["max_f1_idx"] and eo["tp_count"] == er["tp_count"]
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ours")
    ap.add_argument("reference")
    ap.add_argument("--poses", default=None, help="ground-truth pose file (13 columns): also compare max-F1 / arg-max / TP count")
    ap.add_argument("--tol-corr", type=float, default=1e-5)
    ap.add_argument("--tol-pose", type=float, default=1e-4)
    a = ap.parse_args()
    r = compare(a.ours, a.reference, a.poses, a.tol_corr, a.tol_pose)
    for d in r.pop("details"):
        print(d)
    for k in sorted(r):
        print("%-18s %s" % (k, r[k]))
    sys.exit(0 if r["ok"] else 1)


if __name__ == "__main__":
    main()
