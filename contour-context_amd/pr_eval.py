"""Loop-closure evaluation: max-F1 / PR points / TP pose error from (GT pose file, outcome file).

Own counterpart of the reference's harness `scripts/pr_mpe.py:get_points_ours2` (:71-165) and of the file
formats it consumes:
  * pose file  : `ts r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz` per line (scripts/gen_batch_bin_configs.py:101-159)
  * outcome file: `tfpn \t tgt-src \t correlation \t err_x \t err_y \t err_theta \t tgt_path \t src_path`
    (include/eval/evaluator.h:370-425), `src` = `x` when no candidate was returned.
Vectorised (cumulative sums instead of the reference's O(n^2) loop); pinned against numbers produced by the
reference script itself (tests/golden/pr_mpe_kitti08.json).
"""
import gzip
import math

import numpy as np

THRES_DIST = 5.0      # pr_mpe.py:9
EXCL_INDICES = 150    # pr_mpe.py:87  (j < i - 150)


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def load_gt_poses(path):
    """-> [n, 12] row-major 3x4 sensor poses (time column dropped)."""
    rows = []
    with _open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            assert len(p) == 13, "pose line must have 13 columns"
            rows.append([float(x) for x in p[1:]])
    return np.asarray(rows, np.float64)


def load_outcome(path):
    """-> list of dicts(idx_curr, idx_best or None, corr, err[3])."""
    out = []
    with _open(path) as f:
        for line in f:
            p = line.strip().split()
            if not p:
                continue
            assert len(p) > 5
            a, b = p[1].split("-")
            out.append({"idx_curr": int(a), "idx_best": None if b == "x" else int(b), "corr": float(p[2]),
                        "err": (float(p[3]), float(p[4]), float(p[5]))})
    return out


def gt_positive(gt_pose):
    """GT-positive(i) iff some j < i - 150 lies within 5 m (pr_mpe.py:84-89)."""
    from scipy.spatial import cKDTree
    pts = gt_pose[:, [3, 7, 11]]
    tree = cKDTree(pts)
    pos = np.zeros(len(pts), bool)
    for i, near in enumerate(tree.query_ball_point(pts, THRES_DIST)):
        pos[i] = any(j < i - EXCL_INDICES for j in near)
    return pos


def evaluate(gt_pose, outcome):
    """Returns dict(max_f1, max_f1_idx, sim_thres, tp_count, rot_mean_deg, rot_rmse_deg, trans_mean, trans_rmse,
    pr_points [n,2] sorted by recall)."""
    pts = gt_pose[:, [3, 7, 11]]
    gpos = gt_positive(gt_pose)
    n = len(outcome)
    corr = np.array([o["corr"] for o in outcome])
    idx_curr = np.array([o["idx_curr"] for o in outcome])
    correct = np.zeros(n, bool)
    for k, o in enumerate(outcome):
        if o["idx_best"] is not None:
            correct[k] = np.linalg.norm(pts[o["idx_curr"]] - pts[o["idx_best"]]) < THRES_DIST
    is_pos = gpos[idx_curr]
    order = np.argsort(-corr, kind="stable")  # pr_mpe.py:123 uses argsort of -corr (quicksort); ties are data-dependent
    c_s, p_s = correct[order], is_pos[order]
    tp = np.cumsum(c_s)
    fp = np.cumsum(~c_s)
    fn = p_s[::-1].cumsum()[::-1] - p_s          # GT positives among the not-yet-predicted (j > i)
    recall = tp / np.maximum(tp + fn, 1)
    recall = np.where(tp + fn > 0, tp / np.maximum(tp + fn, 1), 0.0)
    precision = tp / (tp + fp)
    f1 = np.where(recall + precision > 0, 2 * recall * precision / np.maximum(recall + precision, 1e-300), 0.0)
    # get_maxf1_idx keeps the FIRST strict maximum (pr_mpe.py:29-41)
    best = int(np.argmax(f1))
    max_f1 = float(f1[best])
    f1_pose_idx = int(idx_curr[order][best])
    sim_thres = outcome[f1_pose_idx]["corr"]      # pr_mpe.py:141 indexes the file's LINE number with the pose idx
    sq_t = sq_r = ab_t = ab_r = 0.0
    cnt = 0
    for k, o in enumerate(outcome):
        if o["corr"] >= sim_thres and correct[k] and is_pos[k]:
            e = o["err"]
            t2 = e[0] ** 2 + e[1] ** 2
            sq_t += t2
            ab_t += math.sqrt(t2)
            sq_r += e[2] ** 2
            ab_r += abs(e[2])
            cnt += 1
    pr = np.stack([recall, precision], 1)
    pr = pr[np.argsort(pr[:, 0], kind="stable")]
    res = {"max_f1": max_f1, "max_f1_idx": f1_pose_idx, "sim_thres": sim_thres, "tp_count": cnt, "pr_points": pr}
    if cnt:
        res.update({"rot_mean_deg": ab_r / cnt / math.pi * 180, "rot_rmse_deg": math.sqrt(sq_r / cnt) / math.pi * 180,
                    "trans_mean": ab_t / cnt, "trans_rmse": math.sqrt(sq_t / cnt)})
    return res


def write_outcome(path, records, paths=None):
    """records: iterable of (tfpn, id_tgt, id_src or -1, correlation, err_x, err_y, err_theta); same row format as
    ContLCDEvaluator::savePredictionResults (evaluator.h:370-425)."""
    with open(path, "w") as f:
        for r in records:
            tfpn, it, isrc, c, ex, ey, et = r
            pair = "%d-x" % it if isrc < 0 else "%d-%d" % (it, isrc)
            pt = paths[it] if paths else "tgt"
            ps = "x" if isrc < 0 else (paths[isrc] if paths else "src")
            f.write("%d\t%s\t%g\t%g\t%g\t%g\t%s\t%s\n" % (tfpn, pair, c, ex, ey, et, pt[-32:], ps[-32:]))
