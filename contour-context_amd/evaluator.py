"""Loop-closure evaluator for the Python API: the reference's `ContLCDEvaluator` (include/eval/evaluator.h:54-431) and
`ConstellCorrelation::evalMetricEst` (include/cont2/correlation.h:241-280) in numpy -- ground-truth loop rule (an
earlier scan by >= 15 s within 5 m), TP/FP/TN/FN rule, pose error of a proposed transform, outcome file.  Same file
formats as the reference (see pr_eval.py, which turns the outcome file into max-F1 / PR points); the C++ twin is
hostcpp/eval/evaluator.h.  Pinned on the result files the reference ships (tests/test_evaluator_mirror.py)."""
import bisect
import math

import numpy as np

TP, FP, TN, FN = 0, 1, 2, 3
TS_DIFF_TOL = 10e-3      # a scan is used only if a gt pose lies within 10 ms
MIN_TIME_EXCL = 15.0     # revisits younger than this are not loops
GT_RADIUS = 5.0


def _rot_via_quaternion(M):
    """Quaterniond(M) -> rotation matrix: how the reference's loader re-orthonormalises a pose (evaluator.h:118-121)."""
    tr = M[0, 0] + M[1, 1] + M[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        x, y, z = (M[2, 1] - M[1, 2]) * s, (M[0, 2] - M[2, 0]) * s, (M[1, 0] - M[0, 1]) * s
    else:
        i = 0
        if M[1, 1] > M[0, 0]:
            i = 1
        if M[2, 2] > M[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(M[i, i] - M[j, j] - M[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * s
        s = 0.5 / s
        w = (M[k, j] - M[j, k]) * s
        q[j] = (M[j, i] + M[i, j]) * s
        q[k] = (M[k, i] + M[i, k]) * s
        x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def eval_metric_est(T_delta, gt_src, gt_tgt, n_row=150, n_col=150, reso=1.0):
    """T_delta = (x, y, theta) of the BEV-frame transform; gt_* = 3x4 sensor poses.  Returns (err_x, err_y, err_theta) of
    the estimated sensor-to-sensor transform against the ground truth projected to the plane (T_gt^-1 * T_est)."""
    ox, oy = n_row // 2 - 0.5, n_col // 2 - 0.5
    c, s = math.cos(T_delta[2]), math.sin(T_delta[2])
    est = np.array([[c, -s, (c * ox - s * oy + T_delta[0] - ox) * reso], [s, c, (s * ox + c * oy + T_delta[1] - oy) * reso], [0, 0, 1.0]])

    def hom(p):
        m = np.eye(4)
        m[:3, :4] = np.asarray(p).reshape(3, 4)
        return m
    rel = np.linalg.inv(hom(gt_tgt)) @ hom(gt_src)
    z1 = rel[:3, 2]
    ax = np.array([-z1[1], z1[0], 0.0])
    n = np.linalg.norm(ax)
    if n > 0:
        ax /= n
    ang = -math.acos(max(-1.0, min(1.0, z1[2])))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rr = (np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)) @ rel[:3, :3]
    yaw = math.atan2(Rr[1, 0], Rr[0, 0])
    gt = np.array([[math.cos(yaw), -math.sin(yaw), rel[0, 3]], [math.sin(yaw), math.cos(yaw), rel[1, 3]], [0, 0, 1.0]])
    e = np.linalg.inv(gt) @ est
    return e[0, 2], e[1, 2], math.atan2(e[1, 0], e[0, 0])


class ContLCDEvaluator:
    def __init__(self, fpath_pose, fpath_laser, sim_thres):
        self.sim_thres = float(sim_thres)
        ts, poses = [], []
        for line in open(fpath_pose):
            v = [float(x) for x in line.split()]
            if len(v) != 13:
                continue
            P = np.array(v[1:]).reshape(3, 4)
            P[:, :3] = _rot_via_quaternion(P[:, :3].copy())
            ts.append(v[0])
            poses.append(P)
        order = np.argsort(np.asarray(ts), kind="stable")
        gt_ts = [ts[i] for i in order]
        gt_poses = [poses[i] for i in order]
        self.scans = []  # dicts: seq, ts, fpath, pose, gt_pos
        for line in open(fpath_laser):
            p = line.split()
            if len(p) < 3:
                continue
            t = float(p[0])
            k = bisect.bisect_left(gt_ts, t)
            cand = [i for i in (k - 1, k) if 0 <= i < len(gt_ts)]
            if not cand:
                continue
            # nearest stamp; the reference prefers the lower neighbour on a tie (tools/algos.h:85)
            best = min(cand, key=lambda i: (abs(gt_ts[i] - t), i))
            if abs(gt_ts[best] - t) > TS_DIFF_TOL:
                continue
            self.scans.append({"seq": int(p[1]), "ts": t, "fpath": p[2], "pose": gt_poses[best], "gt_pos": False})
        for a, b in zip(self.scans, self.scans[1:]):
            assert a["seq"] < b["seq"] and a["ts"] < b["ts"], "scan list must be ordered by seq and ts"
        self._addr = {s["seq"]: i for i, s in enumerate(self.scans)}
        xyz = np.array([s["pose"][:, 3] for s in self.scans]) if self.scans else np.zeros((0, 3))
        tss = np.array([s["ts"] for s in self.scans])
        for i, s in enumerate(self.scans):
            n_old = int(np.searchsorted(tss, s["ts"] - MIN_TIME_EXCL, side="right"))  # slow.ts + 15 <= fast.ts
            if n_old and (np.linalg.norm(xyz[:n_old] - xyz[i], axis=1) < GT_RADIUS).any():
                s["gt_pos"] = True
        self.records = []
        self._tp_t, self._tp_r = [], []

    def add_prediction(self, id_tgt, est_corr, id_src=None, T_delta=(0.0, 0.0, 0.0), n_row=150, n_col=150, reso=1.0):
        tgt = self.scans[self._addr[id_tgt]]
        rec = {"id_tgt": id_tgt, "id_src": -1, "corr": float(est_corr), "err": (0.0, 0.0, 0.0)}
        if id_src is not None and id_src >= 0:
            src = self.scans[self._addr[id_src]]
            rec["id_src"] = id_src
            rec["err"] = eval_metric_est(T_delta, src["pose"], tgt["pose"], n_row, n_col, reso)
            gt_dist = float(np.linalg.norm(src["pose"][:, 3] - tgt["pose"][:, 3]))
            if est_corr >= self.sim_thres:
                rec["tfpn"] = TP if (tgt["gt_pos"] and gt_dist < GT_RADIUS) else FP
                if rec["tfpn"] == TP:
                    self._tp_t.append(math.hypot(rec["err"][0], rec["err"][1]))
                    self._tp_r.append(abs(rec["err"][2]))
            else:
                rec["tfpn"] = FN if tgt["gt_pos"] else TN
        else:
            rec["tfpn"] = FN if tgt["gt_pos"] else TN
        self.records.append(rec)
        return rec

    def save_prediction_results(self, path):
        with open(path, "w") as f:
            for r in self.records:
                tgt = self.scans[self._addr[r["id_tgt"]]]["fpath"]
                src = "x" if r["id_src"] < 0 else self.scans[self._addr[r["id_src"]]]["fpath"]
                pair = "%d-%s" % (r["id_tgt"], "x" if r["id_src"] < 0 else str(r["id_src"]))
                f.write("%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (r["tfpn"], pair, _g6(r["corr"]), _g6(r["err"][0]), _g6(r["err"][1]),
                                                              _g6(r["err"][2]), tgt[-32:], src[-32:]))

    def tp_errors(self):
        """(mean translation, mean rotation, rmse translation, rmse rotation) over the true positives, -1 if none"""
        if not self._tp_t:
            return -1.0, -1.0, -1.0, -1.0
        t, r = np.asarray(self._tp_t), np.asarray(self._tp_r)
        return float(t.mean()), float(r.mean()), float(np.sqrt((t ** 2).mean())), float(np.sqrt((r ** 2).mean()))


def _g6(v):
    """what `std::ostream << double` prints with the default precision of 6"""
    s = "%.6g" % v
    return "0" if s in ("-0",) else s
