#!/usr/bin/env python3
"""Writes the two text files the evaluator reads (`ts 12-element sensor pose` per line, `ts seq bin-path` per line) from
a dataset in its native layout -- the job of the reference's scripts/gen_batch_bin_configs.py.

  KITTI odometry:  python gen_lists.py kitti  <velodyne dir> <poses/NN.txt> <sequences/NN/times.txt> <calib.txt> <out pose> <out list>
      sensor pose = Tr^-1 * T_cam0 * Tr (camera-frame ground truth moved to the LiDAR frame, first LiDAR frame = origin)
  MulRan:          python gen_lists.py mulran <Ouster dir> <global_pose.csv> <out pose> <out list>
      base poses (ns stamps) -> LiDAR poses relative to the first one; the scan stamp is the .bin file name (ns).
"""
import csv
import os
import sys

import numpy as np


def _hom(m34):
    return np.vstack([np.asarray(m34, float).reshape(3, 4), [0, 0, 0, 1]])


def _bins(d):
    files = sorted(f for f in os.listdir(d) if f.endswith(".bin") and os.path.isfile(os.path.join(d, f)))
    for f in files:
        assert " " not in os.path.join(d, f), "paths in the list file must not contain spaces"
    return [os.path.join(d, f) for f in files]


def _write(sav_pose, sav_list, ts, poses12, paths):
    with open(sav_pose, "w") as f:
        for t, p in zip(ts, poses12):
            f.write(" ".join("%.6f" % v for v in [t] + list(p)) + "\n")
    with open(sav_list, "w") as f:
        f.write("\n".join("%.6f %d %s" % (t, i, p) for i, (t, p) in enumerate(paths)))


def gen_kitti(dir_bins, f_pose, f_times, f_calib, sav_pose, sav_list, first_bin=0):
    bins = _bins(dir_bins)
    poses = [[float(v) for v in l.split()] for l in open(f_pose) if l.strip()]
    times = [float(l) for l in open(f_times) if l.strip()]
    assert len(poses) == len(times) and len(bins) >= len(poses) + first_bin
    Tr = np.eye(4)
    for l in open(f_calib):
        p = l.split()
        if p and p[0] == "Tr:":
            Tr = _hom([float(v) for v in p[1:13]])
    Tri = np.linalg.inv(Tr)
    out = [(Tri @ _hom(p) @ Tr)[:3].reshape(-1) for p in poses]
    _write(sav_pose, sav_list, times, out, [(times[i], bins[i + first_bin]) for i in range(len(times))])
    return len(times)


# lidar_to_base_init_se3 of the MulRan calibration: x y z (m), roll pitch yaw (deg)
MULRAN_LIDAR_TO_BASE = (1.7042, -0.021, 1.8047, 0.0001, 0.0003, 179.6654)


def _mulran_extrinsic():
    x, y, z, roll, pitch, yaw = MULRAN_LIDAR_TO_BASE
    r, p, w = np.deg2rad([roll, pitch, yaw])
    rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    rz = np.array([[np.cos(w), -np.sin(w), 0], [np.sin(w), np.cos(w), 0], [0, 0, 1]])
    # NOTE: the reference composes `rotz * roty @ rotx` with an ELEMENTWISE product between the first two factors
    # (scripts/gen_batch_bin_configs.py:46).  With roll and pitch of 1e-4 degrees the difference to Rz Ry Rx is ~1e-6,
    # but files meant to be compared with the reference's must reproduce it, so it is kept.
    T = np.eye(4)
    T[:3, :3] = (rz * ry) @ rx
    T[:3, 3] = [x, y, z]
    return T


def gen_mulran(dir_bins, f_global_pose, sav_pose, sav_list):
    Tlb_inv = np.linalg.inv(_mulran_extrinsic())
    ts, poses, T0_inv = [], [], None
    with open(f_global_pose, newline="") as cf:
        for row in csv.reader(cf, delimiter=","):
            if len(row) != 13:
                continue
            try:
                vals = [float(a) for a in row]
            except ValueError:
                continue
            T_wl = _hom(vals[1:]) @ Tlb_inv
            if T0_inv is None:
                T0_inv = np.linalg.inv(T_wl)
            ts.append(vals[0] * 1e-9)
            poses.append((T0_inv @ T_wl)[:3].reshape(-1))
    bins = _bins(dir_bins)
    _write(sav_pose, sav_list, ts, poses, [(int(os.path.basename(b).split(".")[0]) * 1e-9, b) for b in bins])
    return len(ts), len(bins)


if __name__ == "__main__":
    if len(sys.argv) >= 8 and sys.argv[1] == "kitti":
        print("wrote %d poses" % gen_kitti(*sys.argv[2:8]))
    elif len(sys.argv) >= 6 and sys.argv[1] == "mulran":
        print("wrote %d poses, %d scans" % gen_mulran(*sys.argv[2:6]))
    else:
        print(__doc__)
        sys.exit(2)
