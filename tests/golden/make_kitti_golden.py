"""Generates tests/golden/kitti_fixture.npz: the CPU oracle's outputs on a cut of the KITTI-shaped drive
(synth.World(kitti=True): 4-6 k occupied cells, ~100 contours on the low levels, 18 valid DB keys per scan):
  * `desc`: descriptors of 24 full-size scans of the first pass along a street (scans 1484, 1486, ... 1530 of the drive)
    followed by 8 scans that drive the same street again 118 s later (2672, 2678, ... 2714), `ts` their 10-Hz stamps;
  * `res`: the oracle's replay of the reference driver loop on them (scan i queries the DB of the scans before it whose
    stamps are old enough, then is added): matched scan, correlation, pose, gate counters;
  * `pts` / `pts_desc`: two reduced scans of the same places (32 beams x 900 azimuth steps) with the oracle's descriptors, so
    that the ingest side (BEV, component numbering, 2x2 eigen-solver, sort replay, keys, BCIs) is recorded too.
Any later change to the restated third-party pieces (OpenCV numbering, Eigen, umeyama, Ceres) shows up as a diff of this
file's contents in tests/test_kitti_golden.py.  Re-run only when the oracle changes on purpose."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cc_amd  # noqa: E402
import oracle_py as O  # noqa: E402

cc = cc_amd.load()
w = cc.synth.World(kitti=True)
idx = np.concatenate([np.arange(1484, 1532, 2), np.arange(2672, 2720, 6)])
x, poses, ts = cc.synth.make_sequence(0, world=w, indices=idx)
n = len(idx)
xs = x.numpy().reshape(-1, 4)
offs = np.arange(n + 1, dtype=np.int64) * x.shape[1]
res, _, desc = O.run_sequence(xs, offs, ts, np.arange(n, dtype=np.int32), want_desc=True)
xr, _, _ = cc.synth.make_sequence(0, world=w, indices=[1500, 2690], beams=32, azim=900)
pts = xr.numpy()
pts_desc = np.concatenate([O.Scan(pts[k], int_id=k).desc() for k in range(2)])
np.savez_compressed(os.path.join(HERE, "kitti_fixture.npz"), desc=np.frombuffer(desc.tobytes(), np.uint8), ts=ts, idx=idx,
                    res=np.frombuffer(res.tobytes(), np.uint8), pts=pts, pts_desc=np.frombuffer(pts_desc.tobytes(), np.uint8))
print("loop closures:", int((res["n_res"] > 0).sum()), "of", n, "| matched:", idx[res["cand_gidx"][res["n_res"] > 0]].tolist())
print("occupied cells %.0f, contours per level %s" % (desc["n_pix"].mean(), desc["n_cont"].mean(0).round(1).tolist()))
