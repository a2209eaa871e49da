"""The committed record of the CPU oracle on the KITTI-shaped drive (tests/golden/kitti_fixture.npz, made by
tests/golden/make_kitti_golden.py; SURVEY.md 8(d)'s value distributions: ~5.9 k occupied cells, ~130 contours on the low
levels, 18 valid DB keys per scan):
  * the oracle still reproduces it -- a change to one of the pieces restated from third-party code (OpenCV's component
    numbering, Eigen's 2x2 solver, umeyama, Ceres' line search) shows up here as a diff;
  * the product kernels, run on the CPU harness (tests/emu), match it: the small twin of
    tests/test_gpu_query.py::test_sequence_kitti_shaped."""
import os

import numpy as np

import emu_api
from parity import compare_desc

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy",
              "n_knn_hits"]


def _load(L):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_fixture.npz"))
    desc = np.frombuffer(z["desc"].tobytes(), dtype=L.scan_desc_dt)
    res = np.frombuffer(z["res"].tobytes(), dtype=L.query_result_dt)
    pts_desc = np.frombuffer(z["pts_desc"].tobytes(), dtype=L.scan_desc_dt)
    return desc, z["ts"], z["idx"], res, z["pts"], pts_desc


def _same_result(exp, got, tol):
    for f in INT_FIELDS:
        assert exp[f] == got[f], (f, exp[f], got[f])
    if exp["n_res"]:
        assert abs(exp["correlation"] - got["correlation"]) < tol and np.abs(exp["tf"] - got["tf"]).max() < tol


def test_oracle_reproduces_the_kitti_shaped_record(oracle):
    L = oracle.L
    desc, ts, idx, res, pts, pts_desc = _load(L)
    n = len(desc)
    assert n == 32 and (res["n_res"] > 0).sum() == 7
    assert 4000 < desc["n_pix"].mean() < 9000 and 50 < desc["n_cont"][:, 1].mean() < 150
    assert ((np.abs(desc["keys"].reshape(n, 6, 6, 10)[:, 1:4]).sum(-1) > 0).sum((1, 2)) == 18).all()
    # ingest side: BEV, component numbering, eigen-solver, sort replay, keys, BCIs -- byte for byte
    for k in range(len(pts)):
        d = oracle.Scan(pts[k], int_id=k).desc()
        assert d.tobytes() == pts_desc[k:k + 1].tobytes(), compare_desc(pts_desc[k], d[0])[:5]
    # query side: the driver loop replayed from the descriptors
    odb = oracle.DB()
    for i in range(n):
        s = oracle.Scan.from_desc(desc[i], int_id=i)
        _same_result(res[i], odb.query(s), 1e-12)
        odb.add_scan(s, ts[i])
        odb.push_and_balance(i, ts[i])
    hit = np.nonzero(res["n_res"] > 0)[0]
    assert (np.abs((idx[hit] - 2672) // 6 * 6 + 1486 + 6 - idx[res["cand_gidx"][hit]]) <= 6).all()   # the same stretch of the street


def test_kernels_on_the_cpu_harness_match_the_kitti_shaped_record(oracle):
    L = oracle.L
    desc, ts, idx, res, pts, pts_desc = _load(L)
    n = len(desc)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    got = api.ingest(ctx, pts[0], np.array([0, len(pts[0])], np.int64))
    got = got[0] if isinstance(got, tuple) else got
    bad = compare_desc(pts_desc[0], got[0], float_exact=False)
    assert not bad, bad[:5]
    db = api.db_create(ctx, cap=n)
    seeds = np.arange(n, dtype=np.int32)
    api.db_add(db, desc, ts, seeds)
    hit = np.nonzero(res["n_res"] > 0)[0]
    qs = np.concatenate([hit[[0, 3, 6]], [10, 24]]).astype(np.int32)   # three revisits, a first-pass scan, the revisit without a result
    out = api.db_query(db, desc[qs], qs)
    for k, qi in enumerate(qs):
        _same_result(res[qi], out[k], 1e-6)
