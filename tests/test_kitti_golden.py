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


def test_auto_correlation_term_against_the_full_double_sum(oracle):
    """cc_k_gmm_prep evaluates the upper triangle of the self term and drops far pairs by an f32 bound (k_gmm.h): the value
    must equal the reference's full ordered double sum (correlation.h:102-119), here in numpy on the record's own ellipses,
    to the rounding of an f64 sum."""
    L = oracle.L
    desc = _load(L)[0]
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    _, feat = api.pack(ctx, desc[:6])
    ecap = (feat.shape[1] - 40) // (4 * 32)   # cc_gmm_feat: n_ell[4], flags, pad[3], ac, ell[4][ecap] of 32 B
    for k in range(len(feat)):
        raw = feat[k].tobytes()
        n_ell = np.frombuffer(raw, np.int32, 4, 0)
        ac = np.frombuffer(raw, np.float64, 1, 32)[0]
        ell = np.frombuffer(raw, np.float32, 4 * ecap * 8, 40).reshape(4, ecap, 8).astype(np.float64)
        tot = 0.0
        far = 0
        for li in range(4):
            e = ell[li, :n_ell[li]]
            n00 = 2 * (e[:, None, 0] + e[None, :, 0]); n01 = 2 * (e[:, None, 1] + e[None, :, 1])
            n10 = 2 * (e[:, None, 2] + e[None, :, 2]); n11 = 2 * (e[:, None, 3] + e[None, :, 3])
            mx = e[:, None, 4] - e[None, :, 4]; my = e[:, None, 5] - e[None, :, 5]
            det = n00 * n11 - n10 * n01
            i00, i10, i01, i11 = n11 / det, -n10 / det, -n01 / det, n00 / det
            r0 = -0.5 * mx * i00 - 0.5 * my * i10; r1 = -0.5 * mx * i01 - 0.5 * my * i11
            x = r0 * mx + r1 * my
            far += int((x < -100).sum())
            tot += float((e[:, None, 6] * e[None, :, 6] / np.sqrt(det) * np.exp(x)).sum())
        assert n_ell[:3].min() > 60 and far > 1000          # street-scene sized tables, and the far-pair bound has work to do
        assert abs(ac - tot) <= 1e-13 * abs(tot), (k, ac, tot)
