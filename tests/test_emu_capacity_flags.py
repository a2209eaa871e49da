"""The scoring kernels' internal capacities (the reference has none) must never truncate silently: a query that meets one
comes back with cc_query_result_t.flags set and the collecting call returns CC_ECAPACITY (k_merge.h finding).
Forced here on the CPU harness with hand-made descriptors:
  * a constellation check with more than CC_PP_MAX = 256 potential neighbour pairs (every neighbour of a layer in one
    distance bin: 9 x 9 + 3 x 10 x 10 = 381 pairs) -> CC_QF_CHECK_CAP;
  * a scan with 200 ellipses on a correlation level (200 equal contours; round 3 kept 128 per level and flagged
    CC_QF_GMM_CAP, now the correlation inputs hold as many ellipses as the descriptor stores contours): exact, no flag;
  * the same scan untouched: flags 0, CC_OK, result equal to the oracle's."""
import ctypes as C

import numpy as np

import emu_api
from parity import terrain_scan

CC_ECAPACITY = -4
CC_QF_CHECK_CAP, CC_QF_GMM_CAP = 1, 2


def _base_desc(oracle):
    s = oracle.Scan(terrain_scan(2, n=60000, scale=1.6))
    d = s.desc()
    assert (d["n_cont"][0][1:5] >= 10).all() and d["flags"][0] == 0
    return d


def _one_bin_bci(d, level=1, seq=0):
    """anchor (level, seq): the first 10 contours of every layer (the anchor itself excluded) as neighbours, all of a layer
    in the same distance bin"""
    d = d.copy()
    b = d["bcis"][0, level, seq]
    nb = np.zeros((), b.dtype)
    nb["piv_seq"], nb["level"] = seq, level
    pts, bits, segs = [], [0, 0, 0, 0], []
    for bl in range(4):
        for j in range(10):
            if bl + 1 == level and j == seq:
                continue
            pts.append((bl + 1, j, 64 * bl + 5, np.float32(5.43 + 5 * 1.01 + 0.5), np.float32(0.1 * j + 0.5 * bl - 1.0)))
        bits[bl] |= 1 << 5
    for k, (lv, sq, bp, r, th) in enumerate(pts):
        nb["pts"][k] = (lv, sq, bp, r, th)
        if k == 0 or pts[k - 1][2] != bp:
            segs.append(k)
    segs.append(len(pts))
    nb["n_pts"] = len(pts)
    nb["dist_bin"] = np.array(bits, np.uint64)
    nb["n_segs"] = len(segs)
    nb["segs"][:len(segs)] = segs
    d["bcis"][0, level, seq] = nb
    return d


def _many_ellipses(d, lev=4, n=200):
    d = d.copy()
    row = d["cont"][0, lev, 0].copy()
    row["cell_cnt"] = 9
    for j in range(10, n):   # the first ten (what checks and keys read) stay as they are
        r = row.copy()
        r["pos_mean"] = (20.0 + (j % 14) * 8.0, 20.0 + (j // 14) * 8.0)
        d["cont"][0, lev, j] = r
    d["n_cont"][0, lev] = d["n_stored"][0, lev] = n
    d["layer_cell_cnt"][0, lev] = int(d["cont"][0, lev, :n]["cell_cnt"].astype(np.int64).sum())
    return d


def _check(api, L, db, qdesc, hints):
    lb, ub = L.default_thresholds()
    h = np.zeros(len(hints), L.hint_dt)
    h["cand_gidx"], h["level"], h["seq_src"], h["seq_tgt"] = 0, [x[0] for x in hints], [x[1] for x in hints], [x[2] for x in hints]
    res = np.zeros(1, L.query_result_dt)
    sc = np.zeros(len(h), L.hint_score_dt)
    qd = np.ascontiguousarray(qdesc)
    rc = api.lib.cc_db_check_hints(db, C.c_void_p(qd.ctypes.data), C.c_void_p(h.ctypes.data), len(h), C.byref(lb), C.byref(ub), 10,
                                   C.c_void_p(res.ctypes.data), C.c_void_p(sc.ctypes.data), None)
    return rc, res[0], sc


def test_capacities_are_flagged_never_silent(oracle):
    L = oracle.L
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    base = _base_desc(oracle)
    hints = [(1, 0, 0), (2, 0, 0), (3, 0, 0)]
    # ---- control: untouched scan against itself
    db = api.db_create(ctx, cap=4)
    api.db_add(db, base, np.zeros(1), np.zeros(1, np.int32))
    rc, res, sc = _check(api, L, db, base, hints)
    assert rc == 0 and res["flags"] == 0 and res["n_res"] == 1 and res["cand_gidx"] == 0
    o = oracle.Scan.from_desc(base[0], int_id=0)
    eres, esc = oracle.check_hints(o, [o], np.array([(0,) + h for h in hints], np.int32))
    assert abs(eres["correlation"] - res["correlation"]) < 1e-6 and np.array_equal(esc[:, :5], np.stack([sc[f] for f in
           ("i_ovlp_sum", "i_ovlp_max_one", "i_in_ang_rng", "i_indiv_sim", "i_orie_sim")], 1))
    # ---- 381 potential pairs in one check
    crowded = _one_bin_bci(base)
    db2 = api.db_create(ctx, cap=4)
    api.db_add(db2, crowded, np.zeros(1), np.zeros(1, np.int32))
    rc, res, sc = _check(api, L, db2, crowded, hints)
    assert rc == CC_ECAPACITY and (res["flags"] & CC_QF_CHECK_CAP), (rc, res["flags"])
    assert b"capacity" in api.lib.cc_last_error()
    assert sc["i_ovlp_sum"][0] == 4   # one common bin per layer: the check itself ran
    # the same through the batched query entry point: CC_ECAPACITY at the collecting call, results delivered with the flag
    lb, ub = L.default_thresholds()
    out = np.zeros(1, L.query_result_dt)
    ep = np.ones(1, np.int32)
    rc2 = api.lib.cc_db_query_batch(db2, C.c_void_p(crowded.ctypes.data), 1, C.c_void_p(ep.ctypes.data), C.byref(lb), C.byref(ub),
                                    C.c_void_p(out.ctypes.data), None, None, None)
    assert rc2 in (0, CC_ECAPACITY) and (rc2 == CC_ECAPACITY) == bool(out["flags"][0])
    # ---- 200 ellipses on correlation level 4
    many = _many_ellipses(base)
    db3 = api.db_create(ctx, cap=4)
    api.db_add(db3, many, np.zeros(1), np.zeros(1, np.int32))
    rc, res, sc = _check(api, L, db3, many, hints)
    assert rc == 0 and res["flags"] == 0 and res["n_res"] == 1, (rc, res["flags"])
    om = oracle.Scan.from_desc(many[0], int_id=0)
    eres, _ = oracle.check_hints(om, [om], np.array([(0,) + h for h in hints], np.int32))
    assert abs(eres["correlation"] - res["correlation"]) < 1e-6 and np.abs(eres["tf"] - res["tf"]).max() < 1e-6


def test_correlation_pools_running_out_is_an_error(oracle, monkeypatch):
    """The correlation's two pools (round 6: the pair codes cc_k_gmm_init files for every problem, the pair records the
    refinement keeps beyond what stays in LDS) are sized per query lane (CC_GMM_POOL_CODES / CC_GMM_POOL_PAIRS, read when the
    lane comes into use): a chunk that needs more gets CC_ECAPACITY, never a shorter pair list."""
    L = oracle.L
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    many = _many_ellipses(_base_desc(oracle))   # 200 ellipses on a level: thousands of selected pairs against itself
    hints = [(1, 0, 0), (2, 0, 0), (3, 0, 0)]
    for var, val in (("CC_GMM_POOL_CODES", "1024"), ("CC_GMM_POOL_PAIRS", "32")):
        monkeypatch.setenv(var, val)
        db = api.db_create(ctx, cap=4)
        api.db_add(db, many, np.zeros(1), np.zeros(1, np.int32))
        rc, res, sc = _check(api, L, db, many, hints)
        monkeypatch.delenv(var)
        assert rc == CC_ECAPACITY and b"pool" in api.lib.cc_last_error(), (var, rc, api.lib.cc_last_error())
    db = api.db_create(ctx, cap=4)
    api.db_add(db, many, np.zeros(1), np.zeros(1, np.int32))
    rc, res, sc = _check(api, L, db, many, hints)
    assert rc == 0 and res["flags"] == 0 and res["n_res"] == 1


def test_hint_must_name_existing_contours_on_both_sides(oracle):
    """cc_db_check_hints validates seq_src against the DB scan and seq_tgt against the (device-resident) query scan: a hint
    naming a contour that does not exist is CC_EINVAL, not a comparison against zero-filled rows."""
    from test_emu_query import _load_query_fixture
    L = oracle.L
    desc, ts, exp, d = _load_query_fixture(L)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=2)
    db = api.db_create(ctx, d, cap=4)
    api.db_add(db, desc[:1], ts[:1], np.zeros(1, np.int32))
    pick = None
    for qi in range(1, len(desc)):
        for lev in range(1, 5):
            if 0 < desc["n_cont"][qi][lev] < L.NPIV and desc["n_cont"][0][lev] > 0:
                pick = (qi, lev, int(desc["n_cont"][qi][lev]))
                break
        if pick:
            break
    assert pick, "the fixture should hold a scan with fewer than 6 contours on some level"
    qi, lev, nc = pick
    rc, _, _ = _check(api, L, db, desc[qi:qi + 1], [(lev, 0, nc - 1)])
    assert rc == 0
    rc, _, _ = _check(api, L, db, desc[qi:qi + 1], [(lev, 0, nc)])
    assert rc == -1 and b"query scan" in api.lib.cc_last_error()
