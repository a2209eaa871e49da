"""Kernel logic of the ingest path on the CPU: the product's HIP translation unit compiled against the CPU
execution harness (tests/emu) and driven through the same C-ABI, compared with the oracle bit for bit
(same libm on both sides here)."""
import os

import numpy as np

import emu_api
from parity import compare_desc, terrain_scan

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(oracle, scans, cfg=None):
    api = emu_api.EmuApi(oracle.L)
    ctx = api.create(cfg=cfg, max_batch=len(scans))
    offs = np.concatenate([[0], np.cumsum([len(s) for s in scans])]).astype(np.int64)
    # first WITHOUT the debug outputs, on the fresh context: K1 then writes its dense image / positions only for scans whose
    # active cells overflow its list, and a scan the list kernel hands on gets them rebuilt from the list (cc_k_contours_mid)
    plain = api.ingest(ctx, np.concatenate(scans, 0), offs)
    desc, dbg = api.ingest(ctx, np.concatenate(scans, 0), offs, debug=True)
    assert plain.tobytes() == desc.tobytes(), "descriptors with and without the dense debug outputs differ"
    for i, s in enumerate(scans):
        o = oracle.Scan(s, cfg=cfg)
        ob, opix = o.bev()
        assert np.array_equal(ob, dbg["bev"][i])
        assert np.array_equal(opix, dbg["pix_rc"][i])
        assert np.array_equal(o.labels(), dbg["labels"][i]), "integer contour labels must be bit-exact"
        bad = compare_desc(o.desc()[0], desc[i], float_exact=True)
        assert not bad, bad[:10]
    return desc


def test_terrain_scans_bit_exact(oracle):
    d = _check(oracle, [terrain_scan(2, n=20000, scale=1.2), terrain_scan(101, n=5000, scale=2.0, quant=0.25)])
    assert d["n_cont"].max() > 16 and (d["flags"] == 0).all()


def test_edge_cases(oracle):
    tiny = np.zeros((11, 4), np.float32)
    far = np.full((40, 4), 1000.0, np.float32)
    ties = np.tile(np.array([[10.2, 3.3, 1.0, 0], [10.7, 3.9, 1.0, 0], [10.4, 3.1, 1.0, 0]], np.float32), (30, 1))
    _check(oracle, [tiny, far, ties])


def test_golden_fixture(oracle):
    """Committed inputs + expected descriptors (tests/golden/make_golden.py): the oracle still reproduces them and the
    emulated kernels match them."""
    z = np.load(os.path.join(G, "ingest_fixture.npz"))
    scans = [z["scan0"], z["scan1"]]
    exp = np.frombuffer(z["desc"].tobytes(), dtype=oracle.L.scan_desc_dt)
    for i, s in enumerate(scans):
        od = oracle.Scan(s).desc()[0]
        assert not compare_desc(exp[i], od, float_exact=True)
    d = _check(oracle, scans)
    for i in range(2):
        assert not compare_desc(exp[i], d[i], float_exact=True)


def test_height_ties_in_crowded_cells(oracle):
    """A few cells, hundreds of points each, heights on a 6-value lattice: the cell maximum is attained many times
    all over the file, and the FIRST such point must supply the cell's continuous (row, col) (contour_mng.h:517)."""
    scans = []
    for seed, n in ((5, 6000), (6, 2500)):
        rng = np.random.default_rng(seed)
        s = np.zeros((n, 4), np.float32)
        s[:, 0] = rng.uniform(10.0, 16.0, n)
        s[:, 1] = rng.uniform(-3.0, 3.0, n)
        s[:, 2] = rng.integers(0, 6, n) * 0.5 - 1.0
        # the maximum appears late for some cells, early for others
        late = rng.random(n) < 0.5
        s[: n // 2, 2] = np.where(late[: n // 2], np.minimum(s[: n // 2, 2], 0.5), s[: n // 2, 2])
        scans.append(s)
    _check(oracle, scans)


def test_mulran_level_set(oracle):
    """The other shipped level set (config/batch_bin_test_config.yaml:31, MulRan: wider, taller steps) on tall terrain
    with 200-290 contours on every level (close to the CC_MAXC capacity)."""
    cfg = oracle.L.default_manager_cfg()
    for i, v in enumerate([1.0, 2.5, 4.0, 5.5, 7.0, 8.5]):
        cfg.lv_grads[i] = v
    d = _check(oracle, [terrain_scan(12, n=6000, scale=5.0, quant=0.5)], cfg=cfg)
    assert (d["n_cont"][:, 5] > 0).all() and (d["flags"] == 0).all() and d["n_cont"].max() > 200


def test_component_capacity_is_gone(oracle):
    """More than CC_MAXC components on a level (round 1-4: CC_DESC_INEXACT_COMPONENTS, cc_ingest_host refused the scan): the
    slow path makes the descriptor exact, the host-buffer entry point accepts it."""
    L = oracle.L
    s = terrain_scan(13, n=7000, scale=4.0)
    od = oracle.Scan(s).desc()[0]
    assert od["n_cont"].max() > L.MAXC and od["flags"] == 1
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=1)
    offs = np.array([0, len(s)], np.int64)
    d = api.ingest_host(ctx, s, offs)
    assert not compare_desc(od, d[0], float_exact=True)
    assert d["flags"][0] == 1   # CC_DESC_TRUNCATED: the table holds the CC_MAXC largest contours of such a level
    _check(oracle, [s])         # labels and images, too


def test_min_cont_cell_cnt_above_three(oracle):
    """min_cont_cell_cnt_ = 6: components of 3..5 cells must be dropped too (stats(n,4) < min_cont_cell_cnt_,
    contour_mng.cpp:303); the kernel's saturating counters only prove >= 3, the exact areas decide."""
    cfg = oracle.L.default_manager_cfg()
    cfg.min_cont_cell_cnt = 6
    s = terrain_scan(2, n=20000, scale=1.2)
    d6 = _check(oracle, [s], cfg=cfg)
    assert d6["n_cont"].sum() < oracle.Scan(s).desc()[0]["n_cont"].sum()
    for l in range(oracle.L.NLEV):
        ns = int(d6["n_stored"][0, l])
        assert ns == 0 or d6["cont"][0, l]["cell_cnt"][:ns].min() >= 6


def test_nan_heights_are_ignored(oracle):
    """z = NaN never wins a cell nor the max/min (`bev < NaN` is false, contour_mng.h:517-524)."""
    s = terrain_scan(7, n=20000, scale=1.2)
    s[::7, 2] = np.nan
    clean = s[~np.isnan(s[:, 2])]
    d = _check(oracle, [s])
    assert np.isfinite(d["max_bin_val"]).all() and d["n_pix"][0] == oracle.Scan(clean).desc()[0]["n_pix"]


def test_resolutions_pow2_and_not(oracle):
    """reso_row_/reso_col_ = 2.0 (the paper's setting, contour_mng.h:95) takes the multiply-by-reciprocal instance of the
    rasteriser, 1.5 x 0.75 the IEEE-division instance; both must hash every point like `x / reso` does
    (hashPointToImage, contour_mng.h:448-463)."""
    s = terrain_scan(4, n=20000, scale=1.2)
    for rr, rc, n in ((2.0, 2.0, 74), (1.5, 0.75, 100)):
        cfg = oracle.L.default_manager_cfg()
        cfg.reso_row, cfg.reso_col = rr, rc
        cfg.n_row, cfg.n_col = n, n
        d = _check(oracle, [s], cfg=cfg)
        assert d["n_pix"][0] > 500


def test_large_components_go_through_the_eight_lane_walk(oracle):
    """Components of more than CC_K2_BIG = 128 cells (a plateau of 45 x 40 cells with a tower on it, a long wall, next to small
    clutter): their running sums are accumulated by eight lanes, one sum each (k_contours.h) -- the same values in the same
    order, so the contour rows stay bit-exact."""
    rng = np.random.default_rng(21)
    pts = []
    for (x0, x1, y0, y1, z0, z1, n) in ((-30, 15, -20, 20, 1.6, 2.4, 60000), (-10, 5, -5, 8, 3.1, 4.8, 15000), (20, 22, -60, 60, 2.2, 3.9, 9000),
                                        (-70, 70, -70, 70, 0.0, 5.0, 1500)):
        p = np.zeros((n, 4), np.float32)
        p[:, 0] = rng.uniform(x0, x1, n)
        p[:, 1] = rng.uniform(y0, y1, n)
        p[:, 2] = rng.uniform(z0, z1, n) - 2.0      # the reference adds lidar_height_ = 2
        pts.append(p)
    scan = np.concatenate(pts)[rng.permutation(sum(len(p) for p in pts))]
    d = _check(oracle, [scan])
    assert d["cont"]["cell_cnt"][0, 0].max() > 1000 and (d["cont"]["cell_cnt"][0, :4].max(axis=1) > 128).all() and (d["flags"] == 0).all()


def test_fewer_anchors_and_neighbours_than_the_record_holds(oracle):
    """piv_firsts_ = 4, dist_firsts_ = 8, roi_radius_ = 8 (the record has room for 6 / 10): the slots the reference does not have
    stay all-zero -- keys, BCI headers, points -- exactly like the oracle's record (found by tests/fuzz_gpu_ingest.py: the BCI
    headers of the unused anchors carried their (level, seq))."""
    cfg = oracle.L.default_manager_cfg()
    cfg.piv_firsts, cfg.dist_firsts, cfg.roi_radius, cfg.blind_sq = 4, 8, 8.0, 4.0
    d = _check(oracle, [terrain_scan(4, n=20000, scale=1.2)], cfg=cfg)
    assert (d["bcis"]["n_pts"][0][:, :4] > 0).any() and (d["bcis"]["n_pts"][0][:, 4:] == 0).all()
    assert (d["bcis"]["piv_seq"][0][:, 4:] == 0).all() and (d["bcis"]["level"][0][:, 4:] == 0).all()


def _blob_scene(seed, n_blobs=520, pitch=6):
    """Hundreds of separate little objects (3-14 cells, random shapes and heights, a few points per cell): more than CC_MAXC =
    320 components on several levels, sizes full of ties -- what the slow path of K2 (cc_k_contours_big) is for."""
    rng = np.random.default_rng(seed)
    per_row = 150 // pitch - 1
    pts = []
    for b in range(n_blobs):
        r0, c0 = 3 + pitch * (b // per_row), 3 + pitch * (b % per_row)
        if r0 + 4 >= 150:
            break
        cells = {(0, 0)}
        while len(cells) < rng.integers(3, 15):
            r, c = list(cells)[rng.integers(len(cells))]
            dr, dc = rng.integers(-1, 2, 2)
            if 0 <= r + dr < pitch - 2 and 0 <= c + dc < pitch - 2:
                cells.add((int(r + dr), int(c + dc)))
        top = rng.uniform(-0.3, 2.4)   # + lidar_height 2.0: between the lowest and above the highest level
        for (r, c) in cells:
            x, y = (r0 + r) - 75 + 0.5, (c0 + c) - 75 + 0.5
            if x * x + y * y < 16:
                continue
            for _ in range(2):
                pts.append((x + rng.uniform(-0.4, 0.4), y + rng.uniform(-0.4, 0.4), top - rng.uniform(0, 0.6) * (r + c > 1), 0.0))
    return np.asarray(pts, np.float32)


def test_more_components_than_cc_maxc_take_the_exact_slow_path(oracle):
    """The reference has no limit on cont_views_[l].size() (contour_mng.h:92-110).  A level with more than CC_MAXC = 320
    components is redone by cc_k_contours_big: labels, the 320 largest contours in std::sort's order, keys and BCIs equal
    the oracle's, flags == CC_DESC_TRUNCATED (a table that does not hold every contour), never CC_DESC_INEXACT_*."""
    scenes = [_blob_scene(1), terrain_scan(2, n=20000, scale=1.2), _blob_scene(2, n_blobs=560, pitch=5)]
    d = _check(oracle, scenes)
    assert d["n_cont"][0].max() > 320 and d["n_cont"][2].max() > 320, d["n_cont"]
    assert d["flags"][0] == 1 and d["flags"][1] == 0 and d["flags"][2] == 1, d["flags"]
    assert (d["n_stored"][0] <= 320).all()


def test_mid_path_rebuilds_the_dense_image_from_the_list(oracle, capfd, monkeypatch):
    """K1 writes the dense max-height image and the dense positions only on request (debug outputs) or when a scan's active
    cells overflow its list.  A scan that fits the list but not the list kernel's tables (more than 320 components on a
    level) reaches cc_k_contours_mid without them: it rebuilds both from the list.  `_check` ingests without debug outputs
    first, on a fresh context, and compares those descriptors too."""
    monkeypatch.setenv("CC_EMU_TRACE_K2", "1")
    scenes = [_blob_scene(3, n_blobs=420, pitch=6), _thin_terrain(11, 9000, 1.5, 1.2)]
    capfd.readouterr()
    d = _check(oracle, scenes)
    err = capfd.readouterr().err
    rebuilt = {int(l.split("scan")[1].split(":")[0]) for l in err.splitlines() if l.startswith("[k2 mid]") and "rebuilt" in l}
    assert rebuilt == {0}, (rebuilt, d["n_cont"])
    assert d["n_cont"][0].max() > 320


def test_k1_workgroups_take_several_scans(oracle, monkeypatch):
    """A many-scan launch brings one K1 workgroup per CU, and a workgroup takes scans b, b + grid, ... one after the other
    (csrc/k_rasterize.h): two workgroups for five scans here, the LDS grid re-initialised between a workgroup's scans."""
    monkeypatch.setenv("CC_K1_WGS", "2")
    scans = [terrain_scan(31, n=6000), _thin_terrain(32, 5000, 1.5, 1.2), terrain_scan(33, n=900), np.full((40, 4), 1000.0, np.float32),
             terrain_scan(34, n=12000, scale=2.0)]
    _check(oracle, scans)


# ---- the list front half of K2 (csrc/k_contours_list.h): which scans it takes, and that what it takes is bit-exact ----
def _list_trace(capfd):
    """(scans the list kernel kept, scans it handed to the mid path) from the harness build's trace lines."""
    err = capfd.readouterr().err
    seen = {int(l.split("scan")[1].split(":")[0]) for l in err.splitlines() if l.startswith("[k2 list] scan") and "active cells" in l}
    handed = {int(l.split("scan")[1].split()[0]) for l in err.splitlines() if "handed to the mid path" in l}
    return seen - handed, handed


def _thin_terrain(seed, n, scale, z_shift, quant=None):
    """A terrain scan lowered by z_shift: fewer cells above the lowest level, so that the scan fits the list kernel."""
    s = terrain_scan(seed, n=n, scale=scale, quant=quant)
    s[:, 2] -= z_shift
    return s


def test_list_kernel_takes_what_fits_and_hands_on_the_rest(oracle, capfd, monkeypatch):
    """Scans of up to 3 072 active cells / 12 800 (cell, level) slots are labelled by the list kernel, larger ones by the
    original body behind it (cc_k_contours_mid); both are compared with the oracle like every other scan."""
    monkeypatch.setenv("CC_EMU_TRACE_K2", "1")
    fits = [_thin_terrain(11, 9000, 1.5, 1.2), _thin_terrain(12, 14000, 2.5, 2.0, quant=0.25)]
    too_big = [terrain_scan(13, n=40000, scale=1.6)]
    capfd.readouterr()
    d = _check(oracle, fits + too_big)
    kept, handed = _list_trace(capfd)
    assert kept == {0, 1} and handed == {2}, (kept, handed)
    assert (d["flags"] == 0).all() and d["n_cont"][:2].max() > 8


def test_list_kernel_scan_family(oracle, capfd, monkeypatch):
    """Scans that fit the list kernel, of several kinds: smooth terrain at different densities, plateaus exactly at the level
    thresholds, long thin walls (one-cell-wide runs, many run contacts), blobs that straddle the 64-cell chunk borders and
    the row ends (column 0 / 149 cells are not neighbours of the previous / next row's last / first cell)."""
    monkeypatch.setenv("CC_EMU_TRACE_K2", "1")
    rng = np.random.default_rng(77)
    scans = [_thin_terrain(21 + i, int(rng.integers(3000, 16000)), float(rng.uniform(0.8, 3.0)), float(rng.uniform(1.4, 2.6)),
                           quant=[None, 0.1, 0.5][i % 3]) for i in range(5)]
    n = 6000
    p = np.zeros((n, 4), np.float32)  # plateau steps exactly at the thresholds (strict `>` must hold), sparse enough to fit
    p[:, :2] = rng.uniform(-40, 40, (n, 2))
    lv = np.array([1.5, 2.0, 2.5, 3.0, 3.5, 4.0], np.float32) - 2.0
    p[:, 2] = lv[rng.integers(0, 6, n)] + rng.choice([0.0, 1e-6, -1e-6], n).astype(np.float32)
    scans.append(p)
    k = 18
    a = rng.uniform(-70, 70, (k, 2))
    b = a + rng.uniform(-60, 60, (k, 2))
    i = rng.integers(0, k, 9000)
    t = rng.random(9000)[:, None]
    w = np.zeros((9000, 4), np.float32)  # walls
    w[:, :2] = a[i] * (1 - t) + b[i] * t + rng.normal(0, 0.3, (9000, 2))
    w[:, 2] = rng.integers(0, 5, 9000) * 1.0 - 0.5
    scans.append(w)
    e = np.zeros((12000, 4), np.float32)  # the map's left / right border columns and rows around the chunk borders
    e[:, 0] = rng.uniform(-74, 74, 12000)
    e[:, 1] = np.where(rng.random(12000) < 0.5, rng.uniform(-74.9, -70, 12000), rng.uniform(70, 74.9, 12000))
    e[:, 2] = rng.uniform(-1.0, 3.0, 12000)
    scans.append(e)
    capfd.readouterr()
    _check(oracle, scans)
    kept, handed = _list_trace(capfd)
    assert len(kept) >= 6, (kept, handed)   # (a dense draw may go to the mid path: still compared above)


def test_list_kernel_min_cont_cell_cnt_one_and_two(oracle, capfd, monkeypatch):
    """min_cont_cell_cnt_ of 1 and 2 (the kept test is "has a second cell" / every root); 4 and more is the mid path's."""
    monkeypatch.setenv("CC_EMU_TRACE_K2", "1")
    rng = np.random.default_rng(5)
    k = 14
    c = rng.uniform(-60, 60, (k, 2))
    i = rng.integers(0, k, 12000)
    blobs = np.zeros((12000, 4), np.float32)  # dense blobs of different heights plus a few lone cells: tens of components, not hundreds
    blobs[:, :2] = c[i] + rng.normal(0, 1, (12000, 2)) * rng.uniform(1.0, 4.0, k)[i, None]
    blobs[:, 2] = rng.uniform(-0.3, 3.0, k)[i] + rng.normal(0, 0.3, 12000)
    for mc in (1, 2):
        cfg = oracle.L.default_manager_cfg()
        cfg.min_cont_cell_cnt = mc
        capfd.readouterr()
        d = _check(oracle, [blobs], cfg=cfg)
        kept, handed = _list_trace(capfd)
        assert kept == {0} and not handed, (kept, handed, d["n_cont"])
