"""K0: the host-side LayerDB bookkeeping of the product (bucket membership timeline, re-balancing, bucket ranges)
against the oracle's restatement of TreeBucket/LayerDB, on thousands of hand-made keys that force every bucket to
be used.  Runs the product's C-ABI through the CPU build (tests/emu)."""
import os
import numpy as np
import pytest

import emu_api


def _fake_desc(L, rng, n, key0_lo=3.0, key0_hi=90.0):
    d = np.zeros(n, L.scan_desc_dt)
    k = rng.uniform(1.0, 20.0, (n, L.NLEV, L.NPIV, L.KEY_DIM)).astype(np.float32)
    k[..., 0] = rng.uniform(key0_lo, key0_hi, (n, L.NLEV, L.NPIV)).astype(np.float32)
    # duplicates of key dimension 0 exercise the "contagious value" split search
    k[..., 0] = np.round(k[..., 0] * 4) / 4
    drop = rng.uniform(size=(n, L.NLEV, L.NPIV)) < 0.25     # all-zero keys are never stored
    k[drop] = 0
    d["keys"] = k
    # dummy (all-zero) contour rows so that the anchor lookups of the candidate checks stay in range
    d["n_cont"][:] = L.NPIV
    d["n_stored"][:] = L.NPIV
    d["layer_cell_cnt"][:] = 1
    return d


def test_bucket_timeline_matches_oracle(oracle):
    L = oracle.L
    rng = np.random.default_rng(11)
    n = 700
    desc = _fake_desc(L, rng, n)
    ts = np.cumsum(rng.uniform(0.05, 0.15, n))
    seeds = np.arange(n, dtype=np.int32)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=4)
    db = api.db_create(ctx, cap=n)
    odb = oracle.DB()
    cfg = L.default_manager_cfg()
    step = 100
    for i0 in range(0, n, step):
        api.db_add(db, desc[i0:i0 + step], ts[i0:i0 + step], seeds[i0:i0 + step])
        for i in range(i0, i0 + step):
            s = oracle.Scan.from_desc(desc[i], cfg, int_id=i)
            odb.add_scan(s, ts[i])
            odb.push_and_balance(int(seeds[i]), ts[i])
        osz, org = odb.bucket_state()
        esz, erg = api.bucket_state(db)
        assert np.array_equal(osz, esz), (i0, osz, esz)
        assert np.array_equal(org, erg), (i0, org, erg)
    assert (osz > 0).sum() >= 12, "the test should populate most buckets"


@pytest.mark.parametrize("knn_mode", [0, 2])  # K3: one wave per search | tiled (matrix-core prefilter)
def test_knn_with_buckets_matches_oracle(oracle, knn_mode, monkeypatch):
    """K3 through the C-ABI (CPU build) on a DB spread over several buckets, incl. the bucket-skip quirk of
    layerKNNSearch (src/cont2/contour_db.cpp:341-369)."""
    L = oracle.L
    rng = np.random.default_rng(5)
    n = 480
    desc = _fake_desc(L, rng, n, 3.0, 40.0)
    ts = np.arange(n) * 0.1
    seeds = np.arange(n, dtype=np.int32)
    monkeypatch.setenv("CC_KNN_MODE", str(knn_mode))
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=4)
    db = api.db_create(ctx, cap=n)
    api.db_add(db, desc, ts, seeds)
    odb = oracle.DB()
    cfg = L.default_manager_cfg()
    for i in range(n):
        odb.add_scan(oracle.Scan.from_desc(desc[i], cfg, int_id=i), ts[i])
        odb.push_and_balance(i, ts[i])
    q = _fake_desc(L, np.random.default_rng(6), 3, 3.0, 40.0)
    res, knn, cnt = api.db_query(db, q, np.full(3, n, np.int32), want_knn=True)
    for k in range(3):
        ores, oknn, ocnt = odb.query(oracle.Scan.from_desc(q[k], cfg, int_id=10000 + k), want_knn=True)
        assert np.array_equal(ocnt, cnt[k])
        assert ocnt.sum() > 0
        for ll in range(3):
            for seq in range(6):
                m = ocnt[ll, seq]
                a, b = oknn[ll, seq, :m], knn[k, ll, seq, :m]
                assert np.array_equal(a["dist_sq"], b["dist_sq"])
                assert np.array_equal(a["gidx"], b["gidx"]) and np.array_equal(a["seq"], b["seq"])


@pytest.mark.parametrize("knn_mode", [0, 2])  # K3: one wave per search | tiled (matrix-core prefilter)
def test_knn_crowded_layer_matches_oracle(oracle, knn_mode, monkeypatch):
    """Thousands of near-identical keys: every 64-key step of a search passes the radius test, so the pending candidate
    list grows to 2 * nnk - 1 + 64 entries before it is tightened and the bitonic sort pads it to 256 (the LDS buffer
    must hold the padded width; found by an ASAN run of this harness in round 1's review)."""
    L = oracle.L
    rng = np.random.default_rng(21)
    n = 450
    desc = _fake_desc(L, rng, n)
    base = rng.uniform(8.0, 12.0, L.KEY_DIM).astype(np.float32)
    k = (base[None, None, None, :] + rng.normal(0, 0.05, (n, L.NLEV, L.NPIV, L.KEY_DIM))).astype(np.float32)
    desc["keys"] = k
    ts = np.arange(n) * 0.1
    seeds = np.arange(n, dtype=np.int32)
    cfg = L.default_manager_cfg()
    for nnk in (64,):  # the capacity case: 2 * 64 - 1 + 64 pending candidates before a tightening
        dcfg = L.default_db_cfg()
        dcfg.nnk = nnk
        monkeypatch.setenv("CC_KNN_MODE", str(knn_mode))
        api = emu_api.EmuApi(L)
        ctx = api.create(max_batch=4)
        db = api.db_create(ctx, dcfg, cap=n)
        api.db_add(db, desc, ts, seeds)
        odb = oracle.DB(dcfg)
        for i in range(n):
            odb.add_scan(oracle.Scan.from_desc(desc[i], cfg, int_id=i), ts[i])
            odb.push_and_balance(i, ts[i])
        q = desc[[3, 420]].copy()
        q["keys"] += np.float32(0.01)
        res, knn, cnt = api.db_query(db, q, np.full(2, n, np.int32), want_knn=True)
        for kq in range(2):
            ores, oknn, ocnt = odb.query(oracle.Scan.from_desc(q[kq], cfg, int_id=10000 + kq), want_knn=True)
            assert np.array_equal(ocnt, cnt[kq]) and ocnt.min() == nnk
            for ll in range(3):
                for seq in range(6):
                    m = ocnt[ll, seq]
                    a, b = oknn[ll, seq, :m], knn[kq, ll, seq, :m]
                    assert np.array_equal(a["dist_sq"], b["dist_sq"])
                    assert np.array_equal(a["gidx"], b["gidx"]) and np.array_equal(a["seq"], b["seq"])


def test_knn_tile_full_group_every_pair_passes(oracle, monkeypatch):
    """The tiled K3 with a FULL group of 16 searches over near-identical keys: until the first radius is set every
    (search, key) pair of a 64-key step passes the prefilter -- 1024 pairs per wave and round, the most a wave's work list
    ever has to take."""
    knn_mode = 2
    L = oracle.L
    rng = np.random.default_rng(5)
    n = 320
    desc = _fake_desc(L, rng, n)
    base = rng.uniform(8.0, 12.0, L.KEY_DIM).astype(np.float32)
    desc["keys"] = (base[None, None, None, :] + rng.normal(0, 0.05, (n, L.NLEV, L.NPIV, L.KEY_DIM))).astype(np.float32)
    ts = np.arange(n) * 0.1
    cfg = L.default_manager_cfg()
    dcfg = L.default_db_cfg()
    dcfg.nnk = 50
    monkeypatch.setenv("CC_KNN_MODE", str(knn_mode))
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=16)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, desc, ts, np.arange(n, dtype=np.int32))
    odb = oracle.DB(dcfg)
    for i in range(n):
        odb.add_scan(oracle.Scan.from_desc(desc[i], cfg, int_id=i), ts[i])
        odb.push_and_balance(i, ts[i])
    qi = rng.choice(n, 16, replace=False)
    q = desc[qi].copy()
    q["keys"] += np.float32(0.01)
    res, knn, cnt = api.db_query(db, q, np.full(16, n, np.int32), want_knn=True)
    for kq in range(0, 16, 5):
        ores, oknn, ocnt = odb.query(oracle.Scan.from_desc(q[kq], cfg, int_id=10000 + kq), want_knn=True)
        assert np.array_equal(ocnt, cnt[kq]) and ocnt.min() == dcfg.nnk
        for ll in range(3):
            for seq in range(6):
                m = ocnt[ll, seq]
                a, b = oknn[ll, seq, :m], knn[kq, ll, seq, :m]
                assert np.array_equal(a["dist_sq"], b["dist_sq"]) and np.array_equal(a["gidx"], b["gidx"])


def test_keys_a_bucket_has_not_indexed_yet_are_not_found(oracle, monkeypatch):
    """The reference rebuilds a bucket's kd-tree only when popBufferMax moves something out of that bucket's buffer
    (contour_db.h:109-143): a re-balance that hands a bucket its neighbour's slice without such a pop leaves the slice outside
    the index -- and a bucket that has never popped has no tree at all (contour_db.cpp:387) -- until the bucket's next
    rebuild.  The host bookkeeping carries the indexed interval of every bucket per epoch (cc_hostdb.h), the searches test
    keys against it.  Fixture: the retrieval keys of a 78-scan drive of the randomised GPU campaign (tests/fuzz_gpu_query.py
    seed 12046, short DB delays) in which layer 0's third bucket holds 53 keys and no tree at the epochs of the last scans: the
    reference finds 0 neighbours for a key that has 30 within its radius.  Hit counts of every search of every scan, walk and
    tiled, against the oracle's replay (whose trees are the reference's own nanoflann, oracle/_ref) recorded in the fixture."""
    L = oracle.L
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn_unindexed_bucket_fixture.npz"))
    n = len(z["ts"])
    desc = np.zeros(n, L.scan_desc_dt)
    desc["keys"] = z["keys"]
    d = L.default_db_cfg()
    cfg = z["cfg"]
    d.min_elapse, d.max_elapse, d.nnk, d.max_fine_opt, d.n_q_levels = cfg[0], cfg[1], int(cfg[2]), int(cfg[3]), int(cfg[4])
    for i in range(3):
        d.q_levels[i] = int(cfg[5 + i])
    ts, seeds = z["ts"], z["seeds"]
    ocnt = z["knn_cnt"]      # the oracle's counts on the drive's full descriptors (tests/golden/make_knn_unindexed_fixture.py)
    assert ocnt[65, 0, 0] == 0 and ocnt[64, 0].sum() > 50        # the case the fixture is about
    for mode in ("0", "2"):
        monkeypatch.setenv("CC_KNN_MODE", mode)
        api = emu_api.EmuApi(L)
        ctx = api.create(max_batch=8)
        db = api.db_create(ctx, d, cap=n)
        api.db_add(db, desc, ts, seeds)
        for c0 in range(0, n, 26):
            q = np.arange(c0, min(c0 + 26, n), dtype=np.int32)
            _, knn, cnt = api.db_query(db, desc[q], q, want_knn=True)
            assert np.array_equal(cnt, ocnt[q]), (mode, c0, np.argwhere(cnt != ocnt[q])[:5])
