"""Size-independent properties of the HIP path at the benchmark's scale (BASELINE.json configs[1]: 120k-point scans,
multi-thousand-scan DB), where the CPU oracle is too slow to be the checker:
  * batch invariance: the result of a scan does not depend on which other scans share its launch (ingest and query);
  * the DB built by one cc_db_add_scans call equals the DB built scan by scan (sorted key view merged incrementally);
  * determinism: the same call twice gives the same bytes;
  * KNN hit lists are sorted, within the search radius, unique, and equal to a brute-force top-k over the layer's keys
    (numpy, f32, same accumulation order) under the visibility rules of the final epoch;
  * every revisit of a mapped place is reported as a loop closure with the right candidate."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_DB, N_Q = 2000, 96


@pytest.fixture(scope="module")
def world_db(cc):
    import torch
    w = cc.synth.World()
    P = 64 * 1875
    ctx = cc.Context(0, max_batch=256)
    descs = []
    for c0 in range(0, N_DB, 250):
        x, _, _ = cc.synth.make_sequence(250, world=w, device="cuda", start=c0)
        descs.append(ctx.ingest(x.reshape(-1, 4), np.arange(251, dtype=np.int64) * P))
    desc = torch.cat(descs)
    xq, _, _ = cc.synth.make_sequence(N_Q, world=w, device="cuda", start=N_DB)
    qdesc = ctx.ingest(xq.reshape(-1, 4), np.arange(N_Q + 1, dtype=np.int64) * P)
    torch.cuda.synchronize()
    yield ctx, desc, xq, qdesc, P
    ctx.close()


def _db(cc, ctx, desc, chunks):
    db = cc.Database(ctx, capacity=N_DB + 8)
    n = desc.shape[0]
    ts = np.arange(n, dtype=np.float64) / 10.0
    seeds = np.arange(n, dtype=np.int32)
    for a in range(0, n, chunks):
        b = min(a + chunks, n)
        db.add_scans(desc[a:b].contiguous(), ts[a:b], seeds[a:b])
    return db


def _same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def _desc_same(cc, ta, tb):
    """descriptors equal in everything that is defined (contour tables are only written up to n_stored)"""
    a, b = cc.desc_to_numpy(ta), cc.desc_to_numpy(tb)
    for f in a.dtype.names:
        if f == "cont":
            continue
        if np.ascontiguousarray(a[f]).tobytes() != np.ascontiguousarray(b[f]).tobytes():
            return False
    for i in range(len(a)):
        for l in range(a["cont"].shape[1]):
            n = int(a["n_stored"][i, l])
            if a["cont"][i, l, :n].tobytes() != b["cont"][i, l, :n].tobytes():
                return False
    return True


def knn_bruteforce_check(keys_by_level, dq, knn1, cnt1, ranges, n_db, qis, nnk=50, settle=600):
    """KNN hit lists of a query batch against brute force over the DB's keys (numpy, f32, the reference's accumulation
    order) under the visibility rules of the final epoch.  keys_by_level[ll]: [n_db * 6, 10] keys of query level ll in
    insertion order (scan-major); which keys are searchable is the DB's own bookkeeping (buffers vs trees), so: every key
    in a hit list must be valid, the list must be sorted, unique and inside dist_ub, and no visible key older than `settle`
    scans may be closer than the worst hit."""
    ranges = np.asarray(ranges, np.float32).reshape(3, 7)
    qlev = [1, 2, 3]
    for ll, lev in enumerate(qlev):
        K = keys_by_level[ll]
        valid = K.sum(1) != 0
        for qi in qis:
            for seq in range(6):
                k = dq["keys"][qi, lev, seq].astype(np.float32)
                m = int(cnt1[qi, ll, seq])
                if k.sum() == 0:
                    assert m == 0
                    continue
                hits = knn1[qi, ll, seq, :m]
                assert np.all(np.diff(hits["dist_sq"]) >= 0), "hits sorted by distance"
                ids = hits["gidx"].astype(np.int64) * 6 + hits["seq"]
                assert len(set(ids.tolist())) == m, "no key twice"
                assert valid[ids].all()
                # distances recomputed in f32 with the reference's accumulation order
                c = K[ids]
                dd = (k - c).astype(np.float32)
                r = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2] + dd[:, 3] * dd[:, 3]).astype(np.float32)
                r = (r + (dd[:, 4] * dd[:, 4] + dd[:, 5] * dd[:, 5] + dd[:, 6] * dd[:, 6] + dd[:, 7] * dd[:, 7]).astype(np.float32)).astype(np.float32)
                r = (r + dd[:, 8] * dd[:, 8]).astype(np.float32)
                r = (r + dd[:, 9] * dd[:, 9]).astype(np.float32)
                assert np.allclose(r, hits["dist_sq"], rtol=2e-6, atol=1e-6)
                # dist_ub (contour_db.h:733-749)
                ub = max((k[0] * 0.2) ** 2, (k[0] - k[0] / 0.8) ** 2) + max((k[1] * 0.2) ** 2, (k[1] - k[1] / 0.8) ** 2) + \
                    max((k[2] - k[2] * 0.6) ** 2, (k[2] - k[2] / 0.6) ** 2)
                assert m == 0 or hits["dist_sq"][-1] < ub * (1 + 1e-5)
                if m == nnk:
                    # nothing visible and searchable may be closer than the worst hit: check against all valid keys in the
                    # buckets the search visits that are OLD enough to sit in a tree
                    mid = int(np.searchsorted(ranges[ll], k[0], side="right") - 1)
                    mid = min(max(mid, 0), 5)
                    bk = np.searchsorted(ranges[ll], K[:, 0], side="right") - 1
                    vis = ((bk <= mid) | (bk >= 2 * mid + 1)) & (bk >= 0) & (bk <= 5) & valid
                    scan_of = np.arange(len(K)) // 6
                    dall = ((k[None, :] - K[vis]) ** 2).sum(1)
                    older = scan_of[vis] <= (n_db - 1 - settle)      # well past any insertion delay
                    better = (dall < hits["dist_sq"][-1] * (1 - 1e-5)) & older
                    got = set(ids.tolist())
                    missing = [i for i in np.nonzero(vis)[0][better] if i not in got]
                    assert not missing, "closer visible keys were not returned: %s" % missing[:5]


def test_ingest_batch_invariance_and_determinism(cc, world_db):
    import torch
    ctx, desc, xq, qdesc, P = world_db
    again = ctx.ingest(xq.reshape(-1, 4), np.arange(N_Q + 1, dtype=np.int64) * P)
    assert _desc_same(cc, again, qdesc)
    for i in (0, 17, N_Q - 1):
        one = ctx.ingest(xq[i].reshape(-1, 4).contiguous(), np.array([0, P], np.int64))
        assert _desc_same(cc, one[:1], qdesc[i:i + 1]), "scan %d differs when ingested alone" % i


def test_query_batch_invariance_incremental_db_and_knn(cc, world_db):
    import torch
    ctx, desc, xq, qdesc, P = world_db
    db1 = _db(cc, ctx, desc, N_DB)      # one add
    db2 = _db(cc, ctx, desc, 1)         # scan by scan: 2000 incremental merges of the sorted key view
    assert np.array_equal(db1.bucket_state()[0], db2.bucket_state()[0]) and np.array_equal(db1.bucket_state()[1], db2.bucket_state()[1])
    ep = np.full(N_Q, N_DB, np.int32)
    r1, knn1, cnt1 = db1.query(qdesc, ep, want_knn=True)
    r1b, _, _ = db1.query(qdesc, ep, want_knn=True)
    assert _same(r1, r1b), "same query twice"
    r2, knn2, cnt2 = db2.query(qdesc, ep, want_knn=True)
    assert _same(r1, r2) and np.array_equal(cnt1, cnt2)
    for f in ("gidx", "level", "seq", "dist_sq"):
        m = np.arange(knn1.shape[-1])[None, None, None, :] < cnt1[..., None]
        assert np.array_equal(knn1[f][m], knn2[f][m])
    cat = np.concatenate([db1.query(qdesc[a:a + 16].contiguous(), ep[a:a + 16]) for a in range(0, N_Q, 16)])
    assert _same(r1, cat), "a scan's result depends on its batch"
    # ---- KNN lists against brute force over the DB's keys (final epoch: bucket ranges from the DB itself)
    d = cc.desc_to_numpy(desc)
    dq = cc.desc_to_numpy(qdesc)
    _, ranges = db1.bucket_state()
    knn_bruteforce_check([d["keys"][:, lev].reshape(-1, 10).astype(np.float32) for lev in (1, 2, 3)], dq, knn1, cnt1, ranges, N_DB,
                         (0, 31, N_Q - 1))
    # ---- revisits are found
    assert (r1["n_res"] > 0).mean() > 0.9
    db1.close()
    db2.close()


def test_knn_tiled_equals_walk(cc, world_db, monkeypatch):
    """CC_KNN_MODE=2 (what cc_db picks by itself for layers of 60 000+ keys): 16 searches per workgroup, squared distances
    on v_mfma_f32_16x16x4_f32 as a PREFILTER, exact nanoflann-order distances for the pairs that pass.  Same hits, in the
    same order, bit-identical distances, at mixed epochs."""
    ctx, desc, xq, qdesc, P = world_db
    monkeypatch.setenv("CC_KNN_MODE", "0")
    db1 = _db(cc, ctx, desc, N_DB)
    monkeypatch.setenv("CC_KNN_MODE", "2")
    db2 = _db(cc, ctx, desc, N_DB)
    monkeypatch.delenv("CC_KNN_MODE")
    ep = np.full(N_Q, N_DB, np.int32)
    ep[::3] = N_DB // 2
    ep[1::7] = N_DB // 3
    r1, knn1, cnt1 = db1.query(qdesc, ep, want_knn=True)
    r2, knn2, cnt2 = db2.query(qdesc, ep, want_knn=True)
    assert np.array_equal(cnt1, cnt2) and cnt1.sum() > 0
    m = np.arange(knn1.shape[-1])[None, None, None, :] < cnt1[..., None]
    for f in ("gidx", "level", "seq", "dist_sq"):
        assert np.array_equal(knn1[f][m], knn2[f][m]), f
    assert _same(r1, r2)
    db1.close()
    db2.close()


@pytest.mark.parametrize("n_queries", [400, 1024])
def test_knn_tiled_equals_walk_large_chunks(cc, world_db, monkeypatch, n_queries):
    """The same comparison with chunks of 400 and 1 024 queries: cc_k_knn_order sorts 2 400 / 6 144 searches per layer with four /
    eight keys per lane in registers (k_knn.h: cc_block_bitonic_u32), and a chunk is ONE launch of the tiled search."""
    import torch
    ctx, desc, xq, qdesc, P = world_db
    reps = (n_queries + N_Q - 1) // N_Q
    q = qdesc.repeat(reps, 1)[:n_queries].contiguous()
    ep = np.full(n_queries, N_DB, np.int32)
    ep[::3] = N_DB // 2
    ep[1::7] = N_DB // 3
    out = []
    for mode in ("0", "2"):
        monkeypatch.setenv("CC_KNN_MODE", mode)
        db = _db(cc, ctx, desc, N_DB)
        out.append(db.query(q, ep, want_knn=True))
        db.close()
    monkeypatch.delenv("CC_KNN_MODE")
    (r1, knn1, cnt1), (r2, knn2, cnt2) = out
    assert np.array_equal(cnt1, cnt2) and cnt1.sum() > 0
    m = np.arange(knn1.shape[-1])[None, None, None, :] < cnt1[..., None]
    for f in ("gidx", "level", "seq", "dist_sq"):
        assert np.array_equal(knn1[f][m], knn2[f][m]), f
    assert _same(r1, r2)


def test_knn_tiled_near_ties_large_norms(cc, world_db, monkeypatch):
    """The tiled search's matrix-core value is only a filter; it must never drop a true neighbour (k_knn.h
    cc_knn_tile_slack).  Hand-made keys where that is hardest: large norms (|k| ~ 1000, so the f32 chain's absolute error is
    at its largest) and, around every query key, hundreds of DB keys whose distances differ by a few ulps -- far more
    than nnk_ of them within the radius, tied or nearly tied at the nnk-th distance.  Hit lists of CC_KNN_MODE=0 (one wave
    per search, exact) and CC_KNN_MODE=2 (tiled, prefiltered) must be identical."""
    import torch
    ctx, desc, xq, qdesc, P = world_db
    n = 600
    d = cc.desc_to_numpy(desc[:n]).copy()
    rng = np.random.default_rng(11)
    keys = d["keys"].reshape(n, 6, 6, 10)
    base = rng.uniform(150.0, 420.0, (6, 6, 10)).astype(np.float32)       # one place in key space per (level, anchor)
    for i in range(n):
        # a shell around the base key: radius r0 (1 + j 2^-21) along a random direction, j small: distances a few ulps apart
        u = rng.normal(size=(6, 6, 10))
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        r0 = 3.0 * (1.0 + rng.integers(0, 6, (6, 6, 1)) * 2.0 ** -21)
        keys[i] = (base + (u * r0).astype(np.float32)).astype(np.float32)
    keys[::7] = keys[1::7][:len(keys[::7])]                                # and exact duplicates (ties broken by key id)
    d["keys"] = keys.reshape(d["keys"].shape)
    q = d[:48].copy()
    qk = q["keys"].reshape(48, 6, 6, 10)
    qk[:] = base[None] + rng.normal(0, 0.02, qk.shape).astype(np.float32)
    q["keys"] = qk.reshape(q["keys"].shape)
    dd = torch.from_numpy(np.frombuffer(d.tobytes(), np.uint8).reshape(n, cc.DESC_BYTES).copy()).cuda()
    dq = torch.from_numpy(np.frombuffer(q.tobytes(), np.uint8).reshape(48, cc.DESC_BYTES).copy()).cuda()
    out = []
    for mode in ("0", "2"):
        monkeypatch.setenv("CC_KNN_MODE", mode)
        db = cc.Database(ctx, capacity=n + 8)
        db.add_scans(dd, np.arange(n) / 10.0, np.arange(n, dtype=np.int32))
        out.append(db.query(dq, np.full(48, n, np.int32), want_knn=True, allow_flagged=True))
        db.close()
    monkeypatch.delenv("CC_KNN_MODE")
    (r1, knn1, cnt1), (r2, knn2, cnt2) = out
    assert np.array_equal(cnt1, cnt2) and cnt1.min() == 50, (cnt1.min(), cnt1.max())   # every search is full: the radius tightened
    m = np.arange(knn1.shape[-1])[None, None, None, :] < cnt1[..., None]
    for f in ("gidx", "level", "seq", "dist_sq"):
        assert np.array_equal(knn1[f][m], knn2[f][m]), f
