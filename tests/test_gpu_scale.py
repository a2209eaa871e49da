"""BASELINE configs 3 and 5 at their DATABASE sizes (50 000 and 20 000 scans): retrieval, incremental maintenance of the
sorted key view and the packed-record import at 300 000 keys per layer.  Synthesising 50 000 full-size scans takes
minutes, so the DB is populated with the packed records of 500 real scans, repeated with jittered retrieval keys (the
retrieval structures only see keys; the checks still run on real contour tables).  Query scans are real."""
import numpy as np
import pytest

from test_gpu_properties import knn_bruteforce_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_db", [50000, 20000])
def test_knn_and_incremental_view_at_scale(cc, n_db):
    import torch
    L = cc.L
    w = cc.synth.World()
    P = 64 * 1875
    n_real, n_q = 500, 32
    ctx = cc.Context(0, max_batch=256)
    parts = []
    for c0 in range(0, n_real + n_q, 250):
        c1 = min(c0 + 250, n_real + n_q)
        x, _, _ = cc.synth.make_sequence(c1 - c0, world=w, device="cuda", start=c0)
        parts.append(ctx.ingest(x.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P))
    desc = torch.cat(parts)
    hot, feat = ctx.pack(desc[:n_real].contiguous())
    qdesc = desc[n_real:].contiguous()
    h = hot.cpu().numpy().view(L.hot_desc_dt).reshape(-1)
    rng = np.random.default_rng(3)
    reps = n_db // n_real
    hot_all = np.tile(h, reps)
    jit = (1.0 + rng.normal(0, 0.03, hot_all["keys"].shape)).astype(np.float32)
    hot_all["keys"] = np.where(hot_all["keys"] != 0, hot_all["keys"] * jit, 0).astype(np.float32)
    feat_all = feat.repeat(reps, 1)
    hot_t = torch.from_numpy(hot_all.view(np.uint8).reshape(n_db, -1)).cuda()
    ts = np.arange(n_db, dtype=np.float64) / 10.0
    seeds = np.arange(n_db, dtype=np.int32)
    # one bulk add of most of the DB, then the rest in uneven increments (merge of the sorted view, partial activation upload)
    db = cc.Database(ctx, capacity=n_db + 8)
    cut = n_db - 3000
    db.add_packed(hot_t[:cut].contiguous(), feat_all[:cut].contiguous(), ts[:cut], seeds[:cut])
    a = cut
    for step in (1, 7, 500, 1492, 1000):
        db.add_packed(hot_t[a:a + step].contiguous(), feat_all[a:a + step].contiguous(), ts[a:a + step], seeds[a:a + step])
        a += step
    assert a == n_db == len(db)
    # the same DB in one call: identical bookkeeping and identical answers
    db1 = cc.Database(ctx, capacity=n_db + 8)
    db1.add_packed(hot_t, feat_all, ts, seeds)
    assert np.array_equal(db.bucket_state()[0], db1.bucket_state()[0]) and np.array_equal(db.bucket_state()[1], db1.bucket_state()[1])
    ep = np.full(n_q, n_db, np.int32)
    r, knn, cnt = db.query(qdesc, ep, want_knn=True)
    r1, knn1, cnt1 = db1.query(qdesc, ep, want_knn=True)
    assert r.tobytes() == r1.tobytes() and np.array_equal(cnt, cnt1)
    m = np.arange(knn.shape[-1])[None, None, None, :] < cnt[..., None]
    for f in ("gidx", "level", "seq", "dist_sq"):
        assert np.array_equal(knn[f][m], knn1[f][m])
    dq = cc.desc_to_numpy(qdesc)
    _, ranges = db.bucket_state()
    keys_by_level = [hot_all["keys"][:, lev - 1].reshape(-1, 10).astype(np.float32) for lev in (1, 2, 3)]
    knn_bruteforce_check(keys_by_level, dq, knn, cnt, ranges, n_db, (0, 13, n_q - 1))
    assert (cnt > 0).any() and (r["n_knn_hits"] > 0).all()
    db.close()
    db1.close()
    ctx.close()
