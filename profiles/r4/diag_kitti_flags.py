"""Diagnostic (run on the GPU box): which internal capacity, if any, do KITTI-shaped queries against a 5 k-scan DB meet?
Prints the histogram of cc_query_result_t.flags and the per-query funnel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cc_amd
cc = cc_amd.load()
n_db, B, P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 1024, 64 * 1875
w = cc.synth.World(kitti=True)
ctx = cc.Context(0, max_batch=256)
HB, FB = cc.packed_sizes()
print("packed sizes", HB, FB)
db = cc.Database(ctx, capacity=n_db + 16)
maxc = np.zeros(6, int)
for c0 in range(0, n_db, 256):
    c1 = min(c0 + 256, n_db)
    x, _, ts = cc.synth.make_sequence(c1 - c0, world=w, device="cuda", start=c0)
    d = ctx.ingest(x.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P)
    dn = cc.desc_to_numpy(d)
    maxc = np.maximum(maxc, dn["n_cont"].max(0))
    assert (dn["flags"] == 0).all(), dn["flags"].max()
    db.add_scans(d, ts, np.arange(c0, c1, dtype=np.int32))
print("max contours per level over the DB scans", maxc.tolist())
for s in range(2):
    x, _, _ = cc.synth.make_sequence(B, world=w, device="cuda", start=n_db + s * B)
    d = ctx.ingest(x.reshape(-1, 4), np.arange(B + 1, dtype=np.int64) * P)
    res = db.query(d, np.full(B, n_db, np.int32), allow_flagged=True)
    fl = res["flags"]
    print("flags histogram", {int(v): int((fl == v).sum()) for v in np.unique(fl)})
    for k in ("n_knn_hits", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy"):
        print("  %-16s mean %8.1f max %6d" % (k, res[k].mean(), res[k].max()))
    print("  loop closures", int((res["n_res"] > 0).sum()))
