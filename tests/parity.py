"""Shared comparison helpers: device/emulated descriptors vs the oracle's."""
import numpy as np

HDR = ["n_cont", "n_stored", "layer_cell_cnt", "max_bin_val", "min_bin_val", "n_pix", "flags"]


def compare_desc(od, d, float_exact=True, rtol=1e-5):
    """od: oracle cc_scan_desc_t record, d: product record. Returns list of mismatch strings.
    Integer fields (counts, cell_cnt, flags, BCI bits/segments, levels, seqs) must be identical, and so must every float
    that the device computes with individually rounded IEEE operations in the reference's order: the contour rows
    (moments, Eigen's 2x2 solver restated) and the BCI points (r = sqrtf, theta = glibc's atan2f restated).  Only the
    retrieval keys go through a library function whose last bit is not pinned (the f64 exp of gaussPDF): they are
    identical when float_exact, else within rtol."""
    bad = []
    for f in HDR:
        if not np.array_equal(od[f], d[f]):
            bad.append("hdr.%s oracle=%s got=%s" % (f, od[f], d[f]))
    ko, kd = od["keys"], d["keys"]
    if float_exact:
        if not np.array_equal(ko, kd, equal_nan=True):
            bad.append("keys differ (max abs %g)" % np.nanmax(np.abs(ko - kd)))
    else:
        if not (np.array_equal(np.isnan(ko), np.isnan(kd)) and np.allclose(np.nan_to_num(ko), np.nan_to_num(kd), rtol=rtol, atol=1e-6)):
            bad.append("keys differ beyond tol (max abs %g)" % np.nanmax(np.abs(ko - kd)))
    bo, bd = od["bcis"], d["bcis"]
    for f in ["dist_bin", "piv_seq", "level", "n_pts", "n_segs", "segs"]:
        if not np.array_equal(bo[f], bd[f]):
            bad.append("bci.%s differs" % f)
    for f in ["level", "seq", "bit_pos"]:
        if not np.array_equal(bo["pts"][f], bd["pts"][f]):
            bad.append("bci.pts.%s differs" % f)
    for f in ["r", "theta"]:
        a, b = bo["pts"][f], bd["pts"][f]
        if not np.array_equal(a, b):
            bad.append("bci.pts.%s differs (max abs %g, %d values)" % (f, np.max(np.abs(a - b)), int((a != b).sum())))
    for l in range(od["cont"].shape[0]):
        ns = int(od["n_stored"][l])
        a, b = od["cont"][l][:ns], d["cont"][l][:ns]
        for f in a.dtype.names:
            if f == "pad_":
                continue
            if a[f].dtype.kind in "iu":
                if not np.array_equal(a[f], b[f]):
                    bad.append("cont[%d].%s differs" % (l, f))
            else:
                if not np.array_equal(a[f], b[f]):
                    bad.append("cont[%d].%s differs (max abs %g, %d values)" % (l, f, np.max(np.abs(a[f] - b[f])), int((a[f] != b[f]).sum())))
    return bad


def terrain_scan(seed, n=60000, scale=1.6, quant=None):
    """Random smooth height field sampled by n points: many contours per level, ties when quantised."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-80, 80, (n, 2)).astype(np.float32)
    z = np.zeros(n)
    for _ in range(12):
        f = rng.uniform(0.02, 0.35, 2)
        ph = rng.uniform(0, 6.28, 2)
        z += rng.uniform(0.3, 1.0) * np.sin(f[0] * xy[:, 0] + ph[0]) * np.sin(f[1] * xy[:, 1] + ph[1])
    z = z * scale + rng.normal(0, 0.15, n)
    if quant:
        z = np.round(z / quant) * quant
    pts = np.zeros((n, 4), np.float32)
    pts[:, :2] = xy
    pts[:, 2] = z.astype(np.float32)
    return pts
