"""Tuning aid: field-by-field comparison of two descriptor dumps of profiles/k2_probe.py (CC_PROBE_SAVE)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cc_amd  # noqa: E402

L = cc_amd.load().L
a = np.load(sys.argv[1]).view(L.scan_desc_dt).reshape(-1)
b = np.load(sys.argv[2]).view(L.scan_desc_dt).reshape(-1)
for f in a.dtype.names:
    if a[f].dtype.names:
        for g in a[f].dtype.names:
            x, y = a[f][g], b[f][g]
            ne = x != y
            if x.dtype.kind == "f":
                ne &= ~(np.isnan(x) & np.isnan(y))
            if ne.any():
                idx = np.argwhere(ne)
                print("%s.%s: %d differ; first at %s: %s vs %s" % (f, g, int(ne.sum()), idx[0].tolist(), x[tuple(idx[0])], y[tuple(idx[0])]))
    else:
        x, y = a[f], b[f]
        ne = x != y
        if x.dtype.kind == "f":
            ne &= ~(np.isnan(x) & np.isnan(y))
        if ne.any():
            idx = np.argwhere(ne)
            print("%s: %d differ; first at %s: %s vs %s" % (f, int(ne.sum()), idx[0].tolist(), x[tuple(idx[0])], y[tuple(idx[0])]))
print("compared %d scans" % len(a))
