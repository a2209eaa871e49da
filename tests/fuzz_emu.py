"""Randomised ingest parity campaign on the CPU harness (not collected by pytest; run by hand):
    python tests/fuzz_emu.py <seed0> <n_iter> [fit]
"fit": every scan is lowered / thinned at random so that most draws have few enough cells above the lowest level for K2's list
kernel (csrc/k_contours_list.h: <= 3 072 active cells); without it most draws are dense and take the original body behind it.
Every iteration draws one scan from a family of generators and compares BEV, continuous pixel positions, integer labels
and the whole descriptor of the emulated kernels with the oracle, bit for bit."""
import sys
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import emu_api  # noqa: E402
import oracle_py as oracle  # noqa: E402
from parity import compare_desc, terrain_scan  # noqa: E402


def gen(rng):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return "terrain", terrain_scan(int(rng.integers(1 << 30)), n=int(rng.integers(500, 40000)), scale=float(rng.uniform(0.5, 6)),
                                       quant=[None, 0.1, 0.5][int(rng.integers(3))])
    n = int(rng.integers(200, 30000))
    p = np.zeros((n, 4), np.float32)
    if kind == 1:  # uniform cloud
        p[:, :2] = rng.uniform(-90, 90, (n, 2))
        p[:, 2] = rng.uniform(-3, 8, n)
        return "uniform", p
    if kind == 2:  # blobs of very different sizes
        k = int(rng.integers(3, 60))
        c = rng.uniform(-70, 70, (k, 2))
        r = rng.uniform(0.5, 15, k)
        h = rng.uniform(-1, 6, k)
        i = rng.integers(0, k, n)
        p[:, :2] = c[i] + rng.normal(0, 1, (n, 2)) * r[i, None]
        p[:, 2] = h[i] + rng.normal(0, 0.4, n)
        return "blobs", p
    if kind == 3:  # points on / next to cell borders and the map border
        p[:, :2] = np.round(rng.uniform(-76, 76, (n, 2))) + rng.choice([0.0, 1e-6, -1e-6, 0.5, 0.99999, -0.99999], (n, 2))
        p[:, 2] = np.round(rng.uniform(-2, 5, n) * 2) / 2
        return "borders", p
    if kind == 4:  # walls: long thin structures, few height values (many ties)
        k = int(rng.integers(2, 25))
        a = rng.uniform(-70, 70, (k, 2))
        b = a + rng.uniform(-60, 60, (k, 2))
        i = rng.integers(0, k, n)
        t = rng.random(n)[:, None]
        p[:, :2] = a[i] * (1 - t) + b[i] * t + rng.normal(0, 0.3, (n, 2))
        p[:, 2] = rng.integers(0, 5, n) * 1.0 - 0.5
        return "walls", p
    if kind == 5:  # plateau steps exactly at the level thresholds (strict > must hold)
        p[:, :2] = rng.uniform(-75, 75, (n, 2))
        lv = np.array([1.5, 2.0, 2.5, 3.0, 3.5, 4.0], np.float32) - 2.0
        p[:, 2] = lv[rng.integers(0, 6, n)] + rng.choice([0.0, 1e-6, -1e-6], n).astype(np.float32)
        return "thresholds", p
    # duplicates + far outliers + huge values
    m = n // 3
    p[:m, :2] = rng.uniform(-40, 40, (m, 2))
    p[:m, 2] = rng.uniform(-1, 4, m)
    p[m:2 * m] = p[:m]
    p[2 * m:, :2] = rng.uniform(-1e4, 1e4, (n - 2 * m, 2))
    p[2 * m:, 2] = rng.uniform(-100, 100, n - 2 * m)
    return "dups", p[rng.permutation(n)]


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    oracle.lib()
    L = oracle.L
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=1)
    n_bad = 0
    for it in range(n_it):
        rng = np.random.default_rng(seed0 + it)
        kind, s = gen(rng)
        if len(sys.argv) > 3 and sys.argv[3] == "fit":
            s = s[rng.random(len(s)) < rng.uniform(0.15, 1.0)].copy()
            s[:, 2] -= np.float32(rng.uniform(0.5, 2.5))
        if len(s) <= 10:
            continue
        o = oracle.Scan(s)
        od = o.desc()[0]
        desc, dbg = api.ingest(ctx, s, np.array([0, len(s)], np.int64), debug=True)
        over = int(od["n_cont"].max()) > L.MAXC   # such a scan goes through the slow path and is compared like any other
        if True:
            ob, opix = o.bev()
            bad = compare_desc(od, desc[0], float_exact=True)
            ok = (np.array_equal(ob, dbg["bev"][0]) and np.array_equal(opix, dbg["pix_rc"][0]) and
                  np.array_equal(o.labels(), dbg["labels"][0]) and not bad)
            msg = "n_cont max %d%s, n_pix %d%s" % (int(od["n_cont"].max()), " (slow path)" if over else "", int(od["n_pix"]), "" if ok else "  DIFF " + str(bad[:3]))
        print("seed %d %-10s n=%6d  %s  %s" % (seed0 + it, kind, len(s), "ok " if ok else "BAD", msg), flush=True)
        n_bad += 0 if ok else 1
    print("done: %d bad of %d" % (n_bad, n_it))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
