"""cc_atan2f_fdlibm (csrc/cc_stats.h): the device's restatement of glibc's atan2f, used for BCI::RelativePoint::theta
(contour_mng.h:860).  Compiled for the CPU by the test harness and compared BIT FOR BIT with std::atan2(float, float) of this libm (through
the oracle library; numpy's float32 arctan2 is a SIMD routine of its own): differences of BEV coordinates, random bit
patterns, the special cases."""
import ctypes as C

import numpy as np

import emu_api


def _mine(y, x):
    lib = C.CDLL(emu_api.build())
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(y)
    lib.emu_atan2f(C.c_void_p(y.ctypes.data), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), C.c_long(len(y)))
    return out


def test_bit_identical_to_libm():
    rng = np.random.default_rng(3)
    n = 4_000_000
    # (a) what the BCI build feeds it: differences of contour centres inside the 150 x 150 BEV
    a = (rng.uniform(0, 150, n) - rng.uniform(0, 150, n)).astype(np.float32)
    b = (rng.uniform(0, 150, n) - rng.uniform(0, 150, n)).astype(np.float32)
    # (b) random bit patterns (all magnitudes, both signs), NaN / inf removed
    c = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    d = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    ok = np.isfinite(c) & np.isfinite(d)
    # (c) special cases
    sp = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 3e38, -3e38, 0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** 25, 2.0 ** -29], np.float32)
    sy, sx = np.meshgrid(sp, sp)
    y = np.ascontiguousarray(np.concatenate([a, c[ok], sy.ravel()]), np.float32)
    x = np.ascontiguousarray(np.concatenate([b, d[ok], sx.ravel()]), np.float32)
    import oracle_py
    want = np.zeros_like(y)
    oracle_py.lib().orc_atan2f(C.c_void_p(y.ctypes.data), C.c_void_p(x.ctypes.data), C.c_void_p(want.ctypes.data), C.c_long(len(y)))
    got = _mine(y, x)
    bad = np.nonzero(want.view(np.uint32) != got.view(np.uint32))[0]
    assert len(bad) == 0, [(float(y[i]), float(x[i]), float(want[i]), float(got[i])) for i in bad[:5]]


def test_acosf_bit_identical_to_libm():
    """cc_acosf_fdlibm (csrc/cc_stats.h): the orientation filter of checkConstellCorrespSim (contour_mng.h:1195-1210) compares
    acos values with pi / 6; the device library's acosf is off by an ulp now and then (round 6, fuzz drive 131409)."""
    rng = np.random.default_rng(5)
    n = 6_000_000
    a = rng.uniform(-1.0, 1.0, n).astype(np.float32)                                  # dot products of unit vectors
    b = np.clip(np.cos(rng.uniform(0, np.pi, n)).astype(np.float32) * np.float32(1.0000001), -2, 2).astype(np.float32)  # some beyond +-1: NaN
    c = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    c = c[np.isfinite(c)]
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.49999997, 0.50000006, 2.0 ** -58, 2.0 ** -57, 1.0000001, -1.0000001, 3e38], np.float32)
    x = np.ascontiguousarray(np.concatenate([a, b, c, sp]), np.float32)
    import oracle_py
    want = np.zeros_like(x)
    oracle_py.lib().orc_acosf(C.c_void_p(x.ctypes.data), C.c_void_p(want.ctypes.data), C.c_long(len(x)))
    lib = C.CDLL(emu_api.build())
    got = np.zeros_like(x)
    lib.emu_acosf(C.c_void_p(x.ctypes.data), C.c_void_p(got.ctypes.data), C.c_long(len(x)))
    same = (want.view(np.uint32) == got.view(np.uint32)) | (np.isnan(want) & np.isnan(got))
    bad = np.nonzero(~same)[0]
    assert len(bad) == 0, (len(bad), x[bad[:5]], want[bad[:5]], got[bad[:5]])
