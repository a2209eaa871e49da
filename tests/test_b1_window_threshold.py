"""Stage B1's circular pi/16 window (contour_mng.h:344-357) compares `(double)(f32 difference) + 2 pi * wrap` with the
range; the kernel does it in f32 with a precomputed threshold for the wrapped case (CC_B1_WRAP_T in csrc/k_check.h).
This checks the threshold against the reference's expression on the floats around it and on a random sample."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threshold():
    src = open(os.path.join(ROOT, "contour-context_amd", "csrc", "k_check.h")).read()
    m = re.search(r"#define CC_B1_WRAP_T \((-0x[0-9a-f.]+p[+-]?\d+)f\)", src)
    assert m, "CC_B1_WRAP_T not found"
    t = float.fromhex(m.group(1))
    assert np.float32(t) == t
    return np.float32(t)


def _invalid_ref(d, wrap):
    ar = np.float64(np.float32(np.pi / 16))
    return d.astype(np.float64) + 2 * np.pi * np.float64(wrap) > ar


def test_wrapped_window_threshold_is_exact():
    t = _threshold()
    bits = np.array([t], np.float32).view(np.uint32)[0]
    # the 4096 floats on either side (negative floats: larger bit pattern = more negative)
    near = (np.arange(-4096, 4097, dtype=np.int64) + int(bits)).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(5)
    far = rng.uniform(-2 * np.pi - 0.5, 0.5, 2_000_000).astype(np.float32)
    for d in (near, far):
        assert np.array_equal(_invalid_ref(d, 1), d >= t)
        assert np.array_equal(_invalid_ref(d, 0), d > np.float32(np.pi / 16))
