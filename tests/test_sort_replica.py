"""cc_sort.h (device replica of libstdc++ std::sort) against the real std::sort, on tie-heavy inputs."""
import ctypes as C

import numpy as np

import emu_api


def test_desc_sort_matches_std_sort(oracle):
    lib = C.CDLL(emu_api.build())
    rng = np.random.default_rng(0)
    for n in list(range(0, 40)) + [64, 100, 257, 320]:
        for hi in (2, 5, 40, 5000):
            keys = rng.integers(3, 3 + hi, n).astype(np.int32)
            perm = oracle.sort_desc_perm(keys)
            arr = ((keys.astype(np.uint32) << 16) | np.arange(n, dtype=np.uint32)).copy()
            lib.emu_sort_desc(arr.ctypes.data_as(C.c_void_p), n)
            assert np.array_equal(arr & 0xFFFF, perm), (n, hi)


def test_sorted_and_all_equal(oracle):
    lib = C.CDLL(emu_api.build())
    for n in (17, 33, 200):
        for keys in (np.arange(n), np.arange(n)[::-1], np.full(n, 7)):
            keys = keys.astype(np.int32)
            perm = oracle.sort_desc_perm(keys)
            arr = ((keys.astype(np.uint32) << 16) | np.arange(n, dtype=np.uint32)).copy()
            lib.emu_sort_desc(arr.ctypes.data_as(C.c_void_p), n)
            assert np.array_equal(arr & 0xFFFF, perm)


def test_float_asc_sort(oracle):
    lib = C.CDLL(emu_api.build())
    rng = np.random.default_rng(1)
    dt = np.dtype([("k", "<f4"), ("idx", "<i4")])
    for n in (1, 16, 17, 40, 123, 256):
        keys = np.round(rng.uniform(-3.2, 3.2, n), 1).astype(np.float32)  # many exact ties
        perm = oracle.sort_asc_perm_f(keys)
        arr = np.zeros(n, dt)
        arr["k"] = keys
        arr["idx"] = np.arange(n)
        lib.emu_sort_asc_f(arr.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(arr["idx"], perm)
