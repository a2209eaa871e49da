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


def _killer(n):
    """A sequence on which median-of-3 quicksort degenerates (Musser's construction), so that introsort's depth limit is
    reached and the heapsort branch runs: the wave replay must hand such input to the serial replica."""
    k = n // 2
    a = np.zeros(n, np.int32)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_wave_parallel_replay_matches_std_sort(oracle):
    """std_sort_wave (one wave per array: parallel Hoare partitions + stable rank; K2's size sort) against the real
    std::sort: random tie-heavy inputs of every length class, sorted / reversed / all-equal inputs, and inputs that drive
    introsort into its heapsort branch."""
    lib = C.CDLL(emu_api.build())
    rng = np.random.default_rng(5)
    cases = []
    for n in list(range(0, 40)) + [63, 64, 65, 100, 128, 129, 257, 320, 1000]:
        for hi in (2, 5, 40, 5000):
            cases.append(rng.integers(3, 3 + hi, n).astype(np.int32))
    for n in (17, 33, 200, 320):
        cases += [np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32)[::-1].copy(), np.full(n, 7, np.int32)]
    for n in (64, 200, 320, 2000):
        cases.append(_killer(n))
        cases.append(-_killer(n) + 5000)
    for keys in cases:
        n = len(keys)
        perm = oracle.sort_desc_perm(keys)
        arr = ((keys.astype(np.uint32) << 16) | np.arange(n, dtype=np.uint32)).copy()
        lib.emu_sort_desc_wave(arr.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(arr & 0xFFFF, perm), (n, keys[:8])


def test_order_kernel_sort_and_scans():
    """cc_k_knn_order's building blocks on the CPU harness: the workgroup bitonic sort that keeps 1024 x R keys in registers
    (shuffles inside a wave, LDS only between waves; k_knn.h: cc_block_bitonic_u32) and the in-place block scans."""
    lib = C.CDLL(emu_api.build())
    rng = np.random.default_rng(11)
    for r in (1, 4, 8):
        for kind in range(3):
            n = 1024 * r
            a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
            if kind == 1:
                a[rng.integers(0, n, n // 2)] = 0xFFFFFFFF          # padding keys
            if kind == 2:
                a = (rng.integers(0, 50, n).astype(np.uint32) << 13) | np.arange(n, dtype=np.uint32)   # many equal buckets
            exp = np.sort(a)
            lib.emu_block_bitonic(a.ctypes.data_as(C.c_void_p), r)
            assert np.array_equal(a, exp), (r, kind)
    for n in (64, 1024, 2048, 8192):
        v = rng.integers(0, 3, n).astype(np.int32)
        s = v.copy()
        lib.emu_block_scan(s.ctypes.data_as(C.c_void_p), n, 0)
        assert np.array_equal(s, np.cumsum(v))
        h = np.where(rng.random(n) < 0.05, np.arange(n), 0).astype(np.int32)
        m = h.copy()
        lib.emu_block_scan(m.ctypes.data_as(C.c_void_p), n, 1)
        assert np.array_equal(m, np.maximum.accumulate(h))
