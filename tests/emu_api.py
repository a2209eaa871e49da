"""TEST INFRASTRUCTURE: numpy-pointer driver of the cc_* C-ABI for the CPU-emulated build
(tests/emu/libcc_emu.so).  "Device" pointers are host pointers there."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "libcc_emu.so")


def build():
    srcs = [os.path.join(HERE, "emu", "emu_main.cpp"), os.path.join(HERE, "emu", "hip", "hip_runtime.h")]
    csrc = os.path.join(HERE, "..", "contour-context_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    srcs.append(os.path.join(HERE, "..", "include", "cont2_amd.h"))
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["sh", os.path.join(HERE, "emu", "build.sh")])
    return SO


# one OS thread per HIP thread: keep the grids of the list-driven kernels small on the harness (read at cc_db_create).
# Tests that run a program of their own on the harness pass this to the subprocess; it must never reach the GPU library.
SMALL_GRIDS = {"CC_B1_GRID": "6", "CC_B2_GRID": "6", "CC_GMM_GRID": "6"}


class EmuApi:
    def __init__(self, L):
        self.L = L
        self.lib = C.CDLL(build())
        self.lib.cc_last_error.restype = C.c_char_p
        self.lib.cc_packed_sizes.restype = None
        for f in ("cc_create", "cc_destroy", "cc_ingest_batch", "cc_ingest_host", "cc_db_create", "cc_db_destroy", "cc_db_size",
                  "cc_db_add_scans", "cc_db_add_scans_prepare", "cc_db_query_batch", "cc_db_query_submit", "cc_db_query_wait", "cc_db_bucket_state", "cc_db_check_hints", "cc_pack_scans", "cc_db_add_packed"):
            getattr(self.lib, f).restype = C.c_int

    def chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.cc_last_error().decode()))

    def create(self, cfg=None, max_batch=8):
        cfg = cfg or self.L.default_manager_cfg()
        h = C.c_void_p()
        self.chk(self.lib.cc_create(0, C.byref(cfg), max_batch, C.byref(h)), "cc_create")
        self._cfg = cfg
        return h

    def ingest(self, ctx, xyzi, offsets, debug=False):
        L = self.L
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        desc = np.zeros(n, L.scan_desc_dt)
        ncell = self._cfg.n_row * self._cfg.n_col
        dbg = None
        dbg_p = None
        if debug:
            dbg = {"bev": np.zeros((n, ncell), np.float32), "pix_rc": np.zeros((n, ncell, 2), np.float32),
                   "labels": np.zeros((n, L.NLEV, ncell), np.int16)}
            st = (C.c_void_p * 3)(dbg["bev"].ctypes.data, dbg["pix_rc"].ctypes.data, dbg["labels"].ctypes.data)
            dbg_p = C.cast(st, C.c_void_p)
        self.chk(self.lib.cc_ingest_batch(ctx, C.c_void_p(xyzi.ctypes.data), C.c_void_p(offsets.ctypes.data), n,
                                          C.c_void_p(desc.ctypes.data), dbg_p, None), "cc_ingest_batch")
        return (desc, dbg) if debug else desc

    def ingest_host(self, ctx, xyzi, offsets):
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        desc = np.zeros(n, self.L.scan_desc_dt)
        self.chk(self.lib.cc_ingest_host(ctx, C.c_void_p(xyzi.ctypes.data), C.c_void_p(offsets.ctypes.data), n,
                                         C.c_void_p(desc.ctypes.data)), "cc_ingest_host")
        return desc

    def db_create(self, ctx, cfg=None, cap=1024):
        for k, v in SMALL_GRIDS.items():
            os.environ.setdefault(k, v)
        cfg = cfg or self.L.default_db_cfg()
        h = C.c_void_p()
        self.chk(self.lib.cc_db_create(ctx, C.byref(cfg), cap, C.byref(h)), "cc_db_create")
        return h

    def db_add(self, db, desc, ts, seeds):
        desc = np.ascontiguousarray(desc)
        ts = np.ascontiguousarray(ts, np.float64)
        seeds = np.ascontiguousarray(seeds, np.int32)
        self.chk(self.lib.cc_db_add_scans(db, C.c_void_p(desc.ctypes.data), len(desc), C.c_void_p(ts.ctypes.data),
                                          C.c_void_p(seeds.ctypes.data), None), "cc_db_add_scans")

    def db_add_prepared(self, db, desc, ts, seeds):
        """cc_db_add_scans_prepare + cc_db_add_scans on the same (pointer, n): the streamed form of db_add."""
        desc = np.ascontiguousarray(desc)
        ts = np.ascontiguousarray(ts, np.float64)
        seeds = np.ascontiguousarray(seeds, np.int32)
        self.chk(self.lib.cc_db_add_scans_prepare(db, C.c_void_p(desc.ctypes.data), len(desc), None), "cc_db_add_scans_prepare")
        self.chk(self.lib.cc_db_add_scans(db, C.c_void_p(desc.ctypes.data), len(desc), C.c_void_p(ts.ctypes.data),
                                          C.c_void_p(seeds.ctypes.data), None), "cc_db_add_scans")

    def packed_sizes(self):
        hb, fb = C.c_size_t(), C.c_size_t()
        self.lib.cc_packed_sizes(C.byref(hb), C.byref(fb))
        return int(hb.value), int(fb.value)

    def pack(self, ctx, desc):
        """full descriptors -> (hot, feat) uint8 arrays [n, HOT_BYTES] / [n, FEAT_BYTES] (cc_pack_scans)"""
        desc = np.ascontiguousarray(desc)
        hb, fb = self.packed_sizes()
        hot = np.zeros((len(desc), hb), np.uint8)
        feat = np.zeros((len(desc), fb), np.uint8)
        self.chk(self.lib.cc_pack_scans(ctx, C.c_void_p(desc.ctypes.data), len(desc), C.c_void_p(hot.ctypes.data),
                                        C.c_void_p(feat.ctypes.data), None), "cc_pack_scans")
        return hot, feat

    def db_add_packed(self, db, hot, feat, ts, seeds):
        hot, feat = np.ascontiguousarray(hot), np.ascontiguousarray(feat)
        ts = np.ascontiguousarray(ts, np.float64)
        seeds = np.ascontiguousarray(seeds, np.int32)
        self.chk(self.lib.cc_db_add_packed(db, C.c_void_p(hot.ctypes.data), C.c_void_p(feat.ctypes.data), len(hot),
                                           C.c_void_p(ts.ctypes.data), C.c_void_p(seeds.ctypes.data), None), "cc_db_add_packed")

    def db_query(self, db, qdesc, epochs, lb=None, ub=None, want_knn=False):
        L = self.L
        if lb is None:
            lb, ub = L.default_thresholds()
        qdesc = np.ascontiguousarray(qdesc)
        epochs = np.ascontiguousarray(epochs, np.int32)
        nq = len(qdesc)
        res = np.zeros(nq, L.query_result_dt)
        knn = np.zeros((nq, 3, L.NPIV, L.KNN_MAX), L.knn_hit_dt) if want_knn else None
        cnt = np.zeros((nq, 3, L.NPIV), np.int32) if want_knn else None
        self.chk(self.lib.cc_db_query_batch(db, C.c_void_p(qdesc.ctypes.data), nq, C.c_void_p(epochs.ctypes.data), C.byref(lb),
                                            C.byref(ub), C.c_void_p(res.ctypes.data),
                                            C.c_void_p(knn.ctypes.data) if want_knn else None,
                                            C.c_void_p(cnt.ctypes.data) if want_knn else None, None), "cc_db_query_batch")
        return (res, knn, cnt) if want_knn else res

    def db_query_submit(self, db, qdesc, epochs):
        """cc_db_query_submit: returns (result array, keep-alive tuple); valid after db_query_wait."""
        L = self.L
        lb, ub = L.default_thresholds()
        qdesc = np.ascontiguousarray(qdesc)
        epochs = np.ascontiguousarray(epochs, np.int32)
        res = np.zeros(len(qdesc), L.query_result_dt)
        self.chk(self.lib.cc_db_query_submit(db, C.c_void_p(qdesc.ctypes.data), len(qdesc), C.c_void_p(epochs.ctypes.data), C.byref(lb),
                                             C.byref(ub), C.c_void_p(res.ctypes.data), None, None, None), "cc_db_query_submit")
        return res, (qdesc, epochs)

    def db_query_wait(self, db):
        self.chk(self.lib.cc_db_query_wait(db), "cc_db_query_wait")

    def check_hints(self, db, qdesc, hints, lb=None, ub=None, max_fine_opt=10):
        L = self.L
        if lb is None:
            lb, ub = L.default_thresholds()
        qdesc = np.ascontiguousarray(qdesc)
        hints = np.ascontiguousarray(hints, L.hint_dt)
        res = np.zeros(1, L.query_result_dt)
        sc = np.zeros(len(hints), L.hint_score_dt)
        self.chk(self.lib.cc_db_check_hints(db, C.c_void_p(qdesc.ctypes.data), C.c_void_p(hints.ctypes.data), len(hints), C.byref(lb),
                                            C.byref(ub), int(max_fine_opt), C.c_void_p(res.ctypes.data), C.c_void_p(sc.ctypes.data),
                                            None), "cc_db_check_hints")
        return res[0], sc

    def bucket_state(self, db):
        sizes = np.zeros((3, 6), np.int32)
        ranges = np.zeros((3, 7), np.float32)
        self.chk(self.lib.cc_db_bucket_state(db, C.c_void_p(sizes.ctypes.data), C.c_void_p(ranges.ctypes.data)), "bucket_state")
        return sizes, ranges
