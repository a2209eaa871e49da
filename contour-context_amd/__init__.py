"""contour-context_amd: MI355X-native hot path of lewisjiang/contour-context (cont2contops).

Python host glue over the C-ABI shared library `libcont2_amd.so` (include/cont2_amd.h).  PyTorch is
used only for device memory, streams and torch.distributed plumbing; all compute is in the HIP
kernels under csrc/.  There is no CPU fallback: creating a Context without a HIP device fails.

The directory name carries a hyphen (it mirrors the reference's name), so it is loaded by path:
    import importlib.util; spec = importlib.util.spec_from_file_location("contour_context_amd", ".../__init__.py")
`load()` in the repo-root helper `cc_amd.py` does exactly that.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
import layouts as L  # noqa: E402
import synth  # noqa: E402,F401
import sharding  # noqa: E402,F401

LIB_PATH = os.path.join(_HERE, "libcont2_amd.so")
_SRCS = ["cont2_amd.hip", "cc_dev.h", "cc_group.h", "cc_hostcfg.h", "cc_sort.h", "cc_stats.h", "cc_fmath.h", "k_rasterize.h", "k_contours.h", "k_contours_list.h",
         "k_knn.h", "k_check.h", "k_merge.h", "k_gmm.h", "cc_hostdb.h", "cc_db_api.inc", "cc_comm.inc"]


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", s) for s in _SRCS] + [os.path.join(_HERE, "..", "include", "cont2_amd.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", os.path.join(_HERE, "csrc", "cont2_amd.hip"), "-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None

# every symbol include/cont2_amd.h declares
EXPORTS = ["cc_last_error", "cc_version", "cc_default_manager_cfg", "cc_default_db_cfg", "cc_default_thresholds",
           "cc_create", "cc_destroy", "cc_ingest_batch", "cc_ingest_host", "cc_ingest_host_bev", "cc_db_create", "cc_db_destroy", "cc_db_size",
           "cc_db_add_scans", "cc_db_add_scans_prepare", "cc_db_query_batch", "cc_db_query_submit", "cc_db_query_wait", "cc_db_hot_ptr", "cc_db_feat_ptr", "cc_pack_scans", "cc_db_add_packed",
           "cc_packed_sizes", "cc_db_bucket_state", "cc_est_sens_tf",
           "cc_profile_enable", "cc_profile_read", "cc_db_profile_enable", "cc_db_profile_read",
           "cc_db_add_scan_host", "cc_db_query_host", "cc_db_set_lanes",
           "cc_db_add_scans_host", "cc_db_query_batch_host", "cc_db_check_hints", "cc_db_check_hints_host", "cc_db_debug_passes",
           "cc_stage_points", "cc_stage_points_slot", "cc_stage_points_cancel", "cc_scan_ingest", "cc_scan_desc", "cc_scan_bev", "cc_scan_offload", "cc_scan_on_device", "cc_scan_release", "cc_db_query_scan",
           "cc_db_add_scan", "cc_db_query_scan_submit", "cc_db_query_collect", "cc_db_add_scan_prepare", "cc_runtime_init", "cc_scan_ingest_batch", "cc_scan_ready", "cc_db_add_scan_batch", "cc_db_query_scan_batch_submit",
           "cc_comm_unique_id", "cc_comm_create", "cc_comm_create_from_env", "cc_comm_rank", "cc_comm_world", "cc_comm_allgather_packed", "cc_comm_destroy"]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcont2_amd.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        _lib = C.CDLL(os.environ.get("CC_AMD_LIB") or LIB_PATH)  # CC_AMD_LIB: tuning aid (another build of the same library)
        _lib.cc_last_error.restype = C.c_char_p
        _lib.cc_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.cc_destroy.argtypes = [C.c_void_p]
        _lib.cc_ingest_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_ingest_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.cc_ingest_host_bev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.cc_db_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.cc_db_destroy.argtypes = [C.c_void_p]
        _lib.cc_db_size.argtypes = [C.c_void_p]
        _lib.cc_db_add_scans.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_add_scans_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.cc_db_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        _lib.cc_db_query_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        _lib.cc_db_query_wait.argtypes = [C.c_void_p]
        for f in ("cc_db_hot_ptr", "cc_db_feat_ptr"):
            getattr(_lib, f).argtypes = [C.c_void_p]
            getattr(_lib, f).restype = C.c_void_p
        _lib.cc_packed_sizes.argtypes = [C.c_void_p, C.c_void_p]
        _lib.cc_packed_sizes.restype = None
        _lib.cc_pack_scans.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_add_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_bucket_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_est_sens_tf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.cc_profile_enable.argtypes = [C.c_void_p, C.c_int]
        _lib.cc_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_profile_enable.argtypes = [C.c_void_p, C.c_int]
        _lib.cc_db_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_check_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.cc_db_check_hints_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_void_p]
        _lib.cc_db_set_lanes.argtypes = [C.c_void_p, C.c_int]
    return _lib


class CCError(RuntimeError):
    rc = 0


CC_ECAPACITY = -4


def _chk(rc, what, tolerate=()):
    if rc != 0 and rc not in tolerate:
        e = CCError("%s failed (%d): %s" % (what, rc, lib().cc_last_error().decode()))
        e.rc = rc
        raise e
    return rc


class IngestDebug(C.Structure):
    _fields_ = [("d_bev", C.c_void_p), ("d_pix_rc", C.c_void_p), ("d_labels", C.c_void_p)]


DESC_BYTES = L.scan_desc_dt.itemsize


def comm_from_env():
    """cc_comm_create_from_env: (handle, rank, world) from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT (one node)."""
    h, r, w = C.c_void_p(), C.c_int(), C.c_int()
    lib().cc_comm_create_from_env.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(lib().cc_comm_create_from_env(C.byref(h), C.byref(r), C.byref(w)), "cc_comm_create_from_env")
    return h, r.value, w.value


def comm_allgather(comm, d_send, d_recv, bytes_per_rank, stream):
    lib().cc_comm_allgather_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    _chk(lib().cc_comm_allgather_packed(comm, d_send, d_recv, bytes_per_rank, stream), "cc_comm_allgather_packed")


def comm_destroy(comm):
    lib().cc_comm_destroy.argtypes = [C.c_void_p]
    lib().cc_comm_destroy(comm)


def packed_sizes():
    hb, fb = C.c_size_t(), C.c_size_t()
    lib().cc_packed_sizes(C.byref(hb), C.byref(fb))
    return int(hb.value), int(fb.value)


class Context:
    """cc_ctx: per-device ingest context (ContourManager constructor's role, contour_mng.h:478-498)."""

    def __init__(self, device=0, cfg=None, max_batch=512):
        import torch
        if not torch.cuda.is_available():
            raise CCError("no HIP device: the product path has no CPU fallback")
        self.cfg = cfg or L.default_manager_cfg()
        self.device = device
        self.max_batch = max_batch
        h = C.c_void_p()
        _chk(lib().cc_create(device, C.addressof(self.cfg), max_batch, C.byref(h)), "cc_create")
        self.h = h
        self.n_cell = self.cfg.n_row * self.cfg.n_col

    def close(self):
        if getattr(self, "h", None):
            lib().cc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ingest(self, xyzi, offsets, out=None, debug=False):
        """xyzi: torch float32 CUDA tensor [total_points, 4]; offsets: int64 host array [n+1].
        Returns a torch uint8 CUDA tensor [n, DESC_BYTES] (array of cc_scan_desc_t) (+ debug dict)."""
        import torch
        assert xyzi.is_cuda and xyzi.dtype == torch.float32 and xyzi.is_contiguous()
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        if out is None:
            out = torch.empty((n, DESC_BYTES), dtype=torch.uint8, device=xyzi.device)
        dbg_p, dbg = None, None
        if debug:
            dbg = {"bev": torch.empty((n, self.n_cell), dtype=torch.float32, device=xyzi.device),
                   "pix_rc": torch.empty((n, self.n_cell, 2), dtype=torch.float32, device=xyzi.device),
                   "labels": torch.empty((n, L.NLEV, self.n_cell), dtype=torch.int16, device=xyzi.device)}
            st = IngestDebug(dbg["bev"].data_ptr(), dbg["pix_rc"].data_ptr(), dbg["labels"].data_ptr())
            dbg_p = C.addressof(st)
        stream = torch.cuda.current_stream(xyzi.device).cuda_stream
        _chk(lib().cc_ingest_batch(self.h, xyzi.data_ptr(), offsets.ctypes.data, n, out.data_ptr(), dbg_p, stream),
             "cc_ingest_batch")
        return (out, dbg) if debug else out

    def pack(self, desc):
        """Full descriptors (torch uint8 CUDA [n, DESC_BYTES]) -> (hot [n, HOT_BYTES], feat [n, FEAT_BYTES]): the compact
        per-scan records the database keeps and the ranks exchange (59 KB instead of 169 KB per scan)."""
        import torch
        n = desc.shape[0]
        hb, fb = packed_sizes()
        hot = torch.empty((n, hb), dtype=torch.uint8, device=desc.device)
        feat = torch.empty((n, fb), dtype=torch.uint8, device=desc.device)
        stream = torch.cuda.current_stream(desc.device).cuda_stream
        _chk(lib().cc_pack_scans(self.h, desc.data_ptr(), n, hot.data_ptr(), feat.data_ptr(), stream), "cc_pack_scans")
        return hot, feat

    def ingest_host(self, xyzi, offsets):
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = len(offsets) - 1
        out = np.zeros(n, L.scan_desc_dt)
        _chk(lib().cc_ingest_host(self.h, xyzi.ctypes.data, offsets.ctypes.data, n, out.ctypes.data), "cc_ingest_host")
        return out


def desc_to_numpy(desc_tensor):
    """torch uint8 [n, DESC_BYTES] (any device) -> numpy structured array of cc_scan_desc_t."""
    return desc_tensor.cpu().numpy().view(L.scan_desc_dt).reshape(-1)


class Database:
    """cc_db: device-resident ContourDB (contour_db.h:673-845) + batched queryRangedKNN."""

    def __init__(self, ctx, cfg=None, capacity=8192):
        self.ctx = ctx
        self.cfg = cfg or L.default_db_cfg()
        h = C.c_void_p()
        _chk(lib().cc_db_create(ctx.h, C.addressof(self.cfg), capacity, C.byref(h)), "cc_db_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().cc_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return lib().cc_db_size(self.h)

    def add_scans(self, desc, ts, seeds):
        """desc: torch uint8 CUDA [n, DESC_BYTES]; ts float64 [n]; seeds int32 [n] (the reference passes the scan's
        assigned seq to pushAndBalance, batch_bin_test.cpp:236)."""
        import torch
        ts = np.ascontiguousarray(ts, np.float64)
        seeds = np.ascontiguousarray(seeds, np.int32)
        n = desc.shape[0]
        assert desc.is_cuda and desc.dtype == torch.uint8 and desc.is_contiguous() and len(ts) == n and len(seeds) == n
        stream = torch.cuda.current_stream(desc.device).cuda_stream
        _chk(lib().cc_db_add_scans(self.h, desc.data_ptr(), n, ts.ctypes.data, seeds.ctypes.data, stream), "cc_db_add_scans")

    def add_scans_prepare(self, desc):
        """Queue the device half of add_scans(desc, ...) on the current stream without waiting (cc_db_add_scans_prepare)."""
        import torch
        assert desc.is_cuda and desc.dtype == torch.uint8 and desc.is_contiguous()
        stream = torch.cuda.current_stream(desc.device).cuda_stream
        _chk(lib().cc_db_add_scans_prepare(self.h, desc.data_ptr(), desc.shape[0], stream), "cc_db_add_scans_prepare")

    def query(self, qdesc, epochs, lb=None, ub=None, want_knn=False, allow_flagged=False):
        """qdesc: torch uint8 CUDA [nq, DESC_BYTES]; epochs int32 [nq] (DB state each query sees).
        Returns numpy structured array of cc_query_result_t (+ knn hits / counts as torch tensors).
        allow_flagged: a query that met an internal capacity (cc_query_result_t.flags != 0) makes the library return
        CC_ECAPACITY with every result delivered; True hands the results back (the caller looks at `flags`) instead of raising."""
        import torch
        if lb is None:
            lb, ub = L.default_thresholds()
        epochs = np.ascontiguousarray(epochs, np.int32)
        nq = qdesc.shape[0]
        assert qdesc.is_cuda and qdesc.is_contiguous() and len(epochs) == nq
        res = np.zeros(nq, L.query_result_dt)
        knn = cnt = None
        if want_knn:
            knn = torch.zeros((nq, L.NQLEV, L.NPIV, L.KNN_MAX, L.knn_hit_dt.itemsize), dtype=torch.uint8, device=qdesc.device)
            cnt = torch.zeros((nq, L.NQLEV, L.NPIV), dtype=torch.int32, device=qdesc.device)
        stream = torch.cuda.current_stream(qdesc.device).cuda_stream
        _chk(lib().cc_db_query_batch(self.h, qdesc.data_ptr(), nq, epochs.ctypes.data, C.addressof(lb), C.addressof(ub),
                                     res.ctypes.data, knn.data_ptr() if want_knn else None,
                                     cnt.data_ptr() if want_knn else None, stream), "cc_db_query_batch",
             tolerate=(CC_ECAPACITY,) if allow_flagged else ())
        if want_knn:
            return res, knn.cpu().numpy().view(L.knn_hit_dt).reshape(nq, L.NQLEV, L.NPIV, L.KNN_MAX), cnt.cpu().numpy()
        return res

    def query_submit(self, qdesc, epochs, lb=None, ub=None):
        """Asynchronous form of query(): queues the batch and returns the result array, which is only valid after
        query_wait() (cc_db_query_submit / cc_db_query_wait).  qdesc may be overwritten by work queued afterwards on the
        current stream."""
        import torch
        if lb is None:
            lb, ub = L.default_thresholds()
        epochs = np.ascontiguousarray(epochs, np.int32)
        nq = qdesc.shape[0]
        assert qdesc.is_cuda and qdesc.is_contiguous() and len(epochs) == nq
        res = np.zeros(nq, L.query_result_dt)
        self._pending = getattr(self, "_pending", [])
        self._pending.append(res)  # the library writes into it until query_wait
        stream = torch.cuda.current_stream(qdesc.device).cuda_stream
        _chk(lib().cc_db_query_submit(self.h, qdesc.data_ptr(), nq, epochs.ctypes.data, C.addressof(lb), C.addressof(ub),
                                      res.ctypes.data, None, None, stream), "cc_db_query_submit")
        return res

    def query_wait(self):
        _chk(lib().cc_db_query_wait(self.h), "cc_db_query_wait")
        self._pending = []

    def add_packed(self, hot, feat, ts, seeds):
        """hot / feat: torch uint8 CUDA [n, HOT_BYTES] / [n, FEAT_BYTES] as produced by Context.pack (possibly gathered from
        other ranks).  Same effect as add_scans on the full descriptors."""
        import torch
        ts = np.ascontiguousarray(ts, np.float64)
        seeds = np.ascontiguousarray(seeds, np.int32)
        n = hot.shape[0]
        assert hot.is_cuda and feat.is_cuda and hot.is_contiguous() and feat.is_contiguous() and feat.shape[0] == n
        assert len(ts) == n and len(seeds) == n
        stream = torch.cuda.current_stream(hot.device).cuda_stream
        _chk(lib().cc_db_add_packed(self.h, hot.data_ptr(), feat.data_ptr(), n, ts.ctypes.data, seeds.ctypes.data, stream),
             "cc_db_add_packed")

    def check_hints(self, qdesc, hints, lb=None, ub=None, max_fine_opt=10):
        """CandidateManager driven by explicit hints (checkCandWithHint in the given order, tidyUpCandidates, fineOptimize).
        qdesc: torch uint8 CUDA [DESC_BYTES] of the query scan; hints: array of L.hint_dt (cand_gidx = DB index).
        Returns (cc_query_result_t record, per-hint L.hint_score_dt array)."""
        import torch
        if lb is None:
            lb, ub = L.default_thresholds()
        hints = np.ascontiguousarray(hints, L.hint_dt)
        qdesc = qdesc.reshape(-1)
        assert qdesc.is_cuda and qdesc.is_contiguous() and qdesc.numel() == DESC_BYTES
        res = np.zeros(1, L.query_result_dt)
        sc = np.zeros(len(hints), L.hint_score_dt)
        stream = torch.cuda.current_stream(qdesc.device).cuda_stream
        _chk(lib().cc_db_check_hints(self.h, qdesc.data_ptr(), hints.ctypes.data, len(hints), C.addressof(lb), C.addressof(ub),
                                     int(max_fine_opt), res.ctypes.data, sc.ctypes.data, stream), "cc_db_check_hints")
        return res[0], sc

    def debug_passes(self, cap=1152):
        """Constellations of the last check_hints call that passed all gates: numpy array of L.pass_dbg_dt."""
        out = np.zeros(cap, L.pass_dbg_dt)
        n = C.c_int()
        lib().cc_db_debug_passes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _chk(lib().cc_db_debug_passes(self.h, out.ctypes.data, cap, C.byref(n)), "cc_db_debug_passes")
        return out[:n.value]

    def set_lanes(self, n):
        """1 = query chunks one after the other, 2 (default) = two 256-query chunks in flight on internal streams."""
        _chk(lib().cc_db_set_lanes(self.h, int(n)), "cc_db_set_lanes")

    def bucket_state(self):
        sizes = np.zeros((3, 6), np.int32)
        ranges = np.zeros((3, 7), np.float32)
        _chk(lib().cc_db_bucket_state(self.h, sizes.ctypes.data, ranges.ctypes.data), "cc_db_bucket_state")
        return sizes, ranges
