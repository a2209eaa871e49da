"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper around oracle/libcont2_oracle.so (the CPU restatement of the reference hot path) and
oracle/_ref/libref_knn.so (the real vendored nanoflann of the reference, when it was built).
Import this only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def _load_layouts():
    spec = importlib.util.spec_from_file_location("cc_layouts", os.path.join(_ROOT, "contour-context_amd", "layouts.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


L = _load_layouts()


def build(force=False):
    so = os.path.join(_HERE, "libcont2_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cont2_oracle.cpp", "orc_math.h", "orc_contour.h", "orc_gmm.h", "orc_db.h")]
    srcs.append(os.path.join(_ROOT, "include", "cont2_amd.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_scan_create.restype = C.c_void_p
        _lib.orc_scan_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_db_create.restype = C.c_void_p
        _lib.orc_db_create.argtypes = [C.c_void_p]
        for name in ("orc_scan_free", "orc_db_free"):
            getattr(_lib, name).argtypes = [C.c_void_p]
            getattr(_lib, name).restype = None
        _lib.orc_scan_from_desc.restype = C.c_void_p
        _lib.orc_scan_from_desc.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_scan_export.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_scan_bev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_scan_labels.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_scan_clear_image.argtypes = [C.c_void_p]
        _lib.orc_scan_ncont.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_db_add_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
        _lib.orc_db_push_and_balance.argtypes = [C.c_void_p, C.c_int, C.c_double]
        _lib.orc_db_bucket_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_db_timers.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_set_variant.argtypes = [C.c_uint, C.c_int, C.c_double, C.c_double]
        _lib.orc_db_query.argtypes = [C.c_void_p] * 7
        _lib.orc_run_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9
        _lib.orc_ingest_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_check_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        _lib.orc_check_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_gmm.argtypes = [C.c_void_p] * 7
        _lib.orc_gmm_eval.argtypes = [C.c_void_p] * 7
        _lib.orc_gmm_trace.argtypes = [C.c_void_p] * 5
        _lib.orc_umeyama.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_knn_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        _lib.orc_eigen2f.argtypes = [C.c_void_p] * 3
        _lib.orc_sort_desc_perm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_ccl8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_sort_asc_perm_f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scan:
    """ContourManager of one scan (ctor + makeBEV + makeContoursRecurs)."""

    @classmethod
    def from_desc(cls, desc, cfg=None, int_id=0):
        """Oracle-side ContourManager rebuilt from one cc_scan_desc_t record (numpy structured scalar/array of 1)."""
        self = cls.__new__(cls)
        self.cfg = cfg or L.default_manager_cfg()
        d = np.ascontiguousarray(np.asarray(desc).reshape(1))
        self.h = lib().orc_scan_from_desc(_p(d), C.addressof(self.cfg), int_id)
        self.ncell = self.cfg.n_row * self.cfg.n_col
        return self

    def __init__(self, xyzi, cfg=None, int_id=0, keep_cells=True):
        self.cfg = cfg or L.default_manager_cfg()
        xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
        self.h = lib().orc_scan_create(_p(xyzi), xyzi.shape[0], C.addressof(self.cfg), int_id, int(keep_cells))
        if not self.h:
            raise ValueError("scan rejected (<= 10 points)")
        self.ncell = self.cfg.n_row * self.cfg.n_col

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_scan_free(self.h)
            self.h = None

    def desc(self):
        d = np.zeros(1, dtype=L.scan_desc_dt)
        lib().orc_scan_export(self.h, _p(d))
        return d

    def bev(self):
        bev = np.zeros(self.ncell, np.float32)
        rc = np.zeros((self.ncell, 2), np.float32)
        lib().orc_scan_bev(self.h, _p(bev), _p(rc))
        return bev, rc

    def labels(self):
        lab = np.zeros((L.NLEV, self.ncell), np.int16)
        lib().orc_scan_labels(self.h, _p(lab))
        return lab

    def ncont(self, level):
        return lib().orc_scan_ncont(self.h, level)

    def clear_image(self):
        lib().orc_scan_clear_image(self.h)


class DB:
    def __init__(self, cfg=None):
        self.cfg = cfg or L.default_db_cfg()
        self.h = lib().orc_db_create(C.addressof(self.cfg))
        self.scans = []

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_db_free(self.h)
            self.h = None

    def add_scan(self, scan, ts):
        self.scans.append(scan)  # keep alive
        lib().orc_db_add_scan(self.h, scan.h, float(ts))

    def push_and_balance(self, seed, ts):
        lib().orc_db_push_and_balance(self.h, int(seed), float(ts))

    def timers(self, reset=False):
        """Accumulated reference stage timers of the queries so far: {"KNN search", "Constell", "L2 opt"} in seconds."""
        t = np.zeros(3, np.float64)
        lib().orc_db_timers(self.h, _p(t), 1 if reset else 0)
        return {"KNN search": float(t[0]), "Constell": float(t[1]), "L2 opt": float(t[2])}

    def bucket_state(self):
        sizes = np.zeros((3, 6), np.int32)
        ranges = np.zeros((3, 7), np.float32)
        lib().orc_db_bucket_state(self.h, _p(sizes), _p(ranges))
        return sizes, ranges

    def query(self, scan, lb=None, ub=None, want_knn=False):
        if lb is None:
            lb, ub = L.default_thresholds()
        res = np.zeros(1, L.query_result_dt)
        knn = np.zeros((3, L.NPIV, L.KNN_MAX), L.knn_hit_dt) if want_knn else None
        cnt = np.zeros((3, L.NPIV), np.int32) if want_knn else None
        lib().orc_db_query(self.h, scan.h, C.addressof(lb), C.addressof(ub), _p(res), _p(knn), _p(cnt))
        return (res[0], knn, cnt) if want_knn else res[0]


def set_variant(label_shuffle_seed=0, lbfgs_max_iterations=10, wolfe_c1=1e-4, wolfe_c2=0.9):
    """Sensitivity knobs of the pieces restated from third-party code (component numbering, Ceres line search); the
    defaults are the restatement itself.  Tests only."""
    lib().orc_set_variant(int(label_shuffle_seed), int(lbfgs_max_iterations), float(wolfe_c1), float(wolfe_c2))


def run_sequence(xyzi, offsets, ts, seeds, mcfg=None, dcfg=None, lb=None, ub=None, want_desc=False):
    """The reference driver loop (batch_bin_test.cpp:105-247) on the CPU restatement, timed per stage."""
    mcfg = mcfg or L.default_manager_cfg()
    dcfg = dcfg or L.default_db_cfg()
    if lb is None:
        lb, ub = L.default_thresholds()
    xyzi = np.ascontiguousarray(xyzi, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    ts = np.ascontiguousarray(ts, np.float64)
    seeds = np.ascontiguousarray(seeds, np.int32)
    n = len(ts)
    res = np.zeros(n, L.query_result_dt)
    timers = np.zeros(5, np.float64)
    desc = np.zeros(n, L.scan_desc_dt) if want_desc else None
    rc = lib().orc_run_sequence(_p(xyzi), _p(offsets), n, _p(ts), _p(seeds), C.addressof(mcfg), C.addressof(dcfg),
                                C.addressof(lb), C.addressof(ub), _p(res), _p(timers), _p(desc))
    if rc != 0:
        raise RuntimeError("orc_run_sequence failed")
    names = ["make bev", "KNN search", "Constell", "L2 opt", "Update database"]
    return res, dict(zip(names, timers.tolist())), desc


def ingest_batch(xyzi, offsets, mcfg=None, want_desc=True):
    mcfg = mcfg or L.default_manager_cfg()
    xyzi = np.ascontiguousarray(xyzi, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n = len(offsets) - 1
    desc = np.zeros(n, L.scan_desc_dt) if want_desc else None
    rc = lib().orc_ingest_batch(_p(xyzi), _p(offsets), n, C.addressof(mcfg), _p(desc))
    if rc != 0:
        raise RuntimeError("orc_ingest_batch failed")
    return desc


def check_pair(cand, tgt, level, seq_src, seq_tgt, sim=None, lb=None, ub=None):
    sim = sim or L.default_db_cfg().cont_sim
    if lb is None:
        lb, ub = L.default_thresholds()
    oi = np.zeros(8, np.int32)
    tf = np.zeros(3, np.float64)
    pairs = np.zeros((64, 3), np.int8)
    lib().orc_check_pair(cand.h, tgt.h, level, seq_src, seq_tgt, C.addressof(sim), C.addressof(lb), C.addressof(ub),
                         _p(oi), _p(tf), _p(pairs))
    return oi, tf, pairs[:oi[6]]


def check_hints(tgt, cands, hints, sim=None, lb=None, ub=None, max_fine_opt=10):
    """single-pair flow (kitti_read_bin_test.cpp:226-291): hints = int array [n][4] of (index into cands, level, seq_src, seq_tgt)
    -> (cc_query_result_t record with cand_gidx = index into cands, scores [n][6])"""
    sim = sim or L.default_db_cfg().cont_sim
    if lb is None:
        lb, ub = L.default_thresholds()
    hints = np.ascontiguousarray(hints, np.int32).reshape(-1, 4)
    hs = (C.c_void_p * len(cands))(*[c.h for c in cands])
    res = np.zeros(1, L.query_result_dt)
    sc = np.zeros((len(hints), 6), np.int32)
    lib().orc_check_hints(tgt.h, hs, len(cands), _p(hints), len(hints), C.addressof(sim), C.addressof(lb), C.addressof(ub),
                          int(max_fine_opt), _p(res), _p(sc))
    return res[0], sc


def gmm(src, tgt, tf_init):
    tf_init = np.ascontiguousarray(tf_init, np.float64)
    ci, co = C.c_double(), C.c_double()
    tf = np.zeros(3, np.float64)
    it = np.zeros(3, np.int32)
    lib().orc_gmm(src.h, tgt.h, _p(tf_init), C.byref(ci), C.byref(co), _p(tf), _p(it))
    return ci.value, co.value, tf, it


def gmm_trace(src, tgt, tf_init):
    """accepted points x_0 .. x_n of the L-BFGS refinement (n <= 10) and the termination code"""
    tf_init = np.ascontiguousarray(tf_init, np.float64)
    xs = np.zeros((11, 3), np.float64)
    term = C.c_int32()
    n = lib().orc_gmm_trace(src.h, tgt.h, _p(tf_init), _p(xs), C.byref(term))
    return xs[:n].copy(), term.value


def gmm_eval(src, tgt, tf_init, p):
    tf_init = np.ascontiguousarray(tf_init, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    cost = C.c_double()
    grad = np.zeros(3)
    ac = np.zeros(2)
    lib().orc_gmm_eval(src.h, tgt.h, _p(tf_init), _p(p), C.byref(cost), _p(grad), _p(ac))
    return cost.value, grad, ac


def umeyama(src, tgt, pairs):
    pairs = np.ascontiguousarray(pairs, np.int8).reshape(-1, 3)
    tf = np.zeros(3)
    lib().orc_umeyama(src.h, tgt.h, _p(pairs), len(pairs), _p(tf))
    return tf


def knn_scan(keys, q, k, max_dist_sq):
    keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 10)
    q = np.ascontiguousarray(q, np.float32)
    idx = np.zeros(k, np.int32)
    d = np.zeros(k, np.float32)
    n = lib().orc_knn_scan(_p(keys), len(keys), _p(q), k, max_dist_sq, _p(idx), _p(d))
    return idx[:n], d[:n]


def eigen2f(m):
    m = np.ascontiguousarray(m, np.float32).reshape(4)
    ev = np.zeros(2, np.float32)
    vec = np.zeros(4, np.float32)
    lib().orc_eigen2f(_p(m), _p(ev), _p(vec))
    return ev, vec.reshape(2, 2)


def ccl8(img):
    """The oracle's connectedComponentsWithStats (8-connectivity) on a uint8 patch -> (label image int32, stats [n][5])."""
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    lab = np.zeros((rows, cols), np.int32)
    st = np.zeros((rows * cols + 1, 5), np.int32)
    n = lib().orc_ccl8(_p(img), rows, cols, _p(lab), _p(st))
    return lab, st[:n]


def sort_desc_perm(keys):
    keys = np.ascontiguousarray(keys, np.int32)
    perm = np.zeros(len(keys), np.int32)
    lib().orc_sort_desc_perm(_p(keys), len(keys), _p(perm))
    return perm


def sort_asc_perm_f(keys):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(len(keys), np.int32)
    lib().orc_sort_asc_perm_f(_p(keys), len(keys), _p(perm))
    return perm


_ref = None


def use_ref_kdtree(on=True):
    """Route the oracle's bucket searches through the reference's real nanoflann (oracle/_ref) -- used for
    the timed CPU baseline so that "KNN search" is a kd-tree like the reference's, not an exact scan.
    Returns True if the backend is active."""
    global _ref
    so = os.path.join(_HERE, "_ref", "libref_knn.so")
    if not on:
        lib().orc_set_knn_backend(None, None, None, None)
        return False
    if not os.path.exists(so):
        return False
    if _ref is None:
        _ref = C.CDLL(so)
        _ref.ref_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib().orc_set_knn_backend.argtypes = [C.c_void_p] * 4
    f = lambda n: C.cast(getattr(_ref, n), C.c_void_p)
    lib().orc_set_knn_backend(f("refkd_create"), f("refkd_free"), f("refkd_build"), f("refkd_query"))
    return bool(lib().orc_knn_backend_active())


def ref_knn(keys, q, k, max_dist_sq):
    """Real nanoflann (reference's vendored header) kNN-with-max-dist; None if oracle/_ref was not built."""
    global _ref
    so = os.path.join(_HERE, "_ref", "libref_knn.so")
    if _ref is None:
        if not os.path.exists(so):
            return None
        _ref = C.CDLL(so)
        _ref.ref_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 10)
    q = np.ascontiguousarray(q, np.float32)
    idx = np.zeros(k, np.int32)
    d = np.zeros(k, np.float32)
    n = _ref.ref_knn(_p(keys), len(keys), _p(q), k, max_dist_sq, _p(idx), _p(d))
    return idx[:n], d[:n]
