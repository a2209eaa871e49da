#!/usr/bin/env python3
"""Offline evaluation of a whole sequence at batch throughput: the job of the reference's test/batch_bin_test.cpp
(config file -> scans -> loop-closure predictions -> outcome file), but with the scans ingested and queried in batches
instead of one by one.  The online loop's semantics are kept through the DB epochs: every scan is added, then scan i
is queried against the DB as it was after i scans (what `query, then insert` sees).

    python contour-context_amd/tools/batch_eval.py config.yaml [--chunk 256] [--lib /path/to/libcont2_amd.so]

config.yaml carries the reference's keys (config/batch_bin_test_config.yaml; see hostcpp/examples/batch_bin_test_config.yaml).
Only host buffers cross the C-ABI here (cc_ingest_host, cc_db_add_scans_host, cc_db_query_batch_host): no torch needed.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import evaluator as E  # noqa: E402
import layouts as L    # noqa: E402
import pr_eval         # noqa: E402


def load_config(path):
    """the reference's OpenCV-FileStorage flavoured YAML: `%YAML:1.0` + `---` preamble, otherwise plain YAML"""
    import yaml
    text = "".join(l for l in open(path) if not l.startswith("%") and l.strip() != "---")
    return yaml.safe_load(text)


def structs_from_config(cfg):
    m, d = L.default_manager_cfg(), L.default_db_cfg()
    lb, ub = L.default_thresholds()
    cm = cfg.get("ContourManagerConfig", {})
    if "lv_grads_" in cm:
        assert len(cm["lv_grads_"]) == L.NLEV, "this build handles 6-level configs"
        for i, v in enumerate(cm["lv_grads_"]):
            m.lv_grads[i] = float(v)
    for k in ("reso_row", "reso_col", "n_row", "n_col", "lidar_height", "blind_sq", "min_cont_key_cnt", "min_cont_cell_cnt", "piv_firsts",
              "dist_firsts", "roi_radius"):
        if k + "_" in cm:
            setattr(m, k, type(getattr(m, k))(cm[k + "_"]))
    cd = cfg.get("ContourDBConfig", {})
    if "nnk_" in cd:
        d.nnk = int(cd["nnk_"])
    if "max_fine_opt_" in cd:
        d.max_fine_opt = int(cd["max_fine_opt_"])
    if "q_levels_" in cd:
        d.n_q_levels = len(cd["q_levels_"])
        for i, v in enumerate(cd["q_levels_"]):
            d.q_levels[i] = int(v)
    for k, v in cd.get("ContourSimThresConfig", {}).items():
        setattr(d.cont_sim, k, float(v))
    tb = cd.get("TreeBucketConfig", {})
    if "max_elapse_" in tb:
        d.max_elapse = float(tb["max_elapse_"])
    if "min_elapse_" in tb:
        d.min_elapse = float(tb["min_elapse_"])
    for name, s in (("thres_lb_", lb), ("thres_ub_", ub)):
        for k, v in cfg.get(name, {}).items():
            setattr(s, k, type(getattr(s, k))(v))
    return m, d, lb, ub


def read_bin(path):
    """KITTI velodyne .bin: x, y, z, intensity f32, at most 1e6 floats are read (tools/pointcloud_util.h:9-47)"""
    a = np.fromfile(path, dtype=np.float32, count=1000000)
    return a[:len(a) // 4 * 4].reshape(-1, 4)


def run(config_path, lib_path=None, chunk=256, verbose=True):
    cfg = load_config(config_path)
    m, d, lb, ub = structs_from_config(cfg)
    ev = E.ContLCDEvaluator(cfg["fpath_sens_gt_pose"], cfg["fpath_lidar_bins"], float(cfg["correlation_thres"]))
    n = len(ev.scans)
    if lib_path is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("contour_context_amd", os.path.join(os.path.dirname(HERE), "__init__.py"),
                                                      submodule_search_locations=[os.path.dirname(HERE)])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules["contour_context_amd"] = pkg
        spec.loader.exec_module(pkg)
        lib_path = pkg.LIB_PATH
    lib = C.CDLL(lib_path)
    lib.cc_last_error.restype = C.c_char_p

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.cc_last_error().decode()))

    ctx, db = C.c_void_p(), C.c_void_p()
    chk(lib.cc_create(0, C.byref(m), int(chunk), C.byref(ctx)), "cc_create")
    chk(lib.cc_db_create(ctx, C.byref(d), n + 8, C.byref(db)), "cc_db_create")
    desc = np.zeros(n, L.scan_desc_dt)
    for c0 in range(0, n, chunk):
        c1 = min(c0 + chunk, n)
        pts = [read_bin(ev.scans[i]["fpath"]) for i in range(c0, c1)]
        offs = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int64)
        xyzi = np.ascontiguousarray(np.concatenate(pts), np.float32)
        chk(lib.cc_ingest_host(ctx, C.c_void_p(xyzi.ctypes.data), C.c_void_p(offs.ctypes.data), c1 - c0,
                               C.c_void_p(desc[c0:c1].ctypes.data)), "cc_ingest_host")
        if verbose:
            print("ingested %d / %d scans" % (c1, n), flush=True)
    ts = np.array([s["ts"] for s in ev.scans], np.float64)
    seqs = np.array([s["seq"] for s in ev.scans], np.int32)
    chk(lib.cc_db_add_scans_host(db, C.c_void_p(desc.ctypes.data), n, C.c_void_p(ts.ctypes.data), C.c_void_p(seqs.ctypes.data)),
        "cc_db_add_scans_host")
    res = np.zeros(n, L.query_result_dt)
    epochs = np.arange(n, dtype=np.int32)   # scan i sees the DB after i scans: query first, then insert
    for c0 in range(0, n, 2048):
        c1 = min(c0 + 2048, n)
        chk(lib.cc_db_query_batch_host(db, C.c_void_p(desc[c0:c1].ctypes.data), c1 - c0, C.c_void_p(epochs[c0:c1].ctypes.data),
                                       C.byref(lb), C.byref(ub), C.c_void_p(res[c0:c1].ctypes.data)), "cc_db_query_batch_host")
    for i in range(n):
        r = res[i]
        if r["n_res"] > 0:
            ev.add_prediction(int(seqs[i]), float(r["correlation"]), int(seqs[r["cand_gidx"]]), tuple(float(v) for v in r["tf"]),
                              m.n_row, m.n_col, m.reso_row)
        else:
            ev.add_prediction(int(seqs[i]), 0.0)
    ev.save_prediction_results(cfg["fpath_outcome_sav"])
    lib.cc_db_destroy(db)
    lib.cc_destroy(ctx)
    summary = pr_eval.evaluate(pr_eval.load_gt_poses(cfg["fpath_sens_gt_pose"]), pr_eval.load_outcome(cfg["fpath_outcome_sav"]))
    if verbose:
        tp = ev.tp_errors()
        print("outcome -> %s" % cfg["fpath_outcome_sav"])
        print("max F1 %.4f at similarity %.4f; TP at the configured threshold: mean t %.4f m r %.4f rad, rmse t %.4f m r %.4f rad"
              % (summary["max_f1"], summary["sim_thres"], *tp))
    return ev, res, summary


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--lib", default=None, help="shared library exporting the cc_* C-ABI (default: the package's libcont2_amd.so)")
    ap.add_argument("--chunk", type=int, default=256, help="scans ingested per call")
    a = ap.parse_args()
    run(a.config, a.lib, a.chunk)
