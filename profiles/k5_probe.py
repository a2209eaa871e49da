#!/usr/bin/env python3
"""Tuning aid (round 6): the query chain of ONE 1 024-query chunk against a 5 000-scan DB, alone on the GPU: isolated times of
the kernel groups, a digest of the results (what a changed K3 / K4 / K5 must reproduce: matched scan, gate counters,
candidate counts exactly; correlation / pose rounded to 1e-9) and -- with a library built with -DCC_TUNE_GMM_CLK -- the
per-problem clocks of cc_k_gmm_refine (pairs, evaluations, cycles, cycles inside evaluations).
    python profiles/k5_probe.py [kitti|sparse|dense] [repeats]        CC_PROBE_LIB=<other build> for an A/B"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cc_amd  # noqa: E402

cc = cc_amd.load()
if os.environ.get("CC_PROBE_LIB"):
    cc.LIB_PATH = os.path.abspath(os.environ["CC_PROBE_LIB"])


def main():
    wname = sys.argv[1] if len(sys.argv) > 1 else "kitti"
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_db = int(os.environ.get("CC_PROBE_DB", "5000"))
    B = 1024
    dev = torch.device("cuda", 0)
    ctx = cc.Context(0, max_batch=B)
    lib = cc.lib()
    wld = cc.synth.World(kitti=True) if wname == "kitti" else cc.synth.World(dense=(wname == "dense"))
    HB, FB = cc.packed_sizes()
    rec = torch.empty((n_db, HB + FB), dtype=torch.uint8, device=dev)
    tmp = torch.empty((256, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    P = None
    for c0 in range(0, n_db, 256):
        c1 = min(c0 + 256, n_db)
        x, _, _ = cc.synth.make_sequence(c1 - c0, world=wld, device=dev, start=c0)
        P = x.shape[1]
        d = ctx.ingest(x.reshape(-1, 4), np.arange(c1 - c0 + 1, dtype=np.int64) * P, out=tmp[:c1 - c0])
        hot, feat = ctx.pack(d)
        rec[c0:c1, :HB] = hot
        rec[c0:c1, HB:] = feat
    db = cc.Database(ctx, capacity=n_db + 16)
    db.add_packed(rec[:, :HB].contiguous(), rec[:, HB:].contiguous(), np.arange(n_db) / 10.0, np.arange(n_db, dtype=np.int32))
    db.set_lanes(1)
    x, _, _ = cc.synth.make_sequence(B, world=wld, device=dev, start=n_db + B * int(os.environ.get("CC_PROBE_BATCH", "4")))   # bench.py: batch s of the drive
    q = torch.empty((B, cc.DESC_BYTES), dtype=torch.uint8, device=dev)
    ctx.ingest(x.reshape(-1, 4).contiguous(), np.arange(B + 1, dtype=np.int64) * P, out=q)
    epochs = np.full(B, n_db, np.int32)
    res = db.query(q, epochs)
    torch.cuda.synchronize()
    has_clk = hasattr(lib, "cc_tune_gmm_clk_read")
    if has_clk:
        lib.cc_tune_gmm_clk_read.argtypes = [C.c_void_p, C.c_int]
        buf = np.zeros((65536, 4), np.uint64)
        lib.cc_tune_gmm_clk_read(buf.ctypes.data, 65536)   # drop the warm-up's
        if hasattr(lib, "cc_tune_gmm_scan_clk_read"):
            lib.cc_tune_gmm_scan_clk_read.argtypes = [C.c_void_p]
            lib.cc_tune_gmm_scan_clk_read(np.zeros(8, np.uint64).ctypes.data)
    lib.cc_db_profile_enable(db.h, 1)
    ms5 = (C.c_double * 5)()
    nl = C.c_int()
    lib.cc_db_profile_read(db.h, ms5, C.byref(nl))
    for _ in range(rep):
        res = db.query(q, epochs)
    torch.cuda.synchronize()
    lib.cc_db_profile_read(db.h, ms5, C.byref(nl))
    b = max(nl.value, 1) / float(B)
    kms = {k: round(ms5[i] / b, 4) for i, k in enumerate(("knn", "check", "merge", "gmm", "final"))}
    h = hashlib.sha256()
    names = res.dtype.names
    exact, rounded = [], []
    for nme in names:
        a = np.ascontiguousarray(res[nme])
        if a.dtype.kind == "f":
            rounded.append(nme)
            h.update(np.round(a.astype(np.float64), 7).tobytes())
        else:
            exact.append(nme)
            h.update(a.tobytes())
    out = {"workload": wname, "db": n_db, "kernels_ms": kms, "digest": h.hexdigest()[:16], "found": int((res["n_res"] > 0).sum()),
           "problems": int(res["n_cand_tidy"].sum()) if "n_cand_tidy" in names else None, "fields_rounded": rounded}
    if has_clk:
        buf2 = np.zeros((65536, 4), np.uint64)
        if hasattr(lib, "cc_tune_gmm_clk2_read"):
            lib.cc_tune_gmm_clk2_read.argtypes = [C.c_void_p, C.c_int]
            lib.cc_tune_gmm_clk2_read(buf2.ctypes.data, 65536)
        n = lib.cc_tune_gmm_clk_read(buf.ctypes.data, 65536)
        if os.environ.get("CC_PROBE_DUMP"):
            np.save(os.environ["CC_PROBE_DUMP"], np.concatenate([buf[:n], buf2[:n]], axis=1))
        a = buf[:n].astype(np.int64)
        n_per = max(n // rep, 1)
        np_ = a[:, 0] & 0xFFFFF
        G = (a[:, 0] >> 20) & 0xFFFFF
        it = (a[:, 0] >> 40) & 0xFF
        nev = (a[:, 0] >> 48) & 0xFFFF
        tot, ev, fil = a[:, 1], a[:, 2], a[:, 3]
        clk = {"problems_per_chunk": n_per}
        for g in sorted(set(G.tolist())):
            m = G == g
            o = np.argsort(-tot[m])
            t, e, f, p_, v, i_ = tot[m][o], ev[m][o], fil[m][o], np_[m][o], nev[m][o], it[m][o]
            clk["G%d" % g] = {
                "n": int(m.sum()) // rep, "cycles_mean": float(t.mean()), "cycles_max": int(t[0]), "cycles_p99": float(np.percentile(t, 99)),
                "eval_share_mean": float((e / np.maximum(t, 1)).mean()), "file_share_mean": float((f / np.maximum(t, 1)).mean()),
                "evals_mean": float(v.mean()), "iters_mean": float(i_.mean()), "pairs_mean": float(p_.mean()), "pairs_max": int(p_.max()),
                "serial_cycles_per_eval_mean": float(((t - e - f) / np.maximum(v, 1)).mean()),
                "eval_cycles_per_pair_step": float((e / np.maximum(v * np.ceil(p_ / float(g)), 1)).mean()),
                "longest": [{"pairs": int(p_[k]), "evals": int(v[k]), "iters": int(i_[k]), "cycles": int(t[k]), "eval": int(e[k]), "file": int(f[k])} for k in range(min(8, len(t)))],
                "by_pairs": [{"pairs_lt": int(hi), "n": int(((p_ >= lo) & (p_ < hi)).sum()) // rep,
                              "cycles_mean": float(t[(p_ >= lo) & (p_ < hi)].mean()) if ((p_ >= lo) & (p_ < hi)).any() else 0.0,
                              "eval_share": float((e / np.maximum(t, 1))[(p_ >= lo) & (p_ < hi)].mean()) if ((p_ >= lo) & (p_ < hi)).any() else 0.0}
                             for lo, hi in ((0, 97), (97, 257), (257, 513), (513, 1025), (1025, 2049), (2049, 4097), (4097, 1 << 20))],
            }
        if hasattr(lib, "cc_tune_gmm_scan_clk_read"):
            s8 = np.zeros(8, np.uint64)
            lib.cc_tune_gmm_scan_clk_read.argtypes = [C.c_void_p]
            lib.cc_tune_gmm_scan_clk_read(s8.ctypes.data)
            k = max(int(s8[5]), 1)
            clk["scan_cycles_per_64lane_problem (init + refine)"] = dict(zip(("fill", "f32_sweep", "f64_tests", "filing", "flush"), [round(float(v) / k, 1) for v in s8[:5]]))
            clk["scan_count"] = int(s8[5])
        out["gmm_clk"] = clk
    print(json.dumps(out))
    db.close()


if __name__ == "__main__":
    main()
