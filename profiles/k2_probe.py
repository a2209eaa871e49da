#!/usr/bin/env python3
"""Tuning aid (round 5): isolated K1 / K2 times of one 1 024-scan ingest launch per workload, K2's phase clocks
(CC_K2_PHASES=1) and a digest of the descriptors -- the digest of a changed kernel must equal the digest of the kernel it
replaces (bit-exact descriptors at full size, without the oracle in the loop).
    python profiles/k2_probe.py [sparse,dense,kitti] [n_scans] [repeats]   -> one JSON line per workload"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("CC_PROBE_NOPHASES"):   # the phase clocks cost time themselves (a clock read is a memory operation): true kernel times come without them
    os.environ.setdefault("CC_K2_PHASES", "1")
import torch  # noqa: E402
import cc_amd  # noqa: E402

cc = cc_amd.load()
if os.environ.get("CC_PROBE_LIB"):   # A/B against another build of the library (e.g. profiles/r5/libcont2_r4.so = round 4's kernels)
    cc.LIB_PATH = os.path.abspath(os.environ["CC_PROBE_LIB"])
C = cc.C if hasattr(cc, "C") else __import__("ctypes")


def main():
    worlds = (sys.argv[1] if len(sys.argv) > 1 else "sparse,kitti").split(",")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rep = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda", 0)
    ctx = cc.Context(0, max_batch=n)
    lib = cc.lib()
    for wname in worlds:
        wld = cc.synth.World(kitti=True) if wname == "kitti" else cc.synth.World(dense=(wname == "dense"))
        xyzi, _, _ = cc.synth.make_sequence(n, world=wld, device=dev, start=5000, beams=64, azim=1875)
        P = xyzi.shape[1]
        x = xyzi.reshape(-1, 4).contiguous()
        offs = np.arange(n + 1, dtype=np.int64) * P
        out = torch.zeros((n, cc.DESC_BYTES), dtype=torch.uint8, device=dev)  # unwritten table slots must not carry an earlier workload
        ctx.ingest(x, offs, out=out)
        torch.cuda.synchronize()
        lib.cc_profile_enable(ctx.h, 1)
        for _ in range(rep):
            ctx.ingest(x, offs, out=out)
        torch.cuda.synchronize()
        ms = (C.c_double * 2)()
        nl = C.c_int()
        sys.stderr.write("[%s] " % wname)
        sys.stderr.flush()
        lib.cc_profile_read(ctx.h, ms, C.byref(nl))
        lib.cc_profile_enable(ctx.h, 0)
        d = cc.desc_to_numpy(out)
        _, dbg = ctx.ingest(x[:16 * P], offs[:17], debug=True)
        bevs = dbg["bev"].cpu().numpy()
        n_act = (bevs > 1.5).sum(axis=1)
        n_occ = (bevs > -999.0).sum(axis=1)
        flags = d["flags"]
        dig = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        dk = hashlib.sha256(np.ascontiguousarray(d["keys"]).tobytes()).hexdigest()[:16]
        d2 = d.copy()
        d2["keys"] = 0
        dnk = hashlib.sha256(d2.tobytes()).hexdigest()[:16]
        if os.environ.get("CC_PROBE_SAVE"):   # the whole descriptor array, for profiles/r5/cmp_desc.py
            np.save(os.environ["CC_PROBE_SAVE"] + "_" + wname + ".npy", out.cpu().numpy())
        print(json.dumps({"workload": wname, "scans": n, "k1_ms": ms[0] / nl.value, "k2_ms": ms[1] / nl.value, "launches": nl.value,
                          "flagged": int((flags != 0).sum()), "n_cont_mean": [float(v) for v in d["n_cont"].mean(axis=0)],
                          "n_act_mean": float(n_act.mean()), "n_act_max": int(n_act.max()), "n_occ_mean": float(n_occ.mean()),
                          "largest_comp_mean": float(d["cont"]["cell_cnt"][:, 0, 0].mean()), "largest_comp_max": int(d["cont"]["cell_cnt"][:, 0, 0].max()),
                          "layer_cells_mean": [float(v) for v in d["layer_cell_cnt"].mean(axis=0)], "digest": dig, "digest_keys": dk, "digest_without_keys": dnk}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
