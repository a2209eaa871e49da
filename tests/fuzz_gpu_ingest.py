"""Randomised ingest parity campaign ON THE GPU (not collected by pytest; run by hand on the GPU box):
    python tests/fuzz_gpu_ingest.py <seed0> <n_iter> [fit]
("fit": two scans of three are lowered / thinned at random so that they fit K2's list kernel, csrc/k_contours_list.h; the campaign
counts how many scans that kernel kept -- by its capacities, from the oracle's images -- and how many went on to the bodies behind it)
The scan generators of tests/fuzz_emu.py (terrain, uniform clouds, blobs, cell borders, walls, heights exactly at the level
thresholds, duplicates and far outliers), 24 scans of different sizes per cc_ingest_batch call, a ContourManagerConfig drawn per
batch (shipped / MulRan levels, grids, resolutions, contour / key / RoI settings), every descriptor against the oracle: integers, contour rows and BCIs bit for bit, keys to the last bits of the f64 exp."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import cc_amd  # noqa: E402
import oracle_py as oracle  # noqa: E402
from fuzz_emu import gen  # noqa: E402
from parity import compare_desc  # noqa: E402


def main():
    import torch
    seed0, n_it = int(sys.argv[1]), int(sys.argv[2])
    cc = cc_amd.load()
    L = oracle.L
    n_bad = n_flag = n_scan = n_big = n_list = 0
    fit = len(sys.argv) > 3 and sys.argv[3] == "fit"
    for it in range(n_it):
        rng = np.random.default_rng(seed0 + it)
        # a ContourManagerConfig per batch: the shipped one, the MulRan level set, other grids and resolutions (the
        # multiply-by-reciprocal and the IEEE-division instances of the rasteriser), other contour / key / RoI settings
        mcfg = L.default_manager_cfg(mulran=bool(rng.random() < 0.25))
        v = int(rng.integers(6))
        if v == 1:
            mcfg.reso_row = mcfg.reso_col = 2.0
            mcfg.n_row = mcfg.n_col = 74
        elif v == 2:
            mcfg.reso_row, mcfg.reso_col = 1.5, 0.75
            mcfg.n_row, mcfg.n_col = 100, 120
        elif v == 3:
            mcfg.n_row, mcfg.n_col = 120, 150
            mcfg.min_cont_cell_cnt, mcfg.min_cont_key_cnt = 5, 12
        elif v == 4:
            mcfg.piv_firsts, mcfg.dist_firsts, mcfg.roi_radius = 4, 8, 8.0
            mcfg.blind_sq = 4.0
        elif v == 5:
            mcfg.lidar_height = 1.73
            mcfg.min_cont_cell_cnt = 4
        ctx = cc.Context(0, mcfg, max_batch=32)
        scans = []
        while len(scans) < 24:
            kind, s = gen(rng)
            if fit and len(scans) % 3 != 2:
                s = s[rng.random(len(s)) < rng.uniform(0.15, 1.0)].copy()
                s[:, 2] -= np.float32(rng.uniform(0.5, 2.5))
            if len(s) > 10:
                scans.append((kind, s))
        offs = np.concatenate([[0], np.cumsum([len(s) for _, s in scans])]).astype(np.int64)
        x = torch.from_numpy(np.concatenate([s for _, s in scans], 0)).cuda()
        desc = cc.desc_to_numpy(ctx.ingest(x, offs))   # a scan that exceeds a capacity comes back flagged, the call succeeds
        for k, (kind, s) in enumerate(scans):
            osc = oracle.Scan(s, cfg=mcfg)
            od = osc.desc()[0]
            n_scan += 1
            lvl = (osc.bev()[0][None, :] > np.asarray(list(mcfg.lv_grads), np.float32)[:, None]).sum(0)   # level count per cell
            n_list += int(mcfg.min_cont_cell_cnt <= 3 and (lvl > 0).sum() <= 3072 and lvl.sum() <= 12800 and int(od["n_cont"].max()) <= L.MAXC)
            if desc[k]["flags"] & 6:   # CC_DESC_INEXACT_*: since round 5 only an over-full key RoI (roi_radius_ > 10) can do that
                n_flag += 1
                if desc[k]["flags"] & 2:
                    print("seed %d scan %d (%s): CC_DESC_INEXACT_COMPONENTS (n_cont max %d): the slow path must have made it exact"
                          % (seed0 + it, k, kind, int(od["n_cont"].max())))
                    n_bad += 1
                continue
            n_big += int(od["n_cont"].max()) > L.MAXC   # went through cc_k_contours_big; compared like every other scan
            bad = compare_desc(od, desc[k], float_exact=False)
            if bad:
                print("seed %d scan %d (%s, %d points): %s" % (seed0 + it, k, kind, len(s), bad[:3]))
                n_bad += 1
        ctx.close()
        if it % 10 == 9:
            print("... %d batches, %d scans, %d within the list kernel's capacities, %d through the slow path, %d flagged, %d bad" % (it + 1, n_scan, n_list, n_big, n_flag, n_bad), flush=True)
    print("done: %d bad of %d scans (%d within the list kernel's capacities, %d with more than CC_MAXC components on a level, compared; %d flagged inexact)" % (n_bad, n_scan, n_list, n_big, n_flag))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
