"""Kernel + host logic of the query path on the CPU (product TU compiled against tests/emu) vs the oracle's replay of
the reference loop, on a short looping sequence (shortened DB delays so revisits are searchable early)."""
import numpy as np

import emu_api

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy",
              "n_knn_hits"]


def test_short_loop_sequence(cc, oracle):
    L = oracle.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    P = x.shape[1]
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * P
    seeds = np.arange(n, dtype=np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=dcfg, want_desc=True)
    assert (ores["n_res"] > 0).sum() >= 3
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, odesc, ts, seeds)
    hit = np.nonzero(ores["n_res"] > 0)[0]
    qs = np.concatenate([hit[:3], [5, n - 1]]).astype(np.int32)
    res = api.db_query(db, odesc[qs], qs)
    for k, qi in enumerate(qs):
        for f in INT_FIELDS:
            assert ores[f][qi] == res[f][k], (qi, f, ores[f][qi], res[f][k])
        if ores["n_res"][qi]:
            assert abs(ores["correlation"][qi] - res["correlation"][k]) < 1e-6
            assert np.abs(ores["tf"][qi] - res["tf"][k]).max() < 1e-6
