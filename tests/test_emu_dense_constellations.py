"""Stress of the constellation check's large cases on the CPU harness: descriptors of a real sequence whose BCIs are
rewritten so that every anchor has up to 18 neighbours crowded into three adjacent distance bins per layer.  Every
(src, tgt) anchor pair then yields 100+ potential neighbour pairs with many exactly equal orientation differences:
stage B's large instance (redo of > 64 pairs), the parallel replay of std::sort's partitions on > 16 elements with ties,
the binned rank and the 64-pair constellation cap all run, and the results must still equal the oracle's."""
import numpy as np

import emu_api

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy",
              "n_knn_hits"]


def _crowd(desc, per_level=9, levels=(1, 2)):
    d = desc.copy()
    n_big = 0
    for i in range(len(d)):
        for l in range(d["bcis"].shape[1]):
            for s in range(d["bcis"].shape[2]):
                b = d["bcis"][i, l, s]
                n = int(b["n_pts"])
                if n == 0:
                    continue
                old = b["pts"][:n].copy()
                keep = []
                for lev in levels:
                    idx = [k for k in range(n) if old["level"][k] == lev][:per_level]
                    for j, k in enumerate(idx):
                        p = old[k].copy()
                        p["bit_pos"] = 64 * (lev - 1) + 30 + (j % 3)
                        keep.append(p)
                keep.sort(key=lambda p: int(p["bit_pos"]))      # BCI invariant: points ordered by bit_pos
                nb = np.zeros((), b.dtype)
                nb["piv_seq"], nb["level"] = b["piv_seq"], b["level"]
                nb["n_pts"] = len(keep)
                bits = [0, 0, 0, 0]
                segs = []
                for k, p in enumerate(keep):
                    nb["pts"][k] = p
                    bp = int(p["bit_pos"])
                    bits[bp >> 6] |= 1 << (bp & 63)
                    if k == 0 or int(keep[k - 1]["bit_pos"]) != bp:
                        segs.append(k)
                if keep:
                    segs.append(len(keep))
                nb["dist_bin"] = np.array(bits, np.uint64)
                nb["n_segs"] = len(segs)
                nb["segs"][:len(segs)] = segs
                d["bcis"][i, l, s] = nb
                n_big += len(keep) >= 12
    return d, n_big


def test_crowded_bcis_match_oracle(cc, oracle):
    L = oracle.L
    dcfg = L.default_db_cfg()
    dcfg.max_elapse, dcfg.min_elapse = 2.5, 1.5
    w = cc.synth.World(loop_len=40.0)
    n = 56
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    P = x.shape[1]
    _, _, odesc = oracle.run_sequence(x.numpy().reshape(-1, 4), np.arange(n + 1, dtype=np.int64) * P, ts, np.arange(n, dtype=np.int32),
                                      dcfg=dcfg, want_desc=True)
    desc, n_big = _crowd(odesc)
    assert n_big > 100, "the rewrite should leave many anchors with >= 12 neighbours"
    # oracle: the reference loop (query, then insert) on the rewritten descriptors
    odb = oracle.DB(dcfg)
    exp = []
    for i in range(n):
        s = oracle.Scan.from_desc(desc[i], int_id=i)
        exp.append(odb.query(s))
        odb.add_scan(s, ts[i])
        odb.push_and_balance(i, ts[i])
    exp = np.array(exp)
    assert exp["cand_aft_check2"].max() > 20, "crowded constellations should survive the angular window often"
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, dcfg, cap=n)
    api.db_add(db, desc, ts, np.arange(n, dtype=np.int32))
    qs = np.argsort(-exp["cand_aft_check2"], kind="stable")[:4].astype(np.int32)
    qs = np.concatenate([qs, [n - 1]]).astype(np.int32)
    res = api.db_query(db, desc[qs], qs)
    for k, qi in enumerate(qs):
        for f in INT_FIELDS:
            assert exp[f][qi] == res[f][k], (qi, f, exp[f][qi], res[f][k])
        if exp["n_res"][qi]:
            assert abs(exp["correlation"][qi] - res["correlation"][k]) < 1e-6
            assert np.abs(exp["tf"][qi] - res["tf"][k]).max() < 1e-6
