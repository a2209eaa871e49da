"""Kernel + host logic of the query path on the CPU (product TU compiled against tests/emu) vs the oracle's replay of
the reference loop, on a short looping sequence (shortened DB delays so revisits are searchable early)."""
import numpy as np
import pytest

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


def _variant(cc, oracle, nnk, qlv, mfo, thr, sim):
    L = oracle.L
    d = L.default_db_cfg()
    d.max_elapse, d.min_elapse = 2.5, 1.5
    d.nnk, d.max_fine_opt, d.n_q_levels = nnk, mfo, len(qlv)
    for i, v in enumerate(qlv):
        d.q_levels[i] = v
    for k, v in sim.items():
        setattr(d.cont_sim, k, v)
    lb, ub = L.default_thresholds()
    for k, v in thr.items():
        setattr(lb, k, v)
    w = cc.synth.World(loop_len=40.0)
    n = 64
    x, poses, ts = cc.synth.make_sequence(n, world=w, beams=16, azim=450)
    xs = x.numpy().reshape(-1, 4)
    offs = np.arange(n + 1, dtype=np.int64) * x.shape[1]
    seeds = np.arange(n, dtype=np.int32)
    ores, _, odesc = oracle.run_sequence(xs, offs, ts, seeds, dcfg=d, lb=lb, ub=ub, want_desc=True)
    hit = np.nonzero(ores["n_res"] > 0)[0]
    assert len(hit) >= 3
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, d, cap=n)
    api.db_add(db, odesc, ts, seeds)
    qs = np.unique(np.concatenate([hit[:3], [20, n - 1]])).astype(np.int32)
    res = api.db_query(db, odesc[qs], qs, lb=lb, ub=ub)
    for k, qi in enumerate(qs):
        for f in INT_FIELDS:
            assert ores[f][qi] == res[f][k], (qi, f, ores[f][qi], res[f][k])
        if ores["n_res"][qi]:
            assert abs(ores["correlation"][qi] - res["correlation"][k]) < 1e-6
            assert np.abs(ores["tf"][qi] - res["tf"][k]).max() < 1e-6


def test_non_default_db_config_small_k_two_levels(cc, oracle):
    """nnk_ = 10, q_levels_ = [2, 3], max_fine_opt_ = 2, stricter gates."""
    _variant(cc, oracle, 10, (2, 3), 2, dict(i_ovlp_sum=5, i_ovlp_max_one=4, i_in_ang_rng=4, i_indiv_sim=4, i_orie_sim=5,
                                             correlation=0.5, area_perc=0.05), {})


def test_non_default_db_config_full_k_relaxed(cc, oracle):
    """nnk_ = 64 (the build's upper bound), q_levels_ = [2, 3, 4], relaxed gates and looser contour similarity."""
    _variant(cc, oracle, 64, (2, 3, 4), 10, dict(i_ovlp_sum=2, i_ovlp_max_one=2, i_in_ang_rng=2, i_indiv_sim=2, i_orie_sim=3,
                                                 correlation=0.1, area_perc=0.01, neg_est_dist=-8.0),
             dict(ta_cell_cnt=12.0, tp_cell_cnt=0.4, tp_eigval=0.4, ta_h_bar=0.6, ta_rcom=0.8, tp_rcom=0.5))


def _load_query_fixture(L):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "query_fixture.npz"))
    desc = np.frombuffer(z["desc"].tobytes(), dtype=L.scan_desc_dt)
    r1 = np.frombuffer(z["res"].tobytes(), dtype=L.query_result_v1_dt)  # stored before `flags` joined the record
    res = np.zeros(len(r1), L.query_result_dt)
    for f in r1.dtype.names:
        res[f] = r1[f]
    d = L.default_db_cfg()
    d.min_elapse, d.max_elapse = float(z["elapse"][0]), float(z["elapse"][1])
    return desc, z["ts"], res, d


def _same_result(exp, got, tol):
    for f in INT_FIELDS:
        assert exp[f] == got[f], (f, exp[f], got[f])
    if exp["n_res"]:
        assert abs(exp["correlation"] - got["correlation"]) < tol and np.abs(exp["tf"] - got["tf"]).max() < tol


@pytest.mark.parametrize("knn_mode", [0, 2])  # K3: one wave per search | tiled (matrix-core prefilter)
def test_golden_query_fixture(oracle, knn_mode, monkeypatch):
    """Committed descriptors + expected results (tests/golden/make_query_golden.py): the oracle, replaying the driver loop
    from the descriptors, still reproduces them, and the emulated query kernels match them."""
    L = oracle.L
    desc, ts, exp, d = _load_query_fixture(L)
    n = len(desc)
    assert n == 64 and (exp["n_res"] > 0).sum() == 31
    odb = oracle.DB(d)
    for i in range(n):
        s = oracle.Scan.from_desc(desc[i], int_id=i)
        _same_result(exp[i], odb.query(s), 1e-12)
        odb.add_scan(s, ts[i])
        odb.push_and_balance(i, ts[i])
    monkeypatch.setenv("CC_KNN_MODE", str(knn_mode))
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, d, cap=n)
    seeds = np.arange(n, dtype=np.int32)
    api.db_add(db, desc, ts, seeds)
    hit = np.nonzero(exp["n_res"] > 0)[0]
    qs = np.concatenate([hit[[0, 10, 20, 30]], [3, 40]]).astype(np.int32)
    got = api.db_query(db, desc[qs], qs)
    for k, qi in enumerate(qs):
        _same_result(exp[qi], got[k], 1e-6)
    # the same queries as two submitted batches, collected by one wait (cc_db_query_submit / cc_db_query_wait): a lane's
    # chunk of the first batch is collected when the second batch needs the lane
    r1, k1 = api.db_query_submit(db, desc[qs[:4]], qs[:4])
    r2, k2 = api.db_query_submit(db, desc[qs[3:]], qs[3:])
    api.db_query_wait(db)
    assert np.array_equal(r1.tobytes(), got[:4].tobytes()) and np.array_equal(r2.tobytes(), got[3:].tobytes())


def test_append_while_a_submitted_batch_is_in_flight(oracle):
    """cc_db_add_scans does not collect submitted chunks (it never touches what they read: the sorted key view is
    double-buffered, everything else is append-only): the batch submitted before the append still gets the results of its
    own epochs at the wait, and queries submitted afterwards see the appended scans."""
    L = oracle.L
    desc, ts, exp, d = _load_query_fixture(L)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, d, cap=len(desc))
    seeds = np.arange(len(desc), dtype=np.int32)
    api.db_add(db, desc[:60], ts[:60], seeds[:60])
    hit = np.nonzero(exp["n_res"][:60] > 0)[0]
    qs = hit[[2, 12, 22]].astype(np.int32)
    res, keep = api.db_query_submit(db, desc[qs], qs)
    api.db_add(db, desc[60:], ts[60:], seeds[60:])
    api.db_query_wait(db)
    for k, qi in enumerate(qs):
        _same_result(exp[qi], res[k], 1e-6)
    got = api.db_query(db, desc[[62]], np.array([62], np.int32))
    _same_result(exp[62], got[0], 1e-6)


def test_online_loop_in_sub_batches(oracle):
    """The online loop as bench.py --workload seq and the GPU replay test run it: per sub-batch append, then submit the
    sub-batch's queries at their own epochs, nothing collected until the end (two appends happen while earlier chunks are
    still uncollected, so both buffers of the sorted view get rewritten)."""
    L = oracle.L
    desc, ts, exp, d = _load_query_fixture(L)
    n = len(desc)
    api = emu_api.EmuApi(L)
    ctx = api.create(max_batch=8)
    db = api.db_create(ctx, d, cap=n)
    seeds = np.arange(n, dtype=np.int32)
    hit = set(np.nonzero(exp["n_res"] > 0)[0].tolist())
    outs = []
    for b0 in range(0, n, 16):
        api.db_add(db, desc[b0:b0 + 16], ts[b0:b0 + 16], seeds[b0:b0 + 16])
        hs = [i for i in range(b0, b0 + 16) if i in hit]
        qs = np.asarray((hs[:1] + hs[-1:] if hs else []) + [b0 + 5], np.int32)
        r, keep = api.db_query_submit(db, desc[qs], qs)
        outs.append((qs, r, keep))
    api.db_query_wait(db)
    n_hit = 0
    for qs, r, _ in outs:
        for k, qi in enumerate(qs):
            _same_result(exp[qi], r[k], 1e-6)
            n_hit += int(exp["n_res"][qi] > 0)
    assert n_hit >= 4
