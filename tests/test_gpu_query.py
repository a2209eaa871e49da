"""GPU parity of the full path (ingest -> DB -> batched query) against the CPU oracle's replay of the
reference driver loop (test/batch_bin_test.cpp:105-247), through the C-ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT_FIELDS = ["n_res", "cand_gidx", "cand_aft_check1", "cand_aft_check2", "cand_aft_check3", "n_cand_pose", "n_cand_tidy",
              "n_knn_hits"]


@pytest.fixture(scope="module")
def loop_sequence(cc):
    """420 synthetic Velodyne-64 scans (120k points each) on a 200 m loop: laps 2+ revisit lap 1."""
    import torch
    w = cc.synth.World(loop_len=200.0)
    xyzi, poses, ts = cc.synth.make_sequence(420, world=w, device="cuda")
    torch.cuda.synchronize()
    return xyzi, poses, ts


def test_sequence_matches_oracle(cc, oracle, loop_sequence):
    import torch
    xyzi, poses, ts = loop_sequence
    n, P = xyzi.shape[0], xyzi.shape[1]
    offs = np.arange(n + 1, dtype=np.int64) * P
    seeds = np.arange(n, dtype=np.int32)
    ctx = cc.Context(0, max_batch=128)
    desc = ctx.ingest(xyzi.reshape(-1, 4), offs)
    db = cc.Database(ctx, capacity=512)
    db.add_scans(desc, ts, seeds)
    res, knn, cnt = db.query(desc, seeds, want_knn=True)   # scan i queries the DB as it was after i scans
    torch.cuda.synchronize()
    ores, timers, odesc = oracle.run_sequence(xyzi.cpu().numpy().reshape(-1, 4), offs, ts, seeds, want_desc=True)
    # bookkeeping state
    odb = oracle.DB()
    for i in range(n):
        s = oracle.Scan(xyzi[i].cpu().numpy(), int_id=i, keep_cells=False)
        s.clear_image()
        odb.add_scan(s, ts[i])
        odb.push_and_balance(i, ts[i])
    osz, org = odb.bucket_state()
    gsz, grg = db.bucket_state()
    assert np.array_equal(osz, gsz) and np.array_equal(org, grg)
    assert (ores["n_res"] > 0).sum() > 50, "sequence should contain loop closures"
    bad = []
    for i in range(n):
        for f in INT_FIELDS:
            if ores[f][i] != res[f][i]:
                bad.append("query %d: %s oracle=%d got=%d" % (i, f, ores[f][i], res[f][i]))
        if ores["n_res"][i]:
            if abs(ores["correlation"][i] - res["correlation"][i]) > 1e-4:
                bad.append("query %d: correlation %g vs %g" % (i, ores["correlation"][i], res["correlation"][i]))
            if np.abs(ores["tf"][i] - res["tf"][i]).max() > 1e-4:
                bad.append("query %d: tf %s vs %s" % (i, ores["tf"][i], res["tf"][i]))
    assert not bad, "%d mismatches\n" % len(bad) + "\n".join(bad[:40])
    # loop closures are geometrically right
    hit = np.nonzero(res["n_res"] > 0)[0]
    d = np.hypot(poses[hit, 0] - poses[res["cand_gidx"][hit], 0], poses[hit, 1] - poses[res["cand_gidx"][hit], 1])
    assert (d < 5.0).mean() > 0.9
    db.close()
    ctx.close()


def test_knn_matches_oracle_db(cc, oracle, loop_sequence):
    """K3 hit lists (ids, order, squared distances) vs the oracle's bucketed search at a few epochs."""
    xyzi, poses, ts = loop_sequence
    n, P = 330, xyzi.shape[1]
    offs = np.arange(n + 1, dtype=np.int64) * P
    ctx = cc.Context(0, max_batch=128)
    desc = ctx.ingest(xyzi[:n].reshape(-1, 4), offs)
    db = cc.Database(ctx, capacity=512)
    db.add_scans(desc, ts[:n], np.arange(n, dtype=np.int32))
    odb = oracle.DB()
    scans = []
    qs = [260, 300, 329]
    expect = {}
    for i in range(n):
        s = oracle.Scan(xyzi[i].cpu().numpy(), int_id=i, keep_cells=False)
        s.clear_image()
        scans.append(s)
        if i in qs:
            expect[i] = odb.query(s, want_knn=True)
        odb.add_scan(s, ts[i])
        odb.push_and_balance(i, ts[i])
    res, knn, cnt = db.query(desc[qs], np.asarray(qs, np.int32), want_knn=True)
    for k, qi in enumerate(qs):
        ores, oknn, ocnt = expect[qi]
        assert np.array_equal(ocnt, cnt[k]), (qi, ocnt, cnt[k])
        for ll in range(3):
            for seq in range(6):
                m = ocnt[ll, seq]
                a, b = oknn[ll, seq, :m], knn[k, ll, seq, :m]
                assert np.array_equal(a["gidx"], b["gidx"]) and np.array_equal(a["seq"], b["seq"]), (qi, ll, seq)
                assert np.allclose(a["dist_sq"], b["dist_sq"], rtol=1e-5, atol=1e-6)
    db.close()
    ctx.close()


def test_golden_query_fixture(cc):
    """Committed descriptors of a 64-scan looping sequence + the expected result of every query (tests/golden/
    make_query_golden.py; checked against the oracle by the CPU suite): scan i queries the DB as it was after i scans."""
    import torch
    from test_emu_query import _load_query_fixture, _same_result
    L = cc.L
    desc, ts, exp, d = _load_query_fixture(L)
    n = len(desc)
    ddesc = torch.from_numpy(np.frombuffer(desc.tobytes(), np.uint8).reshape(n, cc.DESC_BYTES).copy()).cuda()
    ctx = cc.Context(0, max_batch=8)
    db = cc.Database(ctx, cfg=d, capacity=n)
    seeds = np.arange(n, dtype=np.int32)
    db.add_scans(ddesc, ts, seeds)
    got = db.query(ddesc, seeds)
    assert (exp["n_res"] > 0).sum() == 31
    for i in range(n):
        _same_result(exp[i], got[i], 1e-4)
    # the asynchronous form (cc_db_query_submit / cc_db_query_wait): three batches in flight over the two lanes, the query
    # descriptors of a batch overwritten as soon as the submit returned
    parts = [(0, 24), (24, 40), (40, 64)]
    buf = torch.empty_like(ddesc[:24])
    outs = []
    for a, b in parts:
        buf[:b - a].copy_(ddesc[a:b])
        outs.append(db.query_submit(buf[:b - a], seeds[a:b]))
    buf.zero_()
    db.query_wait()
    for (a, b), r in zip(parts, outs):
        assert r.tobytes() == got[a:b].tobytes()


def _seq_vs_oracle(cc, oracle, xyzi, ts, mcfg=None, dcfg=None, min_hits=10):
    import torch
    n, P = xyzi.shape[0], xyzi.shape[1]
    offs = np.arange(n + 1, dtype=np.int64) * P
    seeds = np.arange(n, dtype=np.int32)
    ctx = cc.Context(0, mcfg, max_batch=128)
    desc = ctx.ingest(xyzi.reshape(-1, 4), offs)
    db = cc.Database(ctx, cfg=dcfg, capacity=n)
    db.add_scans(desc, ts, seeds)
    res = db.query(desc, seeds)
    torch.cuda.synchronize()
    d = cc.desc_to_numpy(desc)
    assert (d["flags"] == 0).all()
    ores, _, odesc = oracle.run_sequence(xyzi.cpu().numpy().reshape(-1, 4), offs, ts, seeds, mcfg=mcfg, dcfg=dcfg, want_desc=True)
    assert (ores["n_res"] > 0).sum() >= min_hits, "sequence should contain loop closures (%d)" % (ores["n_res"] > 0).sum()
    bad = []
    for i in range(n):
        for f in INT_FIELDS:
            if ores[f][i] != res[f][i]:
                bad.append("query %d: %s oracle=%d got=%d" % (i, f, ores[f][i], res[f][i]))
        if ores["n_res"][i]:
            if abs(ores["correlation"][i] - res["correlation"][i]) > 1e-4:
                bad.append("query %d: correlation %g vs %g" % (i, ores["correlation"][i], res["correlation"][i]))
            if np.abs(ores["tf"][i] - res["tf"][i]).max() > 1e-4:
                bad.append("query %d: tf %s vs %s" % (i, ores["tf"][i], res["tf"][i]))
    assert not bad, "%d mismatches\n" % len(bad) + "\n".join(bad[:40])
    db.close()
    ctx.close()
    return ores, res


def test_sequence_mulran_config(cc, oracle):
    """BASELINE config 4's parameters (config/batch_bin_test_config.yaml:30-31, the MulRan variant): lv_grads_ =
    [1.0, 2.5, 4.0, 5.5, 7.0, 8.5] and ta_h_bar = 0.75, on a looping full-size sequence."""
    L = cc.L
    mcfg = L.default_manager_cfg()
    for i, v in enumerate([1.0, 2.5, 4.0, 5.5, 7.0, 8.5]):
        mcfg.lv_grads[i] = v
    dcfg = L.default_db_cfg()
    dcfg.cont_sim.ta_h_bar = 0.75
    w = cc.synth.World(loop_len=200.0)
    # MulRan's sensor: Ouster OS1-64, 64 x 1024 rays, +-16.6 deg (a Velodyne's +2 deg never sees above 4.6 m inside the BEV,
    # so the taller level set would stay empty)
    xyzi, poses, ts = cc.synth.make_sequence(330, world=w, device="cuda", beams=64, azim=1024, elev_deg=(16.6, -16.6))
    ores, _ = _seq_vs_oracle(cc, oracle, xyzi, ts, mcfg=mcfg, dcfg=dcfg)


def test_sequence_dense_world(cc, oracle):
    """The cluttered bench world (bench.py --workload dense, tens of contours per level) on a 150 m loop."""
    w = cc.synth.World(dense=True, loop_len=150.0)
    xyzi, poses, ts = cc.synth.make_sequence(330, world=w, device="cuda")
    _seq_vs_oracle(cc, oracle, xyzi, ts)
