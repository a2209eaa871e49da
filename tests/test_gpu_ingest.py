"""GPU parity of the ingest path (K1 rasterise + K2 contours/keys/BCI) against the CPU oracle,
through the C-ABI (cc_ingest_batch)."""
import numpy as np
import pytest

from parity import compare_desc, terrain_scan

pytestmark = pytest.mark.gpu


def _run(cc, oracle, scans, cfg=None):
    import torch
    cfg = cfg or cc.L.default_manager_cfg()
    ctx = cc.Context(0, cfg, max_batch=8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in scans])]).astype(np.int64)
    x = torch.from_numpy(np.concatenate(scans, 0)).cuda()
    # first WITHOUT the debug outputs, on the fresh context: K1 then writes its dense image / positions only for scans whose
    # active cells overflow its list, and a scan the list kernel hands on gets them rebuilt from the list (cc_k_contours_mid)
    plain = ctx.ingest(x, offs).clone()
    desc, dbg = ctx.ingest(x, offs, debug=True)
    torch.cuda.synchronize()
    d = cc.desc_to_numpy(desc)
    dp = cc.desc_to_numpy(plain)
    report = []
    for i, s in enumerate(scans):
        o = oracle.Scan(s, cfg)
        ob, opix = o.bev()
        if not np.array_equal(ob, dbg["bev"][i].cpu().numpy()):
            report.append("scan %d: bev differs" % i)
        if not np.array_equal(opix, dbg["pix_rc"][i].cpu().numpy()):
            report.append("scan %d: pix_rc differs" % i)
        if not np.array_equal(o.labels(), dbg["labels"][i].cpu().numpy()):
            report.append("scan %d: canonical label images differ" % i)
        report += ["scan %d: %s" % (i, m) for m in compare_desc(o.desc()[0], d[i], float_exact=False)]
        report += ["scan %d (no debug outputs): %s" % (i, m) for m in compare_desc(o.desc()[0], dp[i], float_exact=False)]
    ctx.close()
    return report, d


def test_ingest_synthetic_velodyne(cc, oracle):
    w = cc.synth.World(loop_len=200.0)
    xyzi, _, _ = cc.synth.make_sequence(3, world=w, device="cuda", start=11)
    scans = [xyzi[i].cpu().numpy() for i in range(3)]
    report, d = _run(cc, oracle, scans)
    assert not report, "\n".join(report[:40])
    assert (d["flags"] == 0).all()


def test_ingest_terrain_many_contours(cc, oracle):
    scans = [terrain_scan(s) for s in range(4)] + [terrain_scan(100 + s, n=30000, scale=2.2, quant=0.25) for s in range(3)]
    report, d = _run(cc, oracle, scans)
    assert not report, "\n".join(report[:40])
    assert d["n_cont"].max() > 16  # exercises the introsort partition path


def test_ingest_ragged_and_edge(cc, oracle):
    rng = np.random.default_rng(5)
    tiny = np.zeros((11, 4), np.float32)                      # minimum accepted size, all in the blind zone
    far = np.full((50, 4), 1000.0, np.float32)                # every point rejected
    ties = np.tile(np.array([[10.2, 3.3, 1.0, 0], [10.7, 3.9, 1.0, 0], [10.4, 3.1, 1.0, 0]], np.float32), (40, 1))
    ties[:, :2] += rng.normal(0, 1e-3, ties[:, :2].shape).astype(np.float32)   # equal heights in one cell: first wins
    edge = np.array([[74.98, 74.98, 2.0, 0], [-74.98, -74.98, 2.0, 0], [74.995, 0, 2, 0], [-74.0, 10, 2, 0]] * 5, np.float32)
    scans = [tiny, far, ties, edge, terrain_scan(9, n=2000)]
    report, _ = _run(cc, oracle, scans)
    assert not report, "\n".join(report[:40])


def test_ingest_rejects_short_scan(cc):
    import torch
    ctx = cc.Context(0, max_batch=2)
    x = torch.zeros((10, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(cc.CCError):
        ctx.ingest(x, np.array([0, 10], np.int64))
    ctx.close()


def test_ingest_host_matches_device(cc, oracle):
    scans = [terrain_scan(21, n=20000)]
    ctx = cc.Context(0, max_batch=2)
    out = ctx.ingest_host(scans[0], np.array([0, len(scans[0])], np.int64))
    o = oracle.Scan(scans[0])
    assert not compare_desc(o.desc()[0], out[0], float_exact=False)
    ctx.close()


def _mulran_cfg(L):
    """config/batch_bin_test_config.yaml:31 (MulRan, Ouster-64): the alternative level set."""
    cfg = L.default_manager_cfg()
    for i, v in enumerate([1.0, 2.5, 4.0, 5.5, 7.0, 8.5]):
        cfg.lv_grads[i] = v
    return cfg


def test_ingest_mulran_level_set(cc, oracle):
    """BASELINE config 4's manager configuration on full-size synthetic scans and on tall terrain."""
    cfg = _mulran_cfg(cc.L)
    w = cc.synth.World(loop_len=200.0)
    xyzi, _, _ = cc.synth.make_sequence(2, world=w, device="cuda", start=40, beams=64, azim=1024, elev_deg=(16.6, -16.6))
    scans = [xyzi[i].cpu().numpy() for i in range(2)] + [terrain_scan(12, n=6000, scale=5.0, quant=0.5), terrain_scan(3, n=40000, scale=4.0)]
    report, d = _run(cc, oracle, scans, cfg=cfg)
    assert not report, "\n".join(report[:40])
    assert (d["n_cont"][2:, 5] > 0).all()


def test_ingest_dense_world(cc, oracle):
    """The cluttered bench world (bench.py --workload dense): tens of contours on every level."""
    w = cc.synth.World(dense=True)
    xyzi, _, _ = cc.synth.make_sequence(3, world=w, device="cuda", start=5000)
    scans = [xyzi[i].cpu().numpy() for i in range(3)]
    report, d = _run(cc, oracle, scans)
    assert not report, "\n".join(report[:40])
    assert (d["flags"] == 0).all() and d["n_cont"][:, :3].min() >= 20


def test_ingest_min_cont_cell_cnt_and_nan(cc, oracle):
    cfg = cc.L.default_manager_cfg()
    cfg.min_cont_cell_cnt = 6
    s = terrain_scan(2, n=20000, scale=1.2)
    s2 = s.copy()
    s2[::7, 2] = np.nan
    report, _ = _run(cc, oracle, [s, s2], cfg=cfg)
    assert not report, "\n".join(report[:40])


def test_ingest_resolutions_pow2_and_not(cc, oracle):
    """2.0 m cells (contour_mng.h:95, multiply-by-reciprocal instance of the rasteriser) and 1.5 x 0.75 m cells (IEEE
    division instance)."""
    s = terrain_scan(4, n=20000, scale=1.2)
    for rr, rc, n in ((2.0, 2.0, 74), (1.5, 0.75, 100)):
        cfg = cc.L.default_manager_cfg()
        cfg.reso_row, cfg.reso_col = rr, rc
        cfg.n_row, cfg.n_col = n, n
        report, d = _run(cc, oracle, [s], cfg=cfg)
        assert not report, "\n".join(report[:40])
        assert d["n_pix"][0] > 500
