"""The fused HIP ray caster of the synthetic scans (tools/synth_hip, bench / test plumbing) against the torch ops it
replaces (synth.cast_scan): same hits, same ranges (no noise), in the sparse and in the dense world."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dense", [False, True, "kitti"])
def test_hip_caster_matches_torch(cc, dense, monkeypatch):
    import torch
    S = cc.synth
    if S._hip_caster() is None:
        pytest.skip("libcc_synth.so not built")
    kitti = dense == "kitti"
    w = S.World(kitti=True) if kitti else S.World(dense=dense)
    idx = [0, 77, 1234]
    a, pa, ta = S.make_sequence(0, world=w, device="cuda", indices=idx, noise_sigma=0.0)
    x, y, yaw = w.path(max(idx) + 1) if kitti else S.trajectory(max(idx) + 1, loop_len=w.loop_len, tile=w.tile)
    for k, i in enumerate(idx):
        # the kitti world's porous volumes (crowns, bushes) end a ray at a depth drawn from a hash of (scan, ray, object)
        b = S.cast_scan(w, (x[i], y[i], yaw[i]), device="cuda", noise_sigma=0.0, scan_seed=20260926 * 1000003 + i)
        ha, hb = a[k, :, 0] < 999.0, b[:, 0] < 999.0
        assert (ha != hb).float().mean().item() < 2e-3            # rays grazing an edge may fall either way
        both = ha & hb
        d = (a[k, both, :3] - b[both, :3]).norm(dim=1)
        # porous foliage (dense world) is decided by the same integer hash; ranges agree to float rounding except where a ray
        # grazes an object edge
        assert (d > 1e-2).float().mean().item() < 5e-3, (d > 1e-2).float().mean().item()
        assert 0.3 < ha.float().mean().item() < 1.0
    assert np.allclose(ta, np.asarray(idx) / 10.0)
    # noise: zero-mean, sigma as asked for
    n, _, _ = S.make_sequence(0, world=w, device="cuda", indices=[77], noise_sigma=0.02)
    hit = (n[0, :, 0] < 999.0) & (a[1, :, 0] < 999.0)
    dr = n[0, hit, :3].norm(dim=1) - a[1, hit, :3].norm(dim=1)
    assert abs(dr.mean().item()) < 2e-3 and 0.015 < dr.std().item() < 0.025
    assert 0.4 < n[0, :, 3].mean().item() < 0.6
