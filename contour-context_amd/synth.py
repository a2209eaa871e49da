"""Seeded synthetic Velodyne-64-like scans.

World: axis-aligned boxes ("buildings", "cars") and vertical cylinders ("poles/trunks") on a ground
plane, placed by a PCG64 stream.  Sensor: `beams` x `azim` rays (64 x 1875 = 120 000 by default),
elevation linspace(+2 deg, -24.8 deg), 1.73 m above ground, 80 m max range, Gaussian range noise.
Rays that miss are emitted as far points outside the +-75 m BEV square so every scan has exactly
beams*azim points in KITTI .bin layout (x, y, z, intensity) f32.

The ray caster is written in torch so the bench can synthesise thousands of scans directly in HBM;
the same code runs on CPU for tests and fixtures.
"""
import math
import numpy as np
import torch

SENSOR_H = 1.73
MAX_RANGE = 80.0


class World:
    """`dense=True` is the second bench workload (SURVEY.md 8(d) density figures: 4-9 k occupied cells, tens to hundreds
    of contours per level): ~6x the object density, most of it vegetation-like clutter (thin trunks, bushes) that rays
    thread through to varied depths, low raised patches (embankments, kerbed lawns) whose tops the downward beams see,
    and walls/fences; the default world keeps SURVEY's sparse 1 object / 150 m2."""

    def __init__(self, seed=20260926, tile=1000.0, density=1.0 / 150.0, loop_len=1500.0, clearance=5.0, dense=False):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.dense = bool(dense)
        if dense:
            self._init_dense(rng, tile, loop_len, clearance)
            return
        # road corridor: the sensor path must not run through objects
        rx, ry, _ = trajectory(int(loop_len), step=1.0, loop_len=loop_len, tile=tile, jitter=False)
        n_obj = int(tile * tile * density)
        kind = rng.random(n_obj)
        cx = rng.uniform(-tile / 2, tile / 2, n_obj)
        cy = rng.uniform(-tile / 2, tile / 2, n_obj)
        boxes, cyls = [], []
        def clear_of_road(x0, y0, x1, y1):
            ddx = np.maximum(np.maximum(x0 - rx, rx - x1), 0.0)
            ddy = np.maximum(np.maximum(y0 - ry, ry - y1), 0.0)
            return np.min(ddx * ddx + ddy * ddy) > clearance * clearance

        for k, x, y in zip(kind, cx, cy):
            if k < 0.45:  # building
                sx, sy = rng.uniform(5, 40), rng.uniform(5, 40)
                h = rng.uniform(3, 20)
                if clear_of_road(x - sx / 2, y - sy / 2, x + sx / 2, y + sy / 2):
                    boxes.append((x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, h))
            elif k < 0.70:  # car
                if rng.random() < 0.5:
                    sx, sy = 4.0, 1.8
                else:
                    sx, sy = 1.8, 4.0
                if clear_of_road(x - sx / 2, y - sy / 2, x + sx / 2, y + sy / 2):
                    boxes.append((x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, 1.5))
            else:  # pole / trunk
                r, h = rng.uniform(0.15, 0.5), rng.uniform(3, 10)
                if clear_of_road(x - r, y - r, x + r, y + r):
                    cyls.append((x, y, r, h))
        self.boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
        self.cyls = np.asarray(cyls, dtype=np.float32).reshape(-1, 4)
        self.tile = tile
        self.loop_len = loop_len


def _world_init_dense(self, rng, tile, loop_len, clearance, density=1.0 / 22.0):
    rx, ry, _ = trajectory(int(loop_len), step=1.0, loop_len=loop_len, tile=tile, jitter=False)
    # objects only matter within sensor range of the path: a coarse occupancy mask of the corridor keeps the count down
    g = 20.0
    ng = int(tile / g) + 1
    near = np.zeros((ng, ng), bool)
    gi = np.clip(((rx + tile / 2) / g).astype(int), 0, ng - 1)
    gj = np.clip(((ry + tile / 2) / g).astype(int), 0, ng - 1)
    R = int(np.ceil(MAX_RANGE / g)) + 1
    for di in range(-R, R + 1):
        for dj in range(-R, R + 1):
            near[np.clip(gi + di, 0, ng - 1), np.clip(gj + dj, 0, ng - 1)] = True
    n_obj = int(tile * tile * density)
    kind = rng.random(n_obj)
    cx = rng.uniform(-tile / 2, tile / 2, n_obj)
    cy = rng.uniform(-tile / 2, tile / 2, n_obj)
    par = rng.random((n_obj, 4))
    ok = near[np.clip(((cx + tile / 2) / g).astype(int), 0, ng - 1), np.clip(((cy + tile / 2) / g).astype(int), 0, ng - 1)]
    boxes, cyls = [], []
    c2 = clearance * clearance

    def clear_of_road(x0, y0, x1, y1):
        ddx = np.maximum(np.maximum(x0 - rx, rx - x1), 0.0)
        ddy = np.maximum(np.maximum(y0 - ry, ry - y1), 0.0)
        return np.min(ddx * ddx + ddy * ddy) > c2

    for i in np.nonzero(ok)[0]:
        k, x, y = kind[i], cx[i], cy[i]
        a, b, c, d = par[i]
        if k < 0.06:  # building
            sx, sy, h = 5 + 30 * a, 5 + 30 * b, 3 + 17 * c
            bb = (x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, h)
        elif k < 0.14:  # car
            sx, sy = (4.0, 1.8) if a < 0.5 else (1.8, 4.0)
            bb = (x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, 1.5)
        elif k < 0.22:  # wall / fence / hedge: long and thin
            ln, th, h = 4 + 16 * a, 0.3 + 0.5 * b, 1.0 + 2.0 * c
            sx, sy = (ln, th) if d < 0.5 else (th, ln)
            bb = (x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, h)
        elif k < 0.30:  # raised patch: embankment, kerbed lawn
            sx, sy, h = 4 + 14 * a, 4 + 14 * b, 0.3 + 1.4 * c
            bb = (x - sx / 2, y - sy / 2, 0.0, x + sx / 2, y + sy / 2, h)
        else:  # trunk, pole, bush
            bb = None
            r = 0.15 + 0.9 * a * a
            h = 1.0 + 7.0 * b
            if clear_of_road(x - r, y - r, x + r, y + r):
                cyls.append((x, y, r, h))
        if bb is not None and clear_of_road(bb[0], bb[1], bb[3], bb[4]):
            boxes.append(bb)
    self.boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
    self.cyls = np.asarray(cyls, dtype=np.float32).reshape(-1, 4)
    self.tile = tile
    self.loop_len = loop_len
    # terrain relief: a few long waves, +-2.5 m; objects stand on it (bases sunk 1.5 m so nothing floats)
    nw = 6
    self.relief = np.stack([rng.uniform(0.25, 0.6, nw), 2 * np.pi / rng.uniform(45.0, 140.0, nw), rng.uniform(0, 2 * np.pi, nw),
                            2 * np.pi / rng.uniform(45.0, 140.0, nw), rng.uniform(0, 2 * np.pi, nw)], axis=1).astype(np.float32)
    gb = self.ground(self.boxes[:, [0, 3]].mean(1), self.boxes[:, [1, 4]].mean(1)) if len(self.boxes) else np.zeros(0, np.float32)
    self.boxes[:, 2] = gb - 1.5
    self.boxes[:, 5] += gb
    self.cyl_base = self.ground(self.cyls[:, 0], self.cyls[:, 1]).astype(np.float32) if len(self.cyls) else np.zeros(0, np.float32)


def _world_ground(self, x, y):
    """terrain height at (x, y): numpy or torch arrays"""
    if self.relief is None:
        return x * 0
    is_t = torch.is_tensor(x)
    sin = torch.sin if is_t else np.sin
    z = x * 0
    for a, fx, px_, fy, py_ in self.relief.tolist():
        z = z + a * sin(fx * x + px_) * sin(fy * y + py_)
    return z


World._init_dense = _world_init_dense
World.ground = _world_ground
World.relief = None
World.cyl_base = None


def trajectory(n_scans, step=1.0, loop_len=1500.0, tile=1000.0, seed=7, jitter=True):
    """Closed figure-eight traversed repeatedly at `step` m/scan; each lap is shifted sideways by a
    small seeded offset so revisits are near (not exact) repeats, with crossings at other headings.
    Returns (x, y, yaw) float64 arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    # arclength-parameterise a lemniscate-like curve
    A, B = 0.36 * tile, 0.22 * tile
    u = np.linspace(0, 2 * np.pi, 20001)
    px, py = A * np.sin(u), B * np.sin(2 * u)
    seg = np.hypot(np.diff(px), np.diff(py))
    s = np.concatenate([[0], np.cumsum(seg)])
    scale = loop_len / s[-1]
    px, py, s = px * scale, py * scale, s * scale
    d = np.arange(n_scans) * step
    lap = np.floor(d / loop_len).astype(int)
    ds = d - lap * loop_len
    x = np.interp(ds, s, px)
    y = np.interp(ds, s, py)
    x2 = np.interp(ds + 0.5, s, px)
    y2 = np.interp(ds + 0.5, s, py)
    yaw = np.arctan2(y2 - y, x2 - x)
    n_lap = lap.max() + 1
    # independent streams so that scan i's pose does not depend on how many scans are requested
    off = rng.normal(0, 0.6, (n_lap, 2))
    off[0] = 0
    if jitter:
        x = x + off[lap, 0]
        y = y + off[lap, 1]
        rng_yaw = np.random.Generator(np.random.PCG64(seed + 1000))
        yaw = yaw + rng_yaw.normal(0, 0.01, n_scans)
    return x, y, yaw


def _ray_dirs(beams, azim, device, hdl64=False, elev_deg=None):
    if elev_deg is not None:  # another sensor, e.g. MulRan's Ouster OS1-64: (+16.6, -16.6) deg
        elev = torch.linspace(math.radians(elev_deg[0]), math.radians(elev_deg[1]), beams, device=device, dtype=torch.float32)
    elif hdl64 and beams % 2 == 0:
        # the HDL-64E's two laser blocks: upper half 1/3 deg apart from +2 deg, lower half 1/2 deg apart down to -24.33 deg
        # (more beams reach beyond the first 15 m of ground than with a uniform fan)
        h = beams // 2
        up = torch.linspace(2.0, 2.0 - (h - 1) * (10.33 / 31.0) * (32.0 / h), h)
        lo = torch.linspace(-8.83, -24.33, h)
        elev = torch.deg2rad(torch.cat([up, lo])).to(device=device, dtype=torch.float32)
    else:
        elev = torch.linspace(math.radians(2.0), math.radians(-24.8), beams, device=device, dtype=torch.float32)
    az = torch.arange(azim, device=device, dtype=torch.float32) * (2 * math.pi / azim)
    ce, se = torch.cos(elev)[:, None], torch.sin(elev)[:, None]
    d = torch.stack([ce * torch.cos(az)[None, :], ce * torch.sin(az)[None, :], se.expand(beams, azim)], dim=-1)
    return d.reshape(-1, 3)  # [N,3], sensor frame


@torch.no_grad()
def cast_scan(world, pose, beams=64, azim=1875, device="cpu", noise_sigma=0.02, gen=None, chunk=32768, elev_deg=None):
    """One scan at pose=(x, y, yaw). Returns float32 [beams*azim, 4] (x,y,z,intensity) in the sensor frame."""
    px, py, yaw = float(pose[0]), float(pose[1]), float(pose[2])
    dev = torch.device(device)
    d_s = _ray_dirs(beams, azim, dev, hdl64=getattr(world, "dense", False), elev_deg=elev_deg)
    c, s = math.cos(yaw), math.sin(yaw)
    # world-frame directions
    dw = torch.stack([c * d_s[:, 0] - s * d_s[:, 1], s * d_s[:, 0] + c * d_s[:, 1], d_s[:, 2]], dim=-1)
    relief = getattr(world, "relief", None) is not None
    gz = float(world.ground(np.float64(px), np.float64(py))) if relief else 0.0
    o = torch.tensor([px, py, gz + SENSOR_H], device=dev, dtype=torch.float32)
    # cull objects
    bx = world.boxes
    keep = (bx[:, 3] > px - MAX_RANGE) & (bx[:, 0] < px + MAX_RANGE) & (bx[:, 4] > py - MAX_RANGE) & (bx[:, 1] < py + MAX_RANGE)
    boxes = torch.from_numpy(bx[keep]).to(dev)
    cy = world.cyls
    keepc = (np.abs(cy[:, 0] - px) < MAX_RANGE) & (np.abs(cy[:, 1] - py) < MAX_RANGE)
    cyls = torch.from_numpy(cy[keepc]).to(dev)
    cbase = torch.from_numpy(world.cyl_base[keepc]).to(dev) if relief else None
    cid = torch.from_numpy(np.nonzero(keepc)[0]).to(dev) if relief else None
    N = dw.shape[0]
    t_best = torch.full((N,), float("inf"), device=dev)
    for i0 in range(0, N, chunk):
        d = dw[i0:i0 + chunk]
        tb = torch.full((d.shape[0],), float("inf"), device=dev)
        if not relief:
            # ground plane z = 0
            tg = torch.where(d[:, 2] < -1e-6, -SENSOR_H / d[:, 2], torch.full_like(d[:, 2], float("inf")))
        else:
            # terrain: all 0.5 m march samples of a ray at once, first sample below ground, linear interpolation, then
            # two secant refinements
            ns = int(MAX_RANGE / 0.5)
            tt = torch.arange(1, ns + 1, device=dev, dtype=torch.float32) * 0.5                       # [ns]
            fx = o[0] + d[:, 0:1] * tt[None, :]
            fy = o[1] + d[:, 1:2] * tt[None, :]
            f = o[2] + d[:, 2:3] * tt[None, :] - world.ground(fx, fy)                                  # [n, ns]
            below = f <= 0
            anyb = below.any(dim=1)
            first = torch.argmax(below.to(torch.uint8), dim=1)                                         # first sample below ground
            f1 = torch.gather(f, 1, first[:, None])[:, 0]
            f0 = torch.where(first > 0, torch.gather(f, 1, (first - 1).clamp(min=0)[:, None])[:, 0], torch.full_like(f1, SENSOR_H))
            t1 = tt[first]
            t0 = t1 - 0.5
            ts_ = t0 + 0.5 * f0 / (f0 - f1).clamp(min=1e-6)
            del f, fx, fy, below
            for _ in range(2):
                fa = o[2] + ts_ * d[:, 2] - world.ground(o[0] + ts_ * d[:, 0], o[1] + ts_ * d[:, 1])
                e = 0.05
                fb = o[2] + (ts_ + e) * d[:, 2] - world.ground(o[0] + (ts_ + e) * d[:, 0], o[1] + (ts_ + e) * d[:, 1])
                df = (fb - fa) / e
                ts_ = ts_ - fa / torch.where(df.abs() < 1e-4, torch.full_like(df, -1e-4), df)
            tg = torch.where(anyb, ts_.clamp(min=0.0), torch.full_like(ts_, float("inf")))
        tb = torch.minimum(tb, tg)
        if boxes.shape[0]:
            inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)  # [n,3]
            t0 = (boxes[None, :, 0:3] - o[None, None, :]) * inv[:, None, :]
            t1 = (boxes[None, :, 3:6] - o[None, None, :]) * inv[:, None, :]
            tmin = torch.minimum(t0, t1).amax(dim=-1)
            tmax = torch.maximum(t0, t1).amin(dim=-1)
            hit = (tmax >= tmin) & (tmax > 0)
            tt = torch.where(hit, torch.where(tmin > 0, tmin, tmax), torch.full_like(tmin, float("inf")))
            tb = torch.minimum(tb, tt.amin(dim=1))
        if cyls.shape[0]:
            ox = o[0] - cyls[None, :, 0]
            oy = o[1] - cyls[None, :, 1]
            dx, dy = d[:, 0:1], d[:, 1:2]
            a = dx * dx + dy * dy
            b = 2 * (ox * dx + oy * dy)
            cc = ox * ox + oy * oy - cyls[None, :, 2] ** 2
            disc = b * b - 4 * a * cc
            sq = torch.sqrt(torch.clamp(disc, min=0))
            tc = (-b - sq) / (2 * a + 1e-12)
            z = o[2] + tc * d[:, 2:3]
            if relief:
                ok = (disc > 0) & (tc > 0) & (z >= cbase[None, :] - 1.5) & (z <= cbase[None, :] + cyls[None, :, 3])
                # foliage is porous: a ray is stopped by a bush / crown (r > 0.45 m) with probability 0.3, decided by an
                # integer hash of (ray, object) so that a scan is reproducible
                ridx = torch.arange(i0, i0 + d.shape[0], device=dev, dtype=torch.int64)[:, None]
                hsh = ((ridx * 2654435761 + cid[None, :] * 40503 + 12345) >> 7) & 1023
                ok = ok & ((cyls[None, :, 2] <= 0.45) | (hsh < 307))
            else:
                ok = (disc > 0) & (tc > 0) & (z >= 0) & (z <= cyls[None, :, 3])
            tc = torch.where(ok, tc, torch.full_like(tc, float("inf")))
            tb = torch.minimum(tb, tc.amin(dim=1))
        t_best[i0:i0 + chunk] = tb
    hit = t_best < MAX_RANGE
    if noise_sigma > 0:
        if gen is None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(1234)
        noise = torch.randn(N, generator=gen, device=dev) * noise_sigma
    else:
        noise = torch.zeros(N, device=dev)
    t = torch.where(hit, t_best + noise, torch.zeros_like(t_best))
    p = d_s * t[:, None]
    far = torch.tensor([1000.0, 1000.0, 0.0], device=dev)
    p = torch.where(hit[:, None], p, far[None, :].expand(N, 3))
    inten = torch.rand(N, generator=gen, device=dev) if gen is not None else torch.rand(N, device=dev)
    return torch.cat([p, inten[:, None]], dim=1).contiguous()


def make_sequence(n_scans, beams=64, azim=1875, device="cpu", seed=20260926, step=1.0, loop_len=1500.0,
                  start=0, world=None, noise_sigma=0.02, elev_deg=None, indices=None):
    """Scans `start .. start+n_scans-1` of the seeded trajectory (or the scans `indices`, e.g. one rank's interleaved
    share): returns (xyzi [n, P, 4] f32, poses [n,3], ts [n]).  Scan i is the same whoever asks for it."""
    world = world or World(seed, loop_len=loop_len)
    idx = np.arange(start, start + n_scans) if indices is None else np.asarray(indices, dtype=np.int64)
    total = int(idx.max()) + 1 if len(idx) else 0
    x, y, yaw = trajectory(total, step=step, loop_len=world.loop_len, tile=world.tile)
    if torch.device(device).type == "cuda" and len(idx) and _hip_caster() is not None:
        # fused HIP ray caster (tools/synth_hip): same geometry, hash-based noise -- ~50x faster than the torch ops below
        xyzi = _cast_scans_hip(world, x[idx], y[idx], yaw[idx], seed * 1000003 + idx, beams, azim, torch.device(device), noise_sigma, elev_deg)
        return xyzi, np.stack([x[idx], y[idx], yaw[idx]], axis=1), idx.astype(np.float64) / 10.0
    gen = torch.Generator(device=torch.device(device))
    out = []
    for i in idx.tolist():
        gen.manual_seed(seed * 1000003 + i)
        out.append(cast_scan(world, (x[i], y[i], yaw[i]), beams, azim, device, noise_sigma, gen, elev_deg=elev_deg))
    xyzi = torch.stack(out, dim=0)
    poses = np.stack([x[idx], y[idx], yaw[idx]], axis=1)
    ts = idx.astype(np.float64) / 10.0
    return xyzi, poses, ts


_HIP = [False, None]


def _hip_caster():
    """ctypes handle of tools/synth_hip/libcc_synth.so (built by __graft_entry__.build()), or None: then torch casts."""
    import ctypes as C
    import os
    if _HIP[0]:
        return _HIP[1]
    _HIP[0] = True
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "synth_hip", "libcc_synth.so")
    if os.environ.get("CC_SYNTH_TORCH") == "1" or not os.path.exists(so):
        return None
    try:
        lib = C.CDLL(so)
        lib.sc_cast_scans.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        _HIP[1] = lib
    except OSError:
        _HIP[1] = None
    return _HIP[1]


def _cast_scans_hip(world, xs, ys, yaws, seeds, beams, azim, dev, noise_sigma, elev_deg, chunk=256):
    relief = getattr(world, "relief", None) is not None
    dirs = _ray_dirs(beams, azim, dev, hdl64=getattr(world, "dense", False), elev_deg=elev_deg).contiguous()
    N = dirs.shape[0]
    n = len(xs)
    out = torch.empty((n, N, 4), dtype=torch.float32, device=dev)
    bx, cy = world.boxes, world.cyls
    cbase = world.cyl_base if relief else np.zeros(len(cy), np.float32)
    cid_bits = np.arange(len(cy), dtype=np.int32).view(np.float32)
    wave = np.ascontiguousarray(world.relief, np.float32).reshape(-1) if relief else None
    lib = _hip_caster()
    stream = torch.cuda.current_stream(dev).cuda_stream
    for c0 in range(0, n, chunk):
        c1 = min(c0 + chunk, n)
        boxes, cyls, boff, coff, poses = [], [], [0], [0], []
        for i in range(c0, c1):
            px, py = float(xs[i]), float(ys[i])
            keep = (bx[:, 3] > px - MAX_RANGE) & (bx[:, 0] < px + MAX_RANGE) & (bx[:, 4] > py - MAX_RANGE) & (bx[:, 1] < py + MAX_RANGE)
            keepc = (np.abs(cy[:, 0] - px) < MAX_RANGE) & (np.abs(cy[:, 1] - py) < MAX_RANGE)
            boxes.append(bx[keep])
            cyls.append(np.concatenate([cy[keepc], cbase[keepc, None], cid_bits[keepc, None]], axis=1))
            boff.append(boff[-1] + int(keep.sum()))
            coff.append(coff[-1] + int(keepc.sum()))
            gz = float(world.ground(np.float64(px), np.float64(py))) if relief else 0.0
            poses.append((px, py, float(yaws[i]), gz))
        h_boxes = np.ascontiguousarray(np.concatenate(boxes) if boff[-1] else np.zeros((1, 6)), np.float32)
        h_cyls = np.ascontiguousarray(np.concatenate(cyls) if coff[-1] else np.zeros((1, 6)), np.float32)
        t_boxes = torch.from_numpy(h_boxes).to(dev)
        t_cyls = torch.from_numpy(h_cyls).to(dev)
        t_boff = torch.tensor(boff, dtype=torch.int32, device=dev)
        t_coff = torch.tensor(coff, dtype=torch.int32, device=dev)
        t_pose = torch.tensor(poses, dtype=torch.float32, device=dev)
        t_seed = torch.from_numpy(np.ascontiguousarray(seeds[c0:c1], np.int64)).to(dev)
        rc = lib.sc_cast_scans(dirs.data_ptr(), N, c1 - c0, t_pose.data_ptr(), t_boxes.data_ptr(), t_boff.data_ptr(), t_cyls.data_ptr(),
                               t_coff.data_ptr(), t_seed.data_ptr(), 1 if relief else 0, wave.ctypes.data if relief else None,
                               float(noise_sigma), out[c0:c1].data_ptr(), stream)
        if rc != 0:
            raise RuntimeError("sc_cast_scans failed")
        torch.cuda.current_stream(dev).synchronize()  # the staged lists go out of scope
    return out
