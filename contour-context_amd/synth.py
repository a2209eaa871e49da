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
    def __init__(self, seed=20260926, tile=1000.0, density=1.0 / 150.0, loop_len=1500.0, clearance=5.0):
        rng = np.random.Generator(np.random.PCG64(seed))
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


def _ray_dirs(beams, azim, device):
    elev = torch.linspace(math.radians(2.0), math.radians(-24.8), beams, device=device, dtype=torch.float32)
    az = torch.arange(azim, device=device, dtype=torch.float32) * (2 * math.pi / azim)
    ce, se = torch.cos(elev)[:, None], torch.sin(elev)[:, None]
    d = torch.stack([ce * torch.cos(az)[None, :], ce * torch.sin(az)[None, :], se.expand(beams, azim)], dim=-1)
    return d.reshape(-1, 3)  # [N,3], sensor frame


@torch.no_grad()
def cast_scan(world, pose, beams=64, azim=1875, device="cpu", noise_sigma=0.02, gen=None, chunk=32768):
    """One scan at pose=(x, y, yaw). Returns float32 [beams*azim, 4] (x,y,z,intensity) in the sensor frame."""
    px, py, yaw = float(pose[0]), float(pose[1]), float(pose[2])
    dev = torch.device(device)
    d_s = _ray_dirs(beams, azim, dev)
    c, s = math.cos(yaw), math.sin(yaw)
    # world-frame directions
    dw = torch.stack([c * d_s[:, 0] - s * d_s[:, 1], s * d_s[:, 0] + c * d_s[:, 1], d_s[:, 2]], dim=-1)
    o = torch.tensor([px, py, SENSOR_H], device=dev, dtype=torch.float32)
    # cull objects
    bx = world.boxes
    keep = (bx[:, 3] > px - MAX_RANGE) & (bx[:, 0] < px + MAX_RANGE) & (bx[:, 4] > py - MAX_RANGE) & (bx[:, 1] < py + MAX_RANGE)
    boxes = torch.from_numpy(bx[keep]).to(dev)
    cy = world.cyls
    keepc = (np.abs(cy[:, 0] - px) < MAX_RANGE) & (np.abs(cy[:, 1] - py) < MAX_RANGE)
    cyls = torch.from_numpy(cy[keepc]).to(dev)
    N = dw.shape[0]
    t_best = torch.full((N,), float("inf"), device=dev)
    for i0 in range(0, N, chunk):
        d = dw[i0:i0 + chunk]
        tb = torch.full((d.shape[0],), float("inf"), device=dev)
        # ground plane z = 0
        tg = torch.where(d[:, 2] < -1e-6, -SENSOR_H / d[:, 2], torch.full_like(d[:, 2], float("inf")))
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
                  start=0, world=None, noise_sigma=0.02):
    """Scans `start .. start+n_scans-1` of the seeded trajectory: returns (xyzi [n, P, 4] f32, poses [n,3], ts [n])."""
    world = world or World(seed, loop_len=loop_len)
    total = start + n_scans
    x, y, yaw = trajectory(total, step=step, loop_len=world.loop_len, tile=world.tile)
    gen = torch.Generator(device=torch.device(device))
    out = []
    for i in range(start, total):
        gen.manual_seed(seed * 1000003 + i)
        out.append(cast_scan(world, (x[i], y[i], yaw[i]), beams, azim, device, noise_sigma, gen))
    xyzi = torch.stack(out, dim=0)
    poses = np.stack([x[start:total], y[start:total], yaw[start:total]], axis=1)
    ts = np.arange(start, total, dtype=np.float64) / 10.0
    return xyzi, poses, ts
