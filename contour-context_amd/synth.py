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

    def __init__(self, seed=20260926, tile=1000.0, density=1.0 / 150.0, loop_len=1500.0, clearance=5.0, dense=False, kitti=False,
                 **kitti_args):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.dense = bool(dense)
        self.kitti = bool(kitti)
        if kitti:
            self.dense = True  # HDL-64E beam table
            self._init_kitti(rng, seed, **kitti_args)
            return
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



# ---------------------------------------------------------------------------------------------------------------------
# KITTI-shaped world (SURVEY.md 8(d) value distributions: 4-9 k occupied cells, 50-150 contours on the low levels, ~18
# valid DB keys per scan, ~8 % of the scans of a 4 k-scan sequence revisit an earlier place): a residential street grid.
# Streets are 150 m apart on a 2 km x 2 km tile; the vehicle does a seeded random walk over the grid (no U-turns, filleted
# corners, right-hand lane), so revisits happen the way they do in a town: now and then a street is driven again, in
# either direction, and crossings are passed at right angles.  Between the streets: houses, parked cars, hedges / fences,
# poles, trunks, and -- what gives a real scan its thousands of occupied cells -- VOLUMETRIC vegetation: bushes and tree
# crowns are cylinders of a porous medium in which a ray ends at a random depth (Beer-Lambert, hash of (scan, ray,
# object)), so the upper beams fill a crown's whole footprint instead of drawing its outline.
KITTI_DEFAULTS = dict(tile=2000.0, block=150.0, explore=1.0, lane=1.5, fillet=8.0, road_half=4.6,
                      house=0.0012, car=1.0 / 420.0, hedge=0.002, pole=1.0 / 700.0, tree=0.02, bush=0.008,
                      crown_dens=0.1, bush_dens=1.2, rough=0.12)


def street_walk(n_edges, nodes, seed, explore=1.0):
    """Seeded random walk over an nodes x nodes street grid from its centre: list of (i, j) nodes, prefix-stable.
    explore > 1 prefers streets not driven yet by that factor (a driver who is going somewhere, not circling the block)."""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    i = j = nodes // 2
    di, dj = 1, 0
    out = [(i, j)]
    seen = set()
    for _ in range(n_edges):
        opts = [(di, dj, 0.5), (-dj, di, 0.25), (dj, -di, 0.25)]  # straight, left, right
        opts = [(a, b, w * (1.0 if frozenset(((i, j), (i + a, j + b))) in seen else explore))
                for a, b, w in opts if 0 <= i + a < nodes and 0 <= j + b < nodes]
        u = rng.random() * sum(w for _, _, w in opts)  # one draw per edge whatever the options: prefix-stable
        for a, b, w in opts:
            u -= w
            if u <= 0:
                break
        di, dj = a, b
        seen.add(frozenset(((i, j), (i + di, j + dj))))
        i, j = i + di, j + dj
        out.append((i, j))
    return out


def _world_init_kitti(self, rng, seed, **kw):
    g = dict(KITTI_DEFAULTS)
    g.update(kw)
    self.kp = g
    tile, block = g["tile"], g["block"]
    self.tile, self.loop_len = tile, float("inf")
    nodes = int(tile // block)               # streets at (k - (nodes-1)/2) * block
    self.nodes = nodes
    self.street_xy = (np.arange(nodes) - (nodes - 1) / 2.0) * block
    self.seed = seed
    # road mask on a 1 m raster (True = road corridor), and its summed-area table for box queries
    n = int(tile)
    c = np.arange(n) + 0.5 - tile / 2
    near = np.min(np.abs(c[:, None] - self.street_xy[None, :]), axis=1) < g["road_half"]
    inside = (np.abs(c) <= self.street_xy[-1] + g["road_half"])
    road = (near[:, None] & inside[None, :]) | (near[None, :] & inside[:, None])
    self._road_sat = np.zeros((n + 1, n + 1), np.int32)
    self._road_sat[1:, 1:] = road.cumsum(0).cumsum(1)
    dist = np.min(np.abs(c[:, None] - self.street_xy[None, :]), axis=1)   # distance to the nearest street line, per coordinate

    def off_road(x0, y0, x1, y1):
        i0 = np.clip(np.floor(x0 + tile / 2).astype(int), 0, n)
        i1 = np.clip(np.ceil(x1 + tile / 2).astype(int), 0, n)
        j0 = np.clip(np.floor(y0 + tile / 2).astype(int), 0, n)
        j1 = np.clip(np.ceil(y1 + tile / 2).astype(int), 0, n)
        S = self._road_sat
        return (S[i1, j1] - S[i0, j1] - S[i1, j0] + S[i0, j0]) == 0

    def street_dist(x, y):
        ix = np.clip((x + tile / 2).astype(int), 0, n - 1)
        iy = np.clip((y + tile / 2).astype(int), 0, n - 1)
        return np.minimum(dist[ix], dist[iy])

    A = tile * tile
    U = lambda k: rng.random(k)
    boxes = []
    # houses: set back from the street
    k = int(A * g["house"])
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    sx, sy, h = 7 + 9 * U(k), 7 + 9 * U(k), 3.5 + 6 * U(k)
    ok = off_road(x - sx / 2 - 4.5, y - sy / 2 - 4.5, x + sx / 2 + 4.5, y + sy / 2 + 4.5)
    boxes.append(np.stack([x - sx / 2, y - sy / 2, 0 * x, x + sx / 2, y + sy / 2, h], 1)[ok])
    hx0, hy0, hx1, hy1 = boxes[0][:, 0], boxes[0][:, 1], boxes[0][:, 3], boxes[0][:, 4]
    # parked cars: at the kerb (within 3 m of the road edge), along the nearer street
    k = int(A * g["car"] * 6)
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    d = street_dist(x, y)
    along_x = dist[np.clip((y + tile / 2).astype(int), 0, n - 1)] <= dist[np.clip((x + tile / 2).astype(int), 0, n - 1)]  # nearest street runs along x
    sx = np.where(along_x, 4.2, 1.8)
    sy = np.where(along_x, 1.8, 4.2)
    ok = (d < g["road_half"] + 3.2) & off_road(x - sx / 2, y - sy / 2, x + sx / 2, y + sy / 2)
    boxes.append(np.stack([x - sx / 2, y - sy / 2, 0 * x, x + sx / 2, y + sy / 2, 1.45 + 0.25 * U(k)], 1)[ok])
    # hedges / fences / walls: thin, 3-14 m long, mostly parallel to the nearer street
    k = int(A * g["hedge"])
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    ln, th, h = 3 + 11 * U(k), 0.25 + 0.6 * U(k), 0.9 + 1.6 * U(k)
    along_x = (dist[np.clip((y + tile / 2).astype(int), 0, n - 1)] <= dist[np.clip((x + tile / 2).astype(int), 0, n - 1)]) ^ (U(k) < 0.3)
    sx, sy = np.where(along_x, ln, th), np.where(along_x, th, ln)
    ok = off_road(x - sx / 2, y - sy / 2, x + sx / 2, y + sy / 2)
    boxes.append(np.stack([x - sx / 2, y - sy / 2, 0 * x, x + sx / 2, y + sy / 2, h], 1)[ok])
    self.boxes = np.concatenate(boxes).astype(np.float32)

    def outside_houses(x, y, m):
        """keep what is not inside a house (coarse: test against houses through a 16 m hash grid)"""
        cell = 16.0
        key = lambda a, b: (np.floor(a / cell).astype(np.int64) + 4096) * 8192 + np.floor(b / cell).astype(np.int64) + 4096
        hk = {}
        for idx in range(len(hx0)):
            for a in range(int(np.floor(hx0[idx] / cell)), int(np.floor(hx1[idx] / cell)) + 1):
                for b in range(int(np.floor(hy0[idx] / cell)), int(np.floor(hy1[idx] / cell)) + 1):
                    hk.setdefault((a + 4096) * 8192 + b + 4096, []).append(idx)
        keep = np.ones(len(x), bool)
        kk = key(x, y)
        for t in range(len(x)):
            for idx in hk.get(int(kk[t]), ()):
                if hx0[idx] - m[t] < x[t] < hx1[idx] + m[t] and hy0[idx] - m[t] < y[t] < hy1[idx] + m[t]:
                    keep[t] = False
                    break
        return keep

    cyls, vols = [], []
    # poles
    k = int(A * g["pole"])
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    r, h = 0.1 + 0.12 * U(k), 4 + 5 * U(k)
    ok = off_road(x - r, y - r, x + r, y + r) & (street_dist(x, y) < g["road_half"] + 2.5)
    cyls.append(np.stack([x, y, r, h], 1)[ok])
    # trees: trunk + crown
    k = int(A * g["tree"])
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    tr, th = 0.12 + 0.25 * U(k), 2.0 + 2.0 * U(k)
    cr, ct = 1.2 + 3.0 * U(k) ** 1.5, 5.0 + 8.0 * U(k)
    ok = off_road(x - tr - 0.6, y - tr - 0.6, x + tr + 0.6, y + tr + 0.6)
    ok &= outside_houses(x, y, tr + 0.5)
    cyls.append(np.stack([x, y, tr, th], 1)[ok])
    vols.append(np.stack([x, y, cr, th - 0.4, np.maximum(ct, th + 1.5), g["crown_dens"] * (0.6 + 0.8 * U(k))], 1)[ok])
    # bushes
    k = int(A * g["bush"])
    x, y = rng.uniform(-tile / 2, tile / 2, k), rng.uniform(-tile / 2, tile / 2, k)
    r, h = 0.5 + 1.3 * U(k) ** 2, 0.8 + 2.0 * U(k)
    ok = off_road(x - r, y - r, x + r, y + r) & outside_houses(x, y, r)
    vols.append(np.stack([x, y, r, 0 * x, h, g["bush_dens"] * (0.5 + U(k))], 1)[ok])
    self.cyls = np.concatenate(cyls).astype(np.float32)
    self.vols = np.concatenate(vols).astype(np.float32)   # x, y, r, z0, z1, extinction [1/m]
    nw = 6
    self.rough = np.stack([np.full(nw, g["rough"]), 2 * np.pi / rng.uniform(2.5, 9.0, nw), rng.uniform(0, 2 * np.pi, nw),
                           2 * np.pi / rng.uniform(2.5, 9.0, nw), rng.uniform(0, 2 * np.pi, nw)], axis=1).astype(np.float32)
    self._path = None


def _world_path(self, n_scans, step=1.0):
    """Pose (x, y, yaw) of scans 0..n_scans-1 of the kitti world's drive (prefix-stable in n_scans)."""
    g = self.kp
    need = n_scans * step + 4 * g["block"]
    if self._path is None or self._path[0][-1] < need:
        n_edges = int(need / g["block"]) + 64
        n_edges = ((n_edges + 255) // 256) * 256
        walk = street_walk(n_edges, self.nodes, self.seed, g["explore"])
        P = np.asarray([(self.street_xy[i], self.street_xy[j]) for i, j in walk])
        rng = np.random.Generator(np.random.PCG64(self.seed + 78))
        lane = g["lane"] + rng.normal(0, 0.25, len(P))              # lateral offset at every node, interpolated along the edge
        f = g["fillet"]
        pts, lat = [P[0]], [lane[0]]
        for k in range(1, len(P) - 1):
            a, b, c = P[k - 1], P[k], P[k + 1]
            u, v = (b - a) / np.linalg.norm(b - a), (c - b) / np.linalg.norm(c - b)
            if abs(u @ v) > 0.99:                                   # straight on
                pts.append(b)
                lat.append(lane[k])
                continue
            s0, s1 = b - f * u, b + f * v                            # quarter circle of radius f
            ctr = s0 + f * v
            a0 = np.arctan2(s0[1] - ctr[1], s0[0] - ctr[0])
            sgn = np.sign(u[0] * v[1] - u[1] * v[0])
            for t in np.linspace(0, 1, 9):
                ang = a0 + sgn * t * np.pi / 2
                pts.append(ctr + f * np.array([np.cos(ang), np.sin(ang)]))
                lat.append(lane[k])
        pts, lat = np.asarray(pts), np.asarray(lat)
        seg = np.hypot(np.diff(pts[:, 0]), np.diff(pts[:, 1]))
        s = np.concatenate([[0], np.cumsum(seg)])
        self._path = (s, pts, lat)
    s, pts, lat = self._path
    d = np.arange(n_scans) * step
    x, y = np.interp(d, s, pts[:, 0]), np.interp(d, s, pts[:, 1])
    x2, y2 = np.interp(d + 0.5, s, pts[:, 0]), np.interp(d + 0.5, s, pts[:, 1])
    yaw = np.arctan2(y2 - y, x2 - x)
    lt = np.interp(d, s, lat)
    x, y = x + lt * np.sin(yaw), y - lt * np.cos(yaw)               # to the right of the driving direction
    rng_yaw = np.random.Generator(np.random.PCG64(self.seed + 1000))
    yaw = yaw + rng_yaw.normal(0, 0.01, n_scans)
    return x, y, yaw


def _world_ground_rough(self, x, y):
    is_t = torch.is_tensor(x)
    sin = torch.sin if is_t else np.sin
    z = x * 0
    for a, fx, px_, fy, py_ in self.rough.tolist():
        z = z + a * sin(fx * x + px_) * sin(fy * y + py_)
    return z


World.ground_rough = _world_ground_rough
World.rough = None
World._init_kitti = _world_init_kitti
World.path = _world_path
World.vols = None
World.kitti = False
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
def _vol_hash(ray, vid, seed32):
    """32-bit integer hash of (ray, object, scan) in int64 torch arithmetic; tools/synth_hip computes the same in uint32."""
    M = 0xFFFFFFFF
    h = ((ray * 0x9E3779B1) & M) ^ (((vid + 0x7F4A7C15) * 0x85EBCA77) & M) ^ (((seed32 + 0x165667B1) & M) * 0xC2B2AE3D & M)
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & M
    return h ^ (h >> 15)


@torch.no_grad()
def cast_scan(world, pose, beams=64, azim=1875, device="cpu", noise_sigma=0.02, gen=None, chunk=32768, elev_deg=None, scan_seed=0):
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
    vols = vid = None
    if getattr(world, "vols", None) is not None:
        vv = world.vols
        keepv = (np.abs(vv[:, 0] - px) < MAX_RANGE + vv[:, 2]) & (np.abs(vv[:, 1] - py) < MAX_RANGE + vv[:, 2])
        vols = torch.from_numpy(vv[keepv]).to(dev)
        vid = torch.from_numpy(np.nonzero(keepv)[0]).to(dev)
    seed32 = int(scan_seed) & 0xFFFFFFFF ^ (int(scan_seed) >> 32) & 0xFFFFFFFF
    N = dw.shape[0]
    t_best = torch.full((N,), float("inf"), device=dev)
    for i0 in range(0, N, chunk):
        d = dw[i0:i0 + chunk]
        tb = torch.full((d.shape[0],), float("inf"), device=dev)
        if not relief:
            # ground plane z = 0
            tg = torch.where(d[:, 2] < -1e-6, -SENSOR_H / d[:, 2], torch.full_like(d[:, 2], float("inf")))
            if getattr(world, "rough", None) is not None:
                # rough ground (grass, kerbs: a few cm): the height under the flat-plane hit point shifts the hit along the ray,
                # which spreads the far ground rings over neighbouring cells as real ground does
                tq = torch.where(torch.isfinite(tg), tg, torch.zeros_like(tg)).clamp(max=2 * MAX_RANGE)
                dg = world.ground_rough(o[0] + tq * d[:, 0], o[1] + tq * d[:, 1])
                tg = torch.where(d[:, 2] < -1e-6, -(SENSOR_H - dg) / d[:, 2], tg)
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
        if vols is not None and vols.shape[0]:
            # porous cylinders (crowns, bushes): chord of the ray inside [r] x [z0, z1]; the ray ends in it with probability
            # 1 - exp(-extinction * chord), at a uniform depth along the chord; both draws from one hash of (scan, ray, object)
            for v0 in range(0, vols.shape[0], 512):
                V = vols[v0:v0 + 512]
                qx = o[0] - V[None, :, 0]
                qy = o[1] - V[None, :, 1]
                dx, dy, dz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
                a = dx * dx + dy * dy
                b = 2 * (qx * dx + qy * dy)
                cc = qx * qx + qy * qy - V[None, :, 2] ** 2
                disc = b * b - 4 * a * cc
                sq = torch.sqrt(torch.clamp(disc, min=0))
                ia = 1.0 / (2 * a + 1e-12)
                t_in, t_out = (-b - sq) * ia, (-b + sq) * ia
                iz = 1.0 / torch.where(dz.abs() < 1e-9, torch.full_like(dz, 1e-9), dz)
                tz0, tz1 = (V[None, :, 3] - o[2]) * iz, (V[None, :, 4] - o[2]) * iz
                lo = torch.maximum(torch.maximum(t_in, torch.minimum(tz0, tz1)), torch.zeros_like(t_in))
                hi = torch.minimum(t_out, torch.maximum(tz0, tz1))
                ln = hi - lo
                ridx = torch.arange(i0, i0 + d.shape[0], device=dev, dtype=torch.int64)[:, None]
                h = _vol_hash(ridx, vid[None, v0:v0 + 512], seed32)
                u1 = (h & 0xFFFF).to(torch.float32) * (1.0 / 65536.0)
                u2 = ((h >> 16) & 0xFFFF).to(torch.float32) * (1.0 / 65536.0)
                stop = (disc > 0) & (ln > 0) & (u1 < 1.0 - torch.exp(-V[None, :, 5] * ln))
                tv = torch.where(stop, lo + u2 * ln, torch.full_like(lo, float("inf")))
                tb = torch.minimum(tb, tv.amin(dim=1))
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
    if getattr(world, "kitti", False):
        x, y, yaw = world.path(total, step=step)
    else:
        x, y, yaw = trajectory(total, step=step, loop_len=world.loop_len, tile=world.tile)
    if torch.device(device).type == "cuda" and len(idx) and _hip_caster() is not None:
        # fused HIP ray caster (tools/synth_hip): same geometry, hash-based noise -- ~50x faster than the torch ops below
        xyzi = _cast_scans_hip(world, x[idx], y[idx], yaw[idx], seed * 1000003 + idx, beams, azim, torch.device(device), noise_sigma, elev_deg)
        return xyzi, np.stack([x[idx], y[idx], yaw[idx]], axis=1), idx.astype(np.float64) / 10.0
    gen = torch.Generator(device=torch.device(device))
    out = []
    for i in idx.tolist():
        gen.manual_seed(seed * 1000003 + i)
        out.append(cast_scan(world, (x[i], y[i], yaw[i]), beams, azim, device, noise_sigma, gen, elev_deg=elev_deg,
                             scan_seed=seed * 1000003 + i))
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
        lib.sc_cast_scans.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
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
    vv = getattr(world, "vols", None)
    rough = getattr(world, "rough", None)
    mode = 1 if relief else (2 if rough is not None else 0)
    if mode == 2:
        wave = np.ascontiguousarray(rough, np.float32).reshape(-1)
    if vv is not None:
        vv8 = np.zeros((len(vv), 8), np.float32)
        vv8[:, :6] = vv
        vv8[:, 6] = np.arange(len(vv), dtype=np.int32).view(np.float32)
    lib = _hip_caster()
    stream = torch.cuda.current_stream(dev).cuda_stream
    # culling per scan: objects sorted by x once per world, so a scan looks at the x-slab around it only
    cache = getattr(world, "_cull_cache", None)
    if cache is None:
        cyl6 = np.concatenate([cy, cbase[:, None], cid_bits[:, None]], axis=1) if len(cy) else np.zeros((0, 6), np.float32)
        ob, oc = np.argsort(bx[:, 0], kind="stable"), np.argsort(cy[:, 0], kind="stable")
        cache = {"bx": bx[ob], "bw": float((bx[:, 3] - bx[:, 0]).max()) if len(bx) else 0.0, "cy": cyl6[oc]}
        if vv is not None:
            ov = np.argsort(vv[:, 0], kind="stable")
            cache["vv"] = vv8[ov]
            cache["vr"] = float(vv[:, 2].max()) if len(vv) else 0.0
        world._cull_cache = cache
    sbx, scy = cache["bx"], cache["cy"]
    for c0 in range(0, n, chunk):
        c1 = min(c0 + chunk, n)
        boxes, cyls, boff, coff, poses = [], [], [0], [0], []
        vols, voff = [], [0]
        for i in range(c0, c1):
            px, py = float(xs[i]), float(ys[i])
            a, b = np.searchsorted(sbx[:, 0], [px - MAX_RANGE - cache["bw"], px + MAX_RANGE])
            sl = sbx[a:b]
            keep = (sl[:, 3] > px - MAX_RANGE) & (sl[:, 0] < px + MAX_RANGE) & (sl[:, 4] > py - MAX_RANGE) & (sl[:, 1] < py + MAX_RANGE)
            boxes.append(sl[keep])
            a, b = np.searchsorted(scy[:, 0], [px - MAX_RANGE, px + MAX_RANGE])
            sc = scy[a:b]
            keepc = (np.abs(sc[:, 0] - px) < MAX_RANGE) & (np.abs(sc[:, 1] - py) < MAX_RANGE)
            cyls.append(sc[keepc])
            boff.append(boff[-1] + int(keep.sum()))
            coff.append(coff[-1] + int(keepc.sum()))
            if vv is not None:
                svv = cache["vv"]
                a, b = np.searchsorted(svv[:, 0], [px - MAX_RANGE - cache["vr"], px + MAX_RANGE + cache["vr"]])
                sv = svv[a:b]
                keepv = (np.abs(sv[:, 0] - px) < MAX_RANGE + sv[:, 2]) & (np.abs(sv[:, 1] - py) < MAX_RANGE + sv[:, 2])
                vols.append(sv[keepv])
                voff.append(voff[-1] + int(keepv.sum()))
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
        t_vols = t_voff = None
        if vv is not None:
            t_vols = torch.from_numpy(np.ascontiguousarray(np.concatenate(vols) if voff[-1] else np.zeros((1, 8)), np.float32)).to(dev)
            t_voff = torch.tensor(voff, dtype=torch.int32, device=dev)
        rc = lib.sc_cast_scans(dirs.data_ptr(), N, c1 - c0, t_pose.data_ptr(), t_boxes.data_ptr(), t_boff.data_ptr(), t_cyls.data_ptr(),
                               t_coff.data_ptr(), t_seed.data_ptr(), mode, wave.ctypes.data if mode else None,
                               float(noise_sigma), out[c0:c1].data_ptr(), stream,
                               t_vols.data_ptr() if vv is not None else None, t_voff.data_ptr() if vv is not None else None)
        if rc != 0:
            raise RuntimeError("sc_cast_scans failed")
        torch.cuda.current_stream(dev).synchronize()  # the staged lists go out of scope
    return out
