"""Synthetic initial states for the BASELINE.json configurations.

``terrain_grid`` restates what the reference's setup pass produces for a new simulation
(shaders/fragment/setupShader.frag:36-92: multi-octave value noise terrain, sea below, land with soil
moisture / vegetation / snow above, initial sounding with a 2 K / 20 K dew-point depression), with the
terrain snapped to even x / even y steps (SwiftShader-safe). ``dry_grid`` is the
config-2 state: inert floor, dry air with small temperature noise.
``init_rain_drops`` restates initRainDrops() (app.js:4901-4913) with a seeded generator.
"""
from __future__ import annotations

import numpy as np

from . import params


def _frac(x):
    return x - np.floor(x)


def _rand(n):
    # setupShader.frag:26  fract(sin(n) * 43758.5453123)
    return _frac(np.sin(n) * 43758.5453123)


def _noise(p):
    # setupShader.frag:28-33
    fl = np.floor(p)
    fc = p - fl
    return _rand(fl) * (1.0 - fc) + _rand(fl + 1.0) * fc - 0.5


def terrain_height(X: int, seed: float = 0.5, height_mult: float = 0.3) -> np.ndarray:
    """Normalised terrain height per column (setupShader.frag:44-61)."""
    fx = np.arange(X, dtype=np.float64) + 0.5
    if height_mult < 0.05:
        return np.zeros(X)
    if height_mult < 0.10:
        return np.full(X, 0.005)
    var = fx * 0.001
    h = np.zeros(X)
    i = 2.0
    while i < 1000.0:
        h += _noise(var * i + _rand(seed + i) * 10.0) * 0.5 / i
        i *= 1.5
    return h * height_mult


def terrain_columns(X: int, Y: int, gui=None, seed: float = 0.5, height_mult: float = 0.3, snap: int = 2, cols=None, cloud_deck: bool = False):
    """The 1-D part of the setup pass (setupShader.frag:36-92): per column the number of wall rows, sea / land, the
    vegetation noise term and the snow height; per row the air temperature, total and cloud water of the initial
    sounding. ``terrain_grid`` expands these descriptors on the host, ``Handle.setup_columns`` (wx_setup_columns) on the
    device. ``cols=(start, count)`` selects the columns ``(start + i) mod X`` (a slab with its ghost columns)."""
    gui = params.merge_settings(None) if gui is None else gui
    sim_h = float(gui["simHeight"])
    dry_lapse = sim_h * float(gui["dryLapseRate"]) / 1000.0
    T0 = params.initial_temperature_profile(Y, sim_h, dry_lapse)
    h = terrain_height(X, seed, height_mult)
    if snap > 1:  # snap in x: constant over `snap` columns
        h = np.repeat(h[::snap], snap)[:X]
    texY = 1.0 / Y
    # rows that are wall: texCoord.y < texelSize.y or texCoord.y < height  -> row index < max(1, height*Y - 0.5)
    nrows = np.maximum(1, np.ceil(h * Y - 0.5).astype(np.int64))
    if snap > 1:
        nrows = ((nrows + snap - 1) // snap) * snap  # even thickness
    nrows = np.minimum(nrows, Y - 8)
    gcol = np.arange(X)
    if cols is not None:
        gcol = (cols[0] + np.arange(cols[1])) % X
        h, nrows = h[gcol], nrows[gcol]
    fx = gcol.astype(np.float64) + 0.5
    return {
        "wall_rows": nrows.astype(np.int32), "sea": (h < texY).astype(np.uint8),
        "veg_noise": (_noise(fx * 0.01 + _rand(seed) * 10.0) * 150.0).astype(np.float64),
        "snow": np.clip((h * sim_h - 2000.0) * 100.0 / 3000.0, 0.0, 100.0).astype(np.float32),
        **sounding_rows(Y, gui, cloud_deck),
    }


def sounding_rows(Y: int, gui=None, cloud_deck: bool = False):
    """Per row: air temperature, total and cloud water of the initial sounding (setupShader.frag:78-89) -- the part of a new
    simulation that stays on the host when the terrain is generated on the device (``Handle.setup_terrain``)."""
    gui = params.merge_settings(None) if gui is None else gui
    sim_h = float(gui["simHeight"])
    dry_lapse = sim_h * float(gui["dryLapseRate"]) / 1000.0
    T0 = params.initial_temperature_profile(Y, sim_h, dry_lapse)
    yy = np.arange(Y)
    tcy = (yy + 0.5) / Y
    T_air = T0[:Y].astype(np.float64)
    realT = T_air - tcy * dry_lapse
    dew = np.where(tcy < 0.20, realT - 2.0, realT - 20.0)
    tot = (dew / 250.0) ** 17
    cloud = np.maximum(tot - (realT / 250.0) ** 17, 0.0)
    T_air, tot, cloud = T_air.astype(np.float32), tot.astype(np.float32), cloud.astype(np.float32)
    if cloud_deck:  # a cloud layer so that droplets spawn, grow and fall during a benchmark run (see add_cloud_deck)
        deck = (yy > Y // 4) & (yy < Y // 2)
        cloud = np.where(deck, np.float32(1.5), cloud).astype(np.float32)
        tot = np.where(deck, tot + np.float32(1.5), tot).astype(np.float32)
    return {"T_air": T_air, "total_water": tot, "cloud_water": cloud}


def terrain_grid(X: int, Y: int, gui=None, seed: float = 0.5, height_mult: float = 0.3, snap: int = 2, cols=None):
    """(base, water, wall) of a new simulation (setupShader.frag:36-92), terrain snapped to ``snap`` cells.

    ``cols=(start, count)`` returns only the columns ``(start + i) mod X`` (a slab with its ghost columns) of the
    X-wide domain; every quantity is a function of the global column, so slabs tile the whole-domain result."""
    d = terrain_columns(X, Y, gui, seed, height_mult, snap, cols)
    nrows, sea_col = d["wall_rows"].astype(np.int64), d["sea"].astype(bool)
    X = len(nrows)
    yy = np.arange(Y)[:, None]
    is_wall = yy < nrows[None, :]
    is_sea = sea_col[None, :] & is_wall
    is_land = is_wall & ~is_sea

    base = np.zeros((Y, X, 4), np.float32)
    water = np.zeros((Y, X, 4), np.float32)
    wall = np.zeros((Y, X, 4), np.int8)

    air = ~is_wall
    base[..., 3] = np.where(air, d["T_air"][:, None], np.float32(0.0))
    water[..., 0] = np.where(air, d["total_water"][:, None], np.float32(0.0))
    water[..., 1] = np.where(air, d["cloud_water"][:, None], np.float32(0.0))

    # walls (setupShader.frag:63-77)
    base[..., 3] = np.where(is_sea, np.float32(25.0) + np.float32(273.15), base[..., 3])
    base[..., 3] = np.where(is_land, 1000.0, base[..., 3])
    water[..., 0] = np.where(is_sea, 1002.0, water[..., 0])
    water[..., 0] = np.where(is_land, 1001.0, water[..., 0])
    water[..., 2] = np.where(is_land, 25.0, water[..., 2])
    water[..., 2] = np.where(is_sea, 100.0, water[..., 2])
    veg = 110.0 - (yy + 0.5) * 2.0 + d["veg_noise"][None, :]
    wall[..., 3] = np.where(is_land, np.clip(np.trunc(veg), 0, 127), 0).astype(np.int8)
    water[..., 3] = np.where(is_land, d["snow"][None, :], water[..., 3])

    wall[..., 0] = np.where(sea_col[None, :], 2, 1).astype(np.int8)  # type, extended upward like the boundary pass does
    vdist = yy - nrows[None, :] + 1  # 1 for the first air row, 0 for the top wall row, negative below
    wall[..., 2] = np.clip(vdist, -127, 127).astype(np.int8)
    wall[..., 1] = np.where(is_wall, 0, np.clip(vdist, 1, 127)).astype(np.int8)
    return base, water, wall


_DRY_BLOCK = 256  # columns per random-number block of dry_grid


def dry_grid(X: int, Y: int, gui=None, seed: int = 1234, noise_K: float = 0.05, cols=None, flow_sigma: float = 0.0):
    """BASELINE config 2: inert floor row, dry air, T = initial_T[y] + N(0, noise_K); still (``flow_sigma`` 0) or with a
    developed velocity field v ~ N(0, flow_sigma) cells / iteration (what the parity tests seed: back-traces that leave the lane's
    own cell). The random numbers are drawn per block of 256 GLOBAL columns (Philox keyed by seed and block), so
    ``cols=(start, count)`` -- the columns ``(start + i) mod X`` of the X-wide domain, a slab with its ghost columns -- tiles the
    whole-domain result without generating it."""
    gui = params.merge_settings(None) if gui is None else gui
    sim_h = float(gui["simHeight"])
    dry_lapse = sim_h * float(gui["dryLapseRate"]) / 1000.0
    T0 = params.initial_temperature_profile(Y, sim_h, dry_lapse)
    gcol = np.arange(X) if cols is None else (cols[0] + np.arange(cols[1])) % X
    n = len(gcol)
    base = np.zeros((Y, n, 4), np.float32)
    water = np.zeros((Y, n, 4), np.float32)
    wall = np.zeros((Y, n, 4), np.int8)
    base[..., 3] = T0[:Y][:, None]
    blk = gcol // _DRY_BLOCK
    for b in np.unique(blk):
        rng = np.random.Generator(np.random.Philox(key=[seed, int(b)]))
        w = min(_DRY_BLOCK, X - int(b) * _DRY_BLOCK)
        noise = [rng.standard_normal((Y, w), dtype=np.float32) for _ in range(3 if flow_sigma else 1)]  # (T first: same with or without flow)
        sel = np.nonzero(blk == b)[0]
        loc = gcol[sel] - int(b) * _DRY_BLOCK
        if len(sel) == sel[-1] - sel[0] + 1 and len(loc) == loc[-1] - loc[0] + 1:  # (the usual case: one contiguous run -> slices)
            sel, loc = slice(sel[0], sel[-1] + 1), slice(loc[0], loc[-1] + 1)
        base[:, sel, 3] += noise[0][:, loc] * np.float32(noise_K)
        if flow_sigma:
            base[1:, sel, 0] += noise[1][1:, loc] * np.float32(flow_sigma)
            base[1:, sel, 1] += noise[2][1:, loc] * np.float32(flow_sigma)
    yy = np.arange(Y)[:, None]
    wall[..., 1] = np.clip(yy, 0, 127)
    wall[..., 2] = np.clip(yy, -127, 127)
    base[0, :, 3] = 1000.0
    water[0, :, 0] = 1001.0
    return base, water, wall


def add_vortices(base: np.ndarray, wall: np.ndarray, centers, radius: float, peak: float) -> int:
    """Adds compact vortices to base[..., 0:2] in place: the discrete curl of Gaussian stream-function blobs psi = A exp(-r^2 / 2 s^2) on
    the staggered grid (vx = psi(x, y) - psi(x, y-1), vy = -(psi(x, y) - psi(x-1, y)): divergence-free for pressureShader.frag:16-43, so the
    pressure pass does not radiate them away), periodic in x, tangential speed `peak` cells / iteration at r = s = `radius`. Nothing clamps
    the velocity: the reference has no clamp either (advectionShader.frag:85-99), and |v| >= 0.9 is what the kernels' exact paths are for.
    ``centers``: (x, y) pairs, or (x, y, sign). Returns the number of air cells with a velocity component of 0.9 or more."""
    Y, X = base.shape[:2]
    A = float(peak) * float(radius) * float(np.exp(0.5))
    R = int(np.ceil(6 * radius)) + 2
    air = wall[..., 1] != 0
    for c in centers:
        cx, cy, sg = float(c[0]), float(c[1]), (float(c[2]) if len(c) > 2 else 1.0)
        ys = np.arange(max(1, int(cy) - R), min(Y, int(cy) + R + 1))
        xs = np.arange(int(cx) - R, int(cx) + R + 1)

        def psi(xx, yy):
            r2 = (xx[None, :] + 0.5 - cx) ** 2 + (yy[:, None] + 0.5 - cy) ** 2
            return sg * A * np.exp(-r2 / (2.0 * radius * radius))

        xf, yf = xs.astype(np.float64), ys.astype(np.float64)
        p11, p10, p01 = psi(xf, yf), psi(xf, yf - 1.0), psi(xf - 1.0, yf)
        vx, vy = (p11 - p10).astype(np.float32), (-(p11 - p01)).astype(np.float32)
        xw = xs % X
        m = air[np.ix_(ys, xw)]
        base[np.ix_(ys, xw, [0])] += np.where(m, vx, 0)[..., None]
        base[np.ix_(ys, xw, [1])] += np.where(m, vy, 0)[..., None]
    return int(((np.maximum(np.abs(base[..., 0]), np.abs(base[..., 1])) >= 0.9) & air).sum())


def init_rain_drops(n: int, seed: int = 7) -> np.ndarray:
    """initRainDrops() (app.js:4901-4913): inactive droplets whose fields are random seeds."""
    rng = np.random.Generator(np.random.Philox(seed))
    d = rng.random((n, 5)).astype(np.float32)
    d[:, 2] = -10.0 + d[:, 2]
    return d


def add_cloud_deck(water: np.ndarray, wall: np.ndarray) -> None:
    """In place: a cloud layer between Y/4 and Y/2 (cloud water 1.5, total water raised accordingly) so that droplets
    spawn, grow and fall during a benchmark run (BASELINE configs[4]). Works on whole grids and on column slabs."""
    Y = wall.shape[0]
    yy = np.arange(Y)[:, None]
    deck = (wall[..., 1] != 0) & (yy > Y // 4) & (yy < Y // 2)
    water[..., 1] = np.where(deck, 1.5, water[..., 1]).astype(np.float32)
    water[..., 0] = np.where(deck, water[..., 0] + 1.5, water[..., 0]).astype(np.float32)


def _hash_u32(x: np.ndarray) -> np.ndarray:
    # common.glsl:103-111 on uint32 arrays (wrap-around arithmetic)
    x = x.astype(np.uint32)
    x = x + (x << np.uint32(10))
    x = x ^ (x >> np.uint32(6))
    x = x + (x << np.uint32(3))
    x = x ^ (x >> np.uint32(11))
    x = x + (x << np.uint32(15))
    return x


def init_rain_drops_hashed(n: int, seed: int = 1) -> np.ndarray:
    """The pool ``Handle.init_droplets(seed)`` (wx_init_droplets) generates on the device, restated: field c of droplet i = 24 bits
    of hash(seed + hash(5 i + c)); (r, r, -10 + r, r, r) as initRainDrops() lays them out (app.js:4901-4913)."""
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint32)[:, None] * np.uint32(5) + np.arange(5, dtype=np.uint32)[None, :]).astype(np.uint32)
        h = _hash_u32(np.uint32(seed & 0xFFFFFFFF) + _hash_u32(idx))
    r = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    r[:, 2] = np.float32(-10.0) + r[:, 2]
    return r
