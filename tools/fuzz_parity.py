#!/usr/bin/env python3
"""Differential fuzzer: the HIP path (through the C ABI, default kernel set and options a host can pick) against the CPU oracle, BIT FOR
BIT, on randomly drawn cases -- grid size (tiny, ragged, a strip / a tile / a band border away from the launch shapes' corners), terrain,
flow speed (up to several cells per iteration: the exact paths), humidity / cloud / smoke / snow, settings (every slider inside the range
the reference's GUI offers), pass mask (all passes / the dry stencil), brush tool and airplane inputs, droplets in deterministic splat
order, the way a host cuts its iterations into steps, dry pairs on / off, row bands, waterTexture_0 on demand / stored. --mode group: the same
scenes cut into 2 .. 8 column slabs on this one GPU (the library's own halo exchange, random halo widths, overlapped / split / in-order
protocol, droplet pool in exact mode) against the undecomposed handle, bit for bit.

    python tools/fuzz_parity.py [--seed S] [--cases N] [--seconds T] [--max-cells C]

Every case prints one line; a mismatch prints the case's recipe (the seed reproduces it: --seed S --only K) and the run exits 1 at the
end. The oracle is the checker (test infrastructure); nothing here is a product path."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import wxpkg  # noqa: E402

GRID_FIELDS = ["BASE_CUR", "BASE_DISP", "WATER_0", "WATER_CUR", "WALL_CUR", "WALL_DISP", "LIGHT_0", "LIGHT_1"]
# sliders of the reference's GUI (app.js:3402-3560: min / max of the controls the simulation reads)
SLIDERS = {"vorticity": (0.0, 0.010), "dragMultiplier": (0.0, 1.0), "wind": (-1.0, 1.0), "globalDrying": (0.0, 0.001), "globalHeating": (-0.002, 0.002),
           "sunIntensity": (0.0, 2.0), "waterTemperature": (0.0, 40.0), "landEvaporation": (0.0, 0.0002), "waterEvaporation": (0.0, 0.0004),
           "evapHeat": (0.0, 5.0), "meltingHeat": (0.0, 5.0), "condensationRate": (0.0, 0.01), "waterWeight": (0.0, 2.0),
           "greenhouseGases": (0.0, 0.01), "waterGreenHouseEffect": (0.0, 0.01), "IR_rate": (0.0, 10.0), "soundingForcing": (0.0, 0.001)}


def draw_case(rng, max_cells, big=False):
    c = {}
    kind = rng.choice(["tiny", "small", "small", "mid", "wide", "tall"])
    if big:  # (--big: the launch shapes of grids that fill the chip -- row bands per XCD, several rounds of segments, tail segments)
        X, Y = int(rng.integers(1000, 9000)), int(rng.integers(512, 2100))
    elif kind == "tiny":
        X, Y = int(rng.integers(2, 70)), int(rng.integers(4, 40))
    elif kind == "small":
        X, Y = int(rng.integers(40, 400)), int(rng.integers(12, 200))
    elif kind == "mid":
        X, Y = int(rng.integers(300, 1500)), int(rng.integers(100, 700))
    elif kind == "wide":
        X, Y = int(rng.integers(1500, 6000)), int(rng.integers(12, 120))
    else:
        X, Y = int(rng.integers(20, 200)), int(rng.integers(500, 1500))
    if rng.random() < 0.3:  # borders of strips (56 output columns), pair strips (48), tiles (64 x 16)
        X = max(2, int(rng.choice([56, 48, 64, 112, 96, 128, 168, 448])) * int(rng.integers(1, 4)) + int(rng.integers(-1, 2)))
    while X * Y > max_cells:
        X, Y = max(2, X // 2), max(4, Y * 3 // 4)
    c["X"], c["Y"] = X, Y
    c["dry"] = bool(rng.random() < 0.35)
    c["terrain"] = bool(Y >= 12 and rng.random() < (0.4 if c["dry"] else 0.9))
    c["tseed"], c["tmult"] = float(rng.random()), float(rng.uniform(0.05, 0.6))
    c["sigma"] = float(rng.choice([0.0, 0.05, 0.2, 0.35, 0.6, 1.2]) if c["dry"] else rng.choice([0.0, 0.05, 0.1, 0.2, 0.3, 0.45]))
    c["vortices"] = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
    c["moist"] = bool(rng.random() < 0.6)
    c["cloud"], c["smoke"], c["snow"] = bool(rng.random() < 0.4), bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    c["sliders"] = {k: float(rng.uniform(*SLIDERS[k])) for k in SLIDERS if rng.random() < 0.35}
    c["sun"] = float(rng.uniform(-30.0, 210.0))
    c["wrap"] = bool(rng.random() < 0.85)
    c["quad_scale"] = int(rng.random() < 0.2)
    c["iter0"] = int(rng.choice([0, 1, 90, 599, 12345]))
    c["steps"] = [int(v) for v in rng.integers(1, 12, size=int(rng.integers(1, 4)))]
    c["brush"] = None
    if rng.random() < 0.3:
        c["brush"] = {"type": int(rng.integers(0, 24)), "values": [float(rng.random()), float(rng.random()), float(rng.uniform(-1, 1)), float(rng.uniform(1, 30))],
                      "move": [float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.02, 0.02))]}
    c["airplane"] = [float(rng.random()), float(rng.random()), float(rng.random()), float(rng.choice([-1.0, 0.0, 1.0]))] if rng.random() < 0.15 else None
    c["drops"] = int(rng.integers(16, 3000)) if (not c["dry"] and Y >= 24 and rng.random() < 0.3) else 0
    c["pairs"] = int(rng.random() < 0.7)
    c["bands"] = int(rng.choice([0, 1, 1, 2]))
    c["kernel_set"] = int(rng.random() < 0.85)  # 1 = the row-marching kernels (default), 0 = one kernel per reference pass
    c["dry_kernel"] = int(rng.random() < 0.8)  # the dry stencil: row-marching (default) / LDS-tiled
    c["water0_on_demand"] = int(rng.random() < 0.7)
    c["data_seed"] = int(rng.integers(0, 2**31))
    c["brush_toggle"] = bool(rng.random() < 0.5)  # the brush is held down in every other step only (a host's mouse-up / mouse-down between frames)
    c["pieces"] = bool(rng.random() < 0.3)  # steps cut into two pieces, the first with WX_OVERLAP_MORE_TO_COME
    c["reupload"] = bool(rng.random() < 0.2)  # after the first step the current state is uploaded again (a host that edits the state: new ping-pong copies, on slabs a fresh exchange period and |vx| scan)
    c["subrect"] = bool(rng.random() < 0.3)  # also read a random sub-rectangle of every field (wx_read_rect's x / y / w / h)
    return c


def build_case(pkg, c):
    S, P = pkg.synth, pkg.params
    X, Y = c["X"], c["Y"]
    rng = np.random.default_rng(c["data_seed"])
    if c["terrain"]:
        base, water, wall = S.terrain_grid(X, Y, seed=c["tseed"], height_mult=c["tmult"])
    else:
        base, water, wall = S.dry_grid(X, Y)
    air = wall[..., 1] != 0
    if c["sigma"] > 0:
        for ch in (0, 1):
            base[..., ch] += np.where(air, rng.normal(0, c["sigma"], (Y, X)), 0).astype(np.float32)
    if c["vortices"]:
        cs = [(float(rng.uniform(0, X)), float(rng.uniform(Y * 0.2, Y * 0.9))) for _ in range(c["vortices"])]
        S.add_vortices(base, wall, cs, radius=float(rng.uniform(3, 12)), peak=float(rng.uniform(0.8, 2.5)))
    base[..., 2] += np.where(air, rng.normal(0, 1e-3, (Y, X)), 0).astype(np.float32)
    base[..., 3] += np.where(air, rng.normal(0, 0.3, (Y, X)), 0).astype(np.float32)
    if c["moist"]:
        water[..., 0] = np.where(air, water[..., 0] * (1.0 + 0.5 * rng.random((Y, X))) + rng.random((Y, X)) * (2.0 if not c["terrain"] else 0.0), water[..., 0]).astype(np.float32)
    if c["cloud"]:
        blob = air & (rng.random((Y, X)) < 0.2)
        water[..., 1] += np.where(blob, rng.random((Y, X)) * 2.0, 0).astype(np.float32)
        water[..., 0] += np.where(blob, water[..., 1], 0).astype(np.float32)
    if c["smoke"]:
        water[..., 3] += np.where(air & (rng.random((Y, X)) < 0.1), rng.random((Y, X)) * 3.0, 0).astype(np.float32)
    if c["snow"]:
        water[..., 2] += np.where(air & (rng.random((Y, X)) < 0.1), rng.random((Y, X)) * 0.5, 0).astype(np.float32)
    gui = P.merge_settings(None)
    gui.update(c["sliders"])
    gui["sunAngle"] = c["sun"]
    gui["wrapHorizontally"] = c["wrap"]
    u = P.uniforms_from_gui(gui, Y, quad_scale=c["quad_scale"], pass_mask=P.PASS_DRY if c["dry"] else P.PASS_ALL)
    u["enablePrecipitation"] = 1 if c["drops"] else 0
    if c["brush"]:
        u["userInputType"] = c["brush"]["type"]
        u["userInputValues"] = tuple(c["brush"]["values"])
        u["userInputMove"] = tuple(c["brush"]["move"])
    if c["airplane"]:
        u["airplaneValues"] = tuple(c["airplane"])
    drops = None
    if c["drops"]:
        drops = S.init_rain_drops(c["drops"], seed=c["data_seed"] % 1000)
        u["splat_order"] = 1
        u["spawnChanceMult"] = 0.01
    return base, water, wall, u, drops


def run_case(pkg, E, wx_oracle, c):
    """One case on its own."""
    g = case_steps(pkg, E, wx_oracle, c)
    while True:
        try:
            next(g)
        except StopIteration as e:
            return e.value


def run_interleaved(pkg, E, wx_oracle, cases):
    """Several handles alive at once, their steps in turn (state that a handle shares with the process -- work lists, hint words, scratch
    sized by another handle -- would show here). Returns the (bad, info) of every case."""
    gens = [case_steps(pkg, E, wx_oracle, c) for c in cases]
    out = [None] * len(gens)
    while any(o is None for o in out):
        for i, g in enumerate(gens):
            if out[i] is None:
                try:
                    next(g)
                except StopIteration as e:
                    out[i] = e.value
    return out


def case_steps(pkg, E, wx_oracle, c):
    """Generator: yields after every step of the case; its return value is (mismatches, info)."""
    X, Y = c["X"], c["Y"]
    base, water, wall, u, drops = build_case(pkg, c)
    nd = 0 if drops is None else len(drops)
    h = E.Handle(X, Y, nd)
    o = wx_oracle.OracleSim(X, Y, nd)
    bad = []
    try:
        h.upload(base, water, wall, drops)
        o.upload(base, water, wall, drops)
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
        o.set_params(u)
        h.iter = c["iter0"]
        o.iter = c["iter0"]
        h.set_option(h.OPT_DRY_PAIRS, c["pairs"])
        h.set_option(h.OPT_ROW_BANDS, c["bands"])
        h.set_option(h.OPT_KERNEL_SET, c["kernel_set"])
        h.set_option(h.OPT_DRY_KERNEL, c["dry_kernel"])
        h.set_option(h.OPT_WATER0_ON_DEMAND, c["water0_on_demand"])
        if nd:
            h.set_option(h.OPT_SPLAT_ORDER, 1)
        rng = np.random.default_rng(c["data_seed"] ^ 0x5EED)
        for k_step, n in enumerate(c["steps"]):
            if c.get("brush_toggle") and c["brush"] and k_step > 0:  # mouse up / down between two steps: new parameters on both sides
                u2 = dict(u, userInputType=(c["brush"]["type"] if k_step % 2 == 0 else -1))
                h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u2), u["initial_T"])
                o.set_params(u2)
            if c.get("reupload") and k_step == 1 and not nd:
                st = [o.field(f) for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR")]
                h.upload(*st)
                o.upload(*st)
            pieces = [n]
            if c.get("pieces") and n > 1:  # the step cut into pieces whose all but the last skip the display-side stores (WX_OVERLAP_MORE_TO_COME)
                k1 = int(rng.integers(1, n))
                pieces = [k1, n - k1]
            for i_p, k_p in enumerate(pieces):
                h.step(k_p, 4 if i_p + 1 < len(pieces) else 0)
            o.step(n)
            yield
            if not c["dry"] and not np.array_equal(h.read_rect("CURL"), o.field("CURL"), equal_nan=True):
                bad.append({"field": "CURL", "after_iterations": h.iter - c["iter0"]})
            if c.get("subrect"):
                x0, y0 = int(rng.integers(0, X)), int(rng.integers(0, Y))
                w, hh = int(rng.integers(1, X - x0 + 1)), int(rng.integers(1, Y - y0 + 1))
                for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR") + (() if c["dry"] else ("LIGHT_1", "BASE_DISP")):
                    a, b = h.read_rect(f, x0, y0, w, hh), o.field(f)[y0:y0 + hh, x0:x0 + w]
                    if not np.array_equal(a, b, equal_nan=True):
                        bad.append({"field": f + " sub-rectangle", "rect": [x0, y0, w, hh], "after_iterations": h.iter - c["iter0"]})
            fields = list(GRID_FIELDS if not c["dry"] else ["BASE_CUR", "BASE_DISP", "WATER_CUR", "WATER_0", "WALL_CUR"])
            for f in fields:
                a, b = h.read_rect(f), o.field(f)
                if not np.array_equal(a, b, equal_nan=True):
                    ne = (a != b) & ~(np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
                    ys, xs = np.nonzero(ne.any(axis=-1))
                    bad.append({"field": f, "after_iterations": h.iter - c["iter0"], "values": int(ne.sum()), "first": [int(xs[0]), int(ys[0])],
                                "max_abs": float(np.nanmax(np.abs(a.astype(np.float64) - b)))})
            if nd:
                for f, a, b in (("DROPS", h.read_particles(), o.field("DROPS")), ("PRECIP_FB", h.read_rect("PRECIP_FB"), o.field("PRECIP_FB")),
                                ("PRECIP_DEP", h.read_rect("PRECIP_DEP"), o.field("PRECIP_DEP")), ("LIGHTNING", h.read_rect("LIGHTNING"), o.field("LIGHTNING"))):
                    if not np.array_equal(a, b, equal_nan=True):
                        bad.append({"field": f, "after_iterations": h.iter - c["iter0"], "values": int((a != b).sum())})
            if bad:
                break
        info = {"blown_up": not bool(np.isfinite(o.field("BASE_CUR")).all() and np.isfinite(o.field("WATER_CUR")).all() and np.abs(o.field("BASE_CUR")[..., :2]).max() < 1e4), "fastest": float(h.fastest_velocity()) if hasattr(h, "fastest_velocity") else None}
        try:
            info["pair_stats"] = h.pair_stats() if c["dry"] and c["pairs"] else None
        except Exception:
            info["pair_stats"] = None
    except E.WxError as e:  # an overflow of the exact path's list is REPORTED (a blown-up state), not a mismatch
        return [], {"error": str(e)}
    finally:
        h.close()
        o.close()
    return bad, info


def draw_group(rng, c):
    """Extra draws of --mode group (after draw_case, so the scene sequence of a seed is the same in both modes)."""
    n = int(rng.choice([2, 2, 3, 4, 4, 5, 6, 8]))
    if c["drops"]:  # wx_create_slab: with particles halo and owned width are multiples of 64 (splat tiles), owned + 2 halo <= X
        halo = 64
        xo = max(64 if n > 2 else 128, -(-c["X"] // n // 64) * 64)
    else:  # (a halo narrower than the flow's dependency cone is refused by the first step: reported, not a mismatch)
        lo = 6 if c["sigma"] <= 0.05 and not c["vortices"] else (12 if c["sigma"] <= 0.3 and not c["vortices"] else 24)
        halo = int(rng.choice([h for h in (6, 12, 18, 24, 42, 48, 64) if h >= lo]))
        xo = max(halo, -(-c["X"] // n))
    c["X"] = n * xo
    c.update(nslab=n, halo=halo, overlap=int(rng.random() < 0.6), split=int(rng.random() < 0.25), pool_exact=1, kernel_set=int(rng.random() < 0.9))
    c["steps"] = [int(v) for v in rng.integers(1, 2 * max(1, halo // 6) + 3, size=int(rng.integers(1, 4)))]
    return c


def run_group_case(pkg, E, c):
    """N slabs on this one GPU (wx_group_*: the library's own halo exchange, device-to-device copies) against the undecomposed handle, bit
    for bit -- SURVEY 8e's determinism check on random scenes, slab counts, halo widths, call boundaries, overlap / split-launch modes."""
    X, Y = c["X"], c["Y"]
    base, water, wall, u, drops = build_case(pkg, c)
    nd = 0 if drops is None else len(drops)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    g = whole = None
    bad = []
    try:
        g = E.Group(c["nslab"], X, Y, halo=c["halo"], devices=[0] * c["nslab"], transport=E.TRANSPORT_LOCAL, n_droplets=nd)
        whole = E.Handle(X, Y, nd)
        g.upload(base, water, wall, drops)
        whole.upload(base, water, wall, drops)
        g.set_params(p, u["initial_T"])
        whole.set_params(p, u["initial_T"])
        for hh in g.slabs + [whole]:
            hh.iter = c["iter0"]
        opts = [(E.Handle.OPT_DRY_PAIRS, c["pairs"]), (E.Handle.OPT_ROW_BANDS, c["bands"]), (E.Handle.OPT_DRY_KERNEL, c["dry_kernel"]), (E.Handle.OPT_KERNEL_SET, c["kernel_set"])]
        if nd:
            opts.append((E.Handle.OPT_SPLAT_ORDER, 1))
        for k, v in opts:
            g.set_option(k, v)
            whole.set_option(k, v)
        g.set_option(E.Handle.OPT_EXCHANGE_OVERLAP, c["overlap"])
        g.set_option(E.Handle.OPT_SPLIT_LAUNCH, c["split"])
        if nd:
            g.set_option(E.Handle.OPT_POOL_EXACT, c["pool_exact"])
        fields = ["BASE_CUR", "WATER_CUR", "WALL_CUR"] + ([] if c["dry"] else ["LIGHT_0", "LIGHT_1", "BASE_DISP", "WATER_0"]) + (["PRECIP_DEP"] if nd else [])
        done = 0
        for k_step, n in enumerate(c["steps"]):
            if c.get("reupload") and k_step == 1 and not nd:
                st = [whole.read_rect(f) for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR")]
                g.upload(*st)
                whole.upload(*st)
            if c.get("brush_toggle") and c["brush"] and k_step > 0:  # mouse up / down between two steps
                u2 = dict(u, userInputType=(c["brush"]["type"] if k_step % 2 == 0 else -1))
                p2 = pkg.params.fill_struct(pkg.params.WxParams(), u2)
                g.set_params(p2, u["initial_T"])
                whole.set_params(p2, u["initial_T"])
            g.step(n)
            whole.step(n)
            done += n
            for f in fields:
                a, b = g.read(f), whole.read_rect(f)
                if not np.array_equal(a, b, equal_nan=True):
                    ne = (a != b) & ~(np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
                    ys, xs = np.nonzero(ne.any(axis=-1))
                    bad.append({"field": f, "after_iterations": done, "values": int(ne.sum()), "first": [int(xs[0]), int(ys[0])]})
            if bad:
                break
        if nd and not bad:
            g.exchange()
            g.sync()
            d, d_ref = g.particles(), whole.read_particles()
            if not np.array_equal(d, d_ref, equal_nan=True):
                bad.append({"field": "DROPS", "after_iterations": done, "values": int((d != d_ref).sum())})
        wb = whole.read_rect("BASE_CUR")
        info = {"blown_up": not bool(np.isfinite(wb).all() and np.abs(wb[..., :2]).max() < 1e4), "fastest": float(np.abs(wb[..., :2]).max())}
    except E.WxError as e:  # reported: the exact path's list overflowed / a slab outran the |vx| bound its period was sized for (a blown-up state)
        return [], {"error": str(e)}
    finally:
        if g is not None:
            g.close()
        if whole is not None:
            whole.close()
    return bad, info


def run_setup_case(pkg, E, c):
    """--mode setup: the device-side initialisers (wx_setup_columns from 1-D descriptors, wx_setup_terrain with the shader's terrain noise on
    the device, wx_init_droplets) against the host generator + wx_upload: bit-identical textures (setup_terrain: all but the handful of
    columns whose height sits on a row boundary in the last bit of sin()), and the same run afterwards."""
    S = pkg.synth
    X, Y = c["X"], max(16, c["Y"])  # (wx_setup_terrain: at least 16 rows)
    snap = int([1, 2, 4][c["data_seed"] % 3])
    cloud = bool(c["cloud"])
    desc = S.terrain_columns(X, Y, seed=c["tseed"], height_mult=c["tmult"], snap=snap, cloud_deck=cloud)
    base, water, wall = S.terrain_grid(X, Y, seed=c["tseed"], height_mult=c["tmult"], snap=snap)
    if cloud:
        S.add_cloud_deck(water, wall)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = c["sun"]
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    a, b, t = E.Handle(X, Y, 0), E.Handle(X, Y, 0), E.Handle(X, Y, 0)
    bad = []
    try:
        a.upload(base, water, wall)
        b.setup_columns(desc)
        t.setup_terrain(S.sounding_rows(Y, cloud_deck=cloud), seed=c["tseed"], height_mult=c["tmult"], snap=snap, sim_height=float(gui["simHeight"]))
        for hh in (a, b):
            hh.set_params(p, u["initial_T"])
        terr = np.zeros(X, bool)
        for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR"):
            terr |= (b.read_rect(f) != t.read_rect(f)).any(axis=(0, 2))
        if terr.sum() > max(4, X // 200):
            bad.append({"field": "terrain", "what": "setup_terrain vs setup_columns: %d of %d columns differ" % (int(terr.sum()), X)})
        for when in ("after setup", "after 5 iterations"):
            for f in ("BASE_CUR", "WATER_CUR", "WATER_0", "WALL_CUR", "LIGHT_0"):
                if not np.array_equal(a.read_rect(f), b.read_rect(f)):
                    bad.append({"field": f, "what": "setup_columns vs upload, " + when})
            if bad:
                break
            a.step(5)
            b.step(5)
        info = {"blown_up": False, "fastest": 0.0, "terrain_columns_differing": int(terr.sum())}
    except E.WxError as e:
        return [], {"error": str(e)}
    finally:
        for hh in (a, b, t):
            hh.close()
    return bad, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=100000)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--max-cells", type=int, default=600000)
    ap.add_argument("--interleave", action="store_true", help="oracle mode: half of the cases run with a second handle alive, steps in turn")
    ap.add_argument("--big", action="store_true", help="grids of 1000-9000 x 512-2100 cells (use with --max-cells 8000000)")
    ap.add_argument("--mode", choices=["oracle", "group", "setup"], default="oracle", help="oracle: one handle against the CPU oracle; group: N slabs against one handle")
    ap.add_argument("--only", type=int, default=-1, help="run only case K of the seed's sequence")
    ap.add_argument("--first", type=int, default=0, help="skip the cases before this one (they are still drawn: same sequence)")
    ap.add_argument("--last", type=int, default=1 << 30)
    ap.add_argument("--override", default="", help="JSON object merged into the recipe of every case that runs (bisecting a failure)")
    ap.add_argument("--check-launches", action="store_true", help="WX_OPT_CHECK_LAUNCHES 1: synchronise and check after every launch")
    a = ap.parse_args()
    pkg = wxpkg.load_package()
    E = pkg.engine
    E.lib().wx_set_option(None, E.Handle.OPT_PLACEMENT_SEARCH, 0)
    if a.check_launches:
        E.lib().wx_set_option(None, E.Handle.OPT_CHECK_LAUNCHES, 1)
    import wx_oracle
    wx_oracle.build()
    rng = np.random.default_rng(a.seed)
    rng_il = np.random.default_rng(a.seed + 1000003)  # (companions come from a sequence of their own: --seed S --only K still reproduces case K alone)
    t0 = time.time()
    failures, ran, reported = [], 0, 0
    for k in range(a.cases):
        c = draw_case(rng, a.max_cells, a.big)
        if a.mode == "group":
            c = draw_group(rng, c)
        if (a.only >= 0 and k != a.only) or k < a.first:
            continue
        if k > a.last:
            break
        if time.time() - t0 > a.seconds:
            break
        if a.override:
            c.update(json.loads(a.override))
        t1 = time.time()
        if a.mode == "oracle" and a.interleave and rng_il.random() < 0.5 and a.only < 0:  # this case and a second small one, handles alive together
            c2 = draw_case(rng_il, min(a.max_cells, 60000))
            (bad, info), (bad2, info2) = run_interleaved(pkg, E, wx_oracle, [c, c2])
            if bad2 and not info2.get("blown_up"):
                failures.append({"case": k, "interleaved_with": c, "recipe": c2, "mismatches": bad2})
                print("MISMATCH in the interleaved companion:", json.dumps(failures[-1]), flush=True)
        else:
            bad, info = run_case(pkg, E, wx_oracle, c) if a.mode == "oracle" else (run_group_case(pkg, E, c) if a.mode == "group" else run_setup_case(pkg, E, c))
        ran += 1
        reported += 1 if info.get("error") else 0
        if bad and info.get("blown_up"):  # NaN / inf / |v| > 1e4 cells per iteration (the reference blows up the same way): float -> int conversions out of range differ between CPU and GPU
            print(f"case {k}: state not finite, {len(bad)} fields differ -- not counted", flush=True)
            bad = []
        tag = "MISMATCH" if bad else ("reported: " + info["error"][:60] if info.get("error") else "ok")
        print(f"case {k:4d} {c['X']:5d}x{c['Y']:<5d} {'dry' if c['dry'] else 'wet'} sigma {c['sigma']:.2f} steps {c['steps']} drops {c['drops']:4d} brush "
              f"{c['brush']['type'] if c['brush'] else '-':>2} pairs {c['pairs']} bands {c['bands']} set {c['kernel_set']}{c['dry_kernel']}{' slabs %d halo %d overlap %d split %d' % (c['nslab'], c['halo'], c['overlap'], c['split']) if a.mode == 'group' else ''} fastest {info.get('fastest')}{' terrain columns differing %s' % info.get('terrain_columns_differing') if a.mode == 'setup' else ''}  {time.time() - t1:.1f}s  {tag}", flush=True)
        if bad:
            failures.append({"case": k, "recipe": c, "mismatches": bad})
            print(json.dumps(failures[-1]), flush=True)
    print(json.dumps({"mode": a.mode, "seed": a.seed, "cases_run": ran, "mismatching_cases": len(failures), "cases_ending_in_a_reported_error": reported,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
