#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE shaders (container-only; needs /root/reference + kaleido).

Runs oracle/golden/harness.js inside kaleido's HeadlessChrome/SwiftShader (software GL), which loads the
reference's shader files from /root/reference at run time, and stores inputs + outputs as small ``.npz``
fixtures under tests/golden/. Only data (arrays, uniform values) is stored -- no reference source.

SwiftShader caveat: advectionShader output is corrupted for air pixels that share a
2x2 pixel quad with a wall pixel when the pass is drawn as the reference's full-screen quad. The round-1 fixtures therefore
keep wall/air boundaries on even x and even y; the round-2 fixtures (save100raw, randwalls64p, lightning64, airplane64,
setup256) draw every pass as one GL_POINT per pixel instead (harness.js, job option `points`), which has no neighbouring
fragments to go wrong with, and put walls anywhere -- including the reference's unmodified save.

usage:  python oracle/golden/gen_golden.py [fixture ...]     (default: all)
"""
from __future__ import annotations

import base64
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
OUT_DIR = os.path.join(ROOT, "tests", "golden")
REF_SAVE = "/root/reference/saves/100 X 100 Test.weathersandbox"


def kaleido_exe() -> str:
    import kaleido
    return os.path.join(os.path.dirname(kaleido.__file__), "executable", "kaleido")


def run_harness(job: dict, timeout: float = 600.0) -> dict:
    """One harness run; ``job`` is passed as gd.layout.wx."""
    req = {"data": {"data": [], "layout": {"wx": job}}, "format": "json", "width": job["X"], "height": job["Y"], "scale": 1}
    cmd = [kaleido_exe(), "plotly", "--plotlyjs=" + os.path.join(HERE, "harness.js"), "--disable-gpu",
           "--allow-file-access-from-files", "--disable-breakpad", "--disable-dev-shm-usage", "--no-sandbox"]
    p = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        p.stdin.write((json.dumps(req) + "\n").encode())
        p.stdin.flush()
        t0 = time.time()
        result = None
        while time.time() - t0 < timeout:
            line = p.stdout.readline()
            if not line:
                break
            try:
                msg = json.loads(line.decode())
            except Exception:
                continue
            if "result" in msg and msg.get("result") is not None:
                result = msg
                break
            if msg.get("code", 0) != 0:
                raise RuntimeError(f"kaleido error: {msg}")
        if result is None:
            raise RuntimeError("no result from kaleido")
    finally:
        try:
            p.stdin.close()
        except Exception:
            pass
        p.kill()
        p.wait()
    res = result["result"]
    if isinstance(res, str):
        res = json.loads(res)
    if "error" in res:
        raise RuntimeError("harness: " + res["error"])
    return res


def _dec(s: str, dtype) -> np.ndarray:
    return np.frombuffer(base64.b64decode(s), dtype=dtype).copy()


def js_uniforms(u: dict) -> dict:
    o = {}
    for k, v in u.items():
        if k in ("initial_T", "sounding_T", "sounding_W", "sounding_Vel"):
            continue
        o[k] = list(v) if isinstance(v, tuple) else v
    return o


def run_fixture(name, X, Y, base, water, wall, drops, u, *, niter, dump_iters, perpass_iter=None, precip=False,
                iter0=0, keep=("base_cur", "water_cur", "wall_cur", "light_0", "light_1", "water_0", "base_disp"), points=False,
                dump_emitted=False, keep_particles=("drops", "lightning", "precip_fb", "precip_dep"), timeout=600.0):
    tmp = tempfile.mkdtemp(prefix="wxgold_")
    np.ascontiguousarray(base, np.float32).tofile(os.path.join(tmp, "base.f32"))
    np.ascontiguousarray(water, np.float32).tofile(os.path.join(tmp, "water.f32"))
    np.ascontiguousarray(wall, np.int8).tofile(os.path.join(tmp, "wall.i8"))
    n_drops = 0 if drops is None else len(drops)
    if n_drops:
        np.ascontiguousarray(drops, np.float32).tofile(os.path.join(tmp, "drops.f32"))
    job = {
        "X": X, "Y": Y, "n_drops": n_drops, "dir": "file://" + tmp + "/",
        "uniforms": js_uniforms(u), "initial_T": [float(v) for v in u["initial_T"]],
        "niter": niter, "dump_iters": list(dump_iters), "precip": bool(precip), "iter0": iter0, "points": bool(points),
        "dump_emitted": bool(dump_emitted),
    }
    if "sounding_T" in u:
        job["sounding"] = {k: [float(v) for v in u["sounding_" + k]] for k in ("T", "W", "Vel")}
    if perpass_iter is not None:
        job["perpass_iter"] = perpass_iter
    probe = run_harness({"X": X, "Y": Y, "probe": True, "n_drops": 0, "points": bool(points)})
    varyings = _dec(probe["probe"], np.float32).reshape(Y, X, 4)
    res = run_harness(job, timeout=timeout)
    print(f"[{name}] renderer={res['renderer']!r} err={res['err']} {res['niter']} it, "
          f"{res['ms_after_first']:.1f} ms after first -> {1000.0 * (res['niter'] - 1) / max(res['ms_after_first'], 1e-9):.1f} it/s")
    out = {
        "X": X, "Y": Y, "iter0": iter0, "niter": niter, "precip": int(bool(precip)),
        "in_base": np.asarray(base, np.float32).reshape(Y, X, 4), "in_water": np.asarray(water, np.float32).reshape(Y, X, 4),
        "in_wall": np.asarray(wall, np.int8).reshape(Y, X, 4),
        "initial_T": np.asarray(u["initial_T"], np.float32),
        # simShader.vert varyings (fragCoord.xy, texCoord.xy) as interpolated by the reference's rasteriser here
        "varyings": varyings,
        "uniforms_json": json.dumps(js_uniforms(u)),
        **({"sounding_" + k: np.asarray(u["sounding_" + k], np.float32) for k in ("T", "W", "Vel")} if "sounding_T" in u else {}),
        "renderer": res["renderer"], "its_per_s": 1000.0 * (res["niter"] - 1) / max(res["ms_after_first"], 1e-9),
        "points": int(bool(points)),  # 1: every pass drawn as one GL_POINT per pixel (see harness.js), 0: as the reference's quad
    }
    if n_drops:
        out["in_drops"] = np.asarray(drops, np.float32).reshape(n_drops, 5)
    shapes = {"curl": (Y, X), "vort": (Y, X, 2), "precip_dep": (Y, X, 2), "drops": (-1, 5), "precip_drops": (-1, 5), "lightning": (4,)}
    for it, d in res["dumps"].items():
        for k, v in d.items():
            if k not in keep and k not in keep_particles and not (dump_emitted and k == "emitted"):
                continue
            dt = np.int8 if "wall" in k else np.float32
            out[f"it{it}_{k}"] = _dec(v, dt).reshape(shapes.get(k, (Y, X, 4)))
    for k, v in res.get("perpass", {}).items():
        dt = np.int8 if "wall" in k else np.float32
        out[f"pp_{k}"] = _dec(v, dt).reshape(shapes.get(k, (Y, X, 4)))
    if "inactiveDroplets" in res:
        out["inactiveDroplets"] = res["inactiveDroplets"]
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    return out


# ------------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------------
def save100_quad_aligned():
    """The reference's only save, with terrain snapped to even x/y: sea and coast thickened to rows 0-1,
    the 20-column island raised to rows 0-3."""
    sf = pkg.codec.load(REF_SAVE)
    base, water, wall = sf.base.copy(), sf.water.copy(), sf.wall.copy()
    air1 = wall[1, :, 1] != 0
    for arr in (base, water, wall):
        arr[1, air1] = arr[0, air1]
    island = ~air1
    for y in (2, 3):
        for arr in (base, water, wall):
            arr[y, island] = arr[1, island]
    return sf, base, water, wall


def fx_save100(precip: bool):
    sf, base, water, wall = save100_quad_aligned()
    gui = pkg.params.merge_settings(sf.settings)
    u = pkg.params.uniforms_from_gui(gui, sf.Y)
    name = "save100qa_precip" if precip else "save100qa"
    if precip:  # particle state / feedback over 50 iterations; grid fields are covered by save100qa
        return run_fixture(name, sf.X, sf.Y, base, water, wall, sf.droplets, u, niter=50, dump_iters=[1, 10, 50],
                           precip=True, keep=("base_cur",))
    return run_fixture(name, sf.X, sf.Y, base, water, wall, None, u,
                       niter=50, dump_iters=[1, 10, 50], perpass_iter=0, precip=False,
                       keep=("base_cur", "water_cur", "wall_cur", "light_0", "light_1"))


def synth_terrain(X, Y, rng):
    """Quad-aligned terrain with every wall type, snow, vegetation, moist warm air and cloud."""
    gui = dict(pkg.params.GUI_DEFAULTS)
    gui["sunAngle"] = 60.0
    u = pkg.params.uniforms_from_gui(gui, Y)
    T0 = u["initial_T"]
    base = np.zeros((Y, X, 4), np.float32)
    water = np.zeros((Y, X, 4), np.float32)
    wall = np.zeros((Y, X, 4), np.int8)
    height = np.full(X, 2)
    height[8:20] = 4
    height[12:16] = 8
    height[28:40] = 2
    height[40:44] = 6
    types = np.full(X, 2)  # sea
    types[6:24] = 1  # land island with hill
    types[28:34] = 4  # urban
    types[34:40] = 6  # industrial
    types[40:44] = 1
    types[44:48] = 3  # fire
    types[48:52] = 5  # runway
    types[52:56] = 0  # inert
    for x in range(X):
        h = height[x]
        wall[:h, x, 0] = types[x]
        wall[:h, x, 1] = 0
        wall[:h, x, 2] = np.arange(-(h - 1), 1)
        wall[:h, x, 3] = {1: 60 + (x % 7) * 9, 3: 90, 4: 40, 6: 10}.get(int(types[x]), 0)
        wall[h:, x, 0] = types[x]
        wall[h:, x, 1] = np.minimum(np.arange(1, Y - h + 1), 127)
        wall[h:, x, 2] = np.minimum(np.arange(1, Y - h + 1), 127)
        if types[x] == 2:
            base[:h, x, 3] = 298.15 + 0.05 * (x % 5)
            water[:h, x, 0] = 1002.0
            water[:h, x, 2] = 100.0
        else:
            base[:h, x, 3] = 1000.0
            water[:h, x, 0] = 1001.0
            water[:h, x, 2] = 5.0 + (x % 11) * 3.0  # soil moisture
            water[:h, x, 3] = 12.0 if 12 <= x < 16 else 0.0  # snow on the hill
    yy = np.arange(Y)[:, None]
    air = wall[..., 1] != 0
    base[..., 3] = np.where(air, T0[:Y][:, None] + rng.normal(0, 0.3, (Y, X)).astype(np.float32) + 2.0 * np.exp(-((yy - 10) / 6.0) ** 2), base[..., 3])
    base[..., 0] = np.where(air, rng.normal(0, 0.02, (Y, X)), 0).astype(np.float32)
    base[..., 1] = np.where(air, rng.normal(0, 0.02, (Y, X)), 0).astype(np.float32)
    base[..., 2] = np.where(air, rng.normal(0, 0.002, (Y, X)), base[..., 2]).astype(np.float32)
    realT = base[..., 3] - ((yy + 0.5) / Y) * u["dryLapse"]
    maxw = (realT / 250.0) ** 17
    tot = maxw * (0.7 + 0.5 * rng.random((Y, X)))
    water[..., 0] = np.where(air, tot, water[..., 0]).astype(np.float32)
    water[..., 1] = np.where(air, np.maximum(tot - maxw, 0), water[..., 1]).astype(np.float32)
    water[..., 2] = np.where(air, 0.05 * rng.random((Y, X)), water[..., 2]).astype(np.float32)
    water[..., 3] = np.where(air, 0.3 * rng.random((Y, X)) * (yy < 14), water[..., 3]).astype(np.float32)
    water[4:6, 44:48, 3] = 5.0  # flames above the fire cells
    return gui, u, base, water, wall


def fx_synth64():
    rng = np.random.default_rng(1234)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    # iterations 95..104: crosses iterNum%100==0 (soil smoothing, vegetation, fire spread) and %20==0 (sea T)
    return run_fixture("synth64", X, Y, base, water, wall, None, u, niter=10, dump_iters=[1, 5, 6, 10],
                       perpass_iter=5, precip=False, iter0=95,
                       keep=("base_cur", "water_cur", "wall_cur", "light_0", "light_1", "water_0"))


def fx_sounding64():
    """Real-sounding forcing (advectionShader.frag:154-181 with soundingForcing != 0) + globalDrying + globalHeating inside
    an altitude window: the per-row arrays are what app.js:5444-5463 builds from a sounding (here: synthetic profiles)."""
    rng = np.random.default_rng(77)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    sim_h = float(gui["simHeight"])
    u["soundingForcing"] = 0.6
    u["globalDrying"] = 2e-5
    u["globalHeating"] = 1e-4
    u["globalEffectsStartAlt"] = 1500.0 / sim_h
    u["globalEffectsEndAlt"] = 9000.0 / sim_h
    y = np.arange(Y + 1, dtype=np.float64)
    real_t = 292.0 - 70.0 * y / Y + 3.0 * np.sin(y * 0.4)  # K, with an inversion-like wiggle
    u["sounding_T"] = (real_t + (y / Y) * u["dryLapse"]).astype(np.float32)  # realToPotentialT
    u["sounding_W"] = (((real_t - 4.0 - 6.0 * (y / Y)) / 250.0) ** 17).astype(np.float32)  # maxWater(dew point)
    u["sounding_Vel"] = (0.05 + 0.25 * y / Y).astype(np.float32)  # cells / iteration
    return run_fixture("sounding64", X, Y, base, water, wall, None, u, niter=10, dump_iters=[1, 10], perpass_iter=0, precip=False,
                       keep=("base_cur", "water_cur", "wall_cur"))


def fx_precip64():
    """Particle pass: hand-built droplet set over a cloudy field (spawn / grow / freeze / melt / deposit)."""
    rng = np.random.default_rng(99)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    yy = np.arange(Y)[:, None]
    air = wall[..., 1] != 0
    # dense cloud deck (warm below, cold above) so inactive droplets do spawn
    deck = air & (yy >= 14) & (yy < 40)
    water[..., 1] = np.where(deck, 1.2 + 2.5 * rng.random((Y, X)), water[..., 1]).astype(np.float32)
    water[..., 0] = np.where(deck, water[..., 0] + water[..., 1], water[..., 0]).astype(np.float32)
    u["spawnChanceMult"] = 0.02
    u["enablePrecipitation"] = 1
    n = 256
    drops = np.zeros((n, 5), np.float32)
    drops[:, 0] = rng.random(n)
    drops[:, 1] = rng.random(n)
    drops[:, 2] = -10.0 + rng.random(n)
    drops[:, 3] = rng.random(n)
    drops[:, 4] = rng.random(n)
    k = 96  # active ones
    drops[:k, 0] = rng.uniform(-0.98, 0.98, k)
    drops[:k, 1] = rng.uniform(-0.9, 0.9, k)
    drops[:k, 2] = rng.uniform(0.0, 0.6, k)  # water
    drops[:k, 3] = np.where(rng.random(k) < 0.5, rng.uniform(0.0, 0.8, k), 0.0)  # ice
    drops[:k, 4] = np.where(drops[:k, 3] > 0, rng.uniform(0.2, 1.0, k), 1.0)
    drops[:8, 2] = 0.01  # too small -> evaporate
    drops[:8, 3] = 0.01
    drops[8:16, 1] = -0.97  # inside the ground -> deposit
    drops[16:20, 1] = -0.999
    return run_fixture("precip64", X, Y, base, water, wall, drops, u, niter=4, dump_iters=[1, 2, 4],
                       perpass_iter=0, precip=True,
                       keep=("base_cur", "water_cur", "wall_cur", "light_0", "light_1"))


BRUSH_CASES = [
    # (name, userInputType, (x, y), intensity, iterations); wall-editing tools run ONE iteration: afterwards the
    # circular edit is no longer quad aligned and SwiftShader's advection bug would corrupt the golden
    ("temperature", 1, (0.30, 0.30), 0.5, 3), ("temperature_sea", 1, (0.02, 0.03), 0.5, 3), ("water", 2, (0.30, 0.45), 0.1, 3),
    ("water_neg", 2, (0.30, 0.45), -0.1, 3), ("smoke", 3, (0.60, 0.30), 0.05, 3), ("wind", 4, (0.50, 0.50), 0.8, 3),
    ("wholewidth_temp", 1, (-1.0, 0.40), 0.2, 3), ("wholewidth_wind", 4, (-1.0, 0.40), 0.8, 3),
    ("wall_inert", 10, (0.70, 0.40), 0.01, 1), ("wall_land", 11, (0.30, 0.30), 0.01, 1), ("wall_sea", 12, (0.55, 0.20), 0.01, 1),
    ("wall_remove", 10, (0.22, 0.10), -0.01, 1), ("fire", 13, (0.20, 0.13), 0.01, 1), ("fire_out", 13, (0.72, 0.05), -0.01, 1),
    ("urban", 14, (0.20, 0.13), 0.01, 1), ("runway", 15, (0.50, 0.05), 0.01, 1), ("industrial", 16, (0.66, 0.10), 0.01, 1),
    ("urban_remove", 14, (0.48, 0.03), -0.01, 1), ("moisture", 20, (0.20, 0.13), 0.5, 3), ("snow", 21, (0.20, 0.13), 0.5, 3),
    ("snow_remove", 21, (0.22, 0.15), -0.5, 3), ("vegetation", 22, (0.20, 0.13), 0.01, 3), ("vegetation_remove", 22, (0.20, 0.13), -0.01, 3),
]


def fx_brush64():
    """User-brush branch of advectionShader.frag:229-401: every tool once, on the synth64 terrain."""
    rng = np.random.default_rng(1234)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    probe = run_harness({"X": X, "Y": Y, "probe": True, "n_drops": 0})
    out = {"X": X, "Y": Y, "in_base": base, "in_water": water, "in_wall": wall, "initial_T": np.asarray(u["initial_T"], np.float32),
           "varyings": _dec(probe["probe"], np.float32).reshape(Y, X, 4), "cases": json.dumps([c[0] for c in BRUSH_CASES])}
    tmp = tempfile.mkdtemp(prefix="wxgold_")
    base.tofile(os.path.join(tmp, "base.f32"))
    water.tofile(os.path.join(tmp, "water.f32"))
    wall.tofile(os.path.join(tmp, "wall.i8"))
    for name, ut, (bx, by), inten, nit in BRUSH_CASES:
        uu = dict(u, userInputType=ut, userInputValues=(bx, by, inten, 6.0), userInputMove=(0.004, -0.002))
        job = {"X": X, "Y": Y, "n_drops": 0, "dir": "file://" + tmp + "/", "uniforms": js_uniforms(uu),
               "initial_T": [float(v) for v in u["initial_T"]], "niter": nit, "dump_iters": [nit], "precip": False, "iter0": 0}
        res = run_harness(job)
        d = res["dumps"][str(nit)]
        out[f"{name}_uniforms"] = json.dumps(js_uniforms(uu))
        out[f"{name}_niter"] = nit
        out[f"{name}_base"] = _dec(d["base_cur"], np.float32).reshape(Y, X, 4)
        out[f"{name}_water"] = _dec(d["water_cur"], np.float32).reshape(Y, X, 4)
        out[f"{name}_wall"] = _dec(d["wall_cur"], np.int8).reshape(Y, X, 4)
        changed = int((out[f"{name}_wall"] != wall).any(-1).sum())
        print(f"[brush64] {name}: type {ut}, {nit} it, err={res['err']}, wall cells changed vs input: {changed}")
    path = os.path.join(OUT_DIR, "brush64.npz")
    np.savez_compressed(path, **out)
    print(f"[brush64] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def fx_randwalls64():
    """Random 2x2-aligned wall blocks of every type: floating islands, overhangs, caves, one-block-wide gaps, sea next to
    air (dyke rule), walls in the top rows (y wrap) -- the wall-geometry branches of boundaryShader.frag:155-196,
    :245-269 and :373-388 that smooth terrain never reaches. One iteration at iterNum = 100 (soil / snow smoothing, vegetation, fire spread branches)."""
    rng = np.random.default_rng(4242)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    nb = 70
    for _ in range(nb):
        bx, by = int(rng.integers(0, X // 2)) * 2, int(rng.integers(1, Y // 2)) * 2
        w, h = int(rng.integers(1, 4)) * 2, int(rng.integers(1, 3)) * 2
        t = int(rng.integers(0, 7))
        ys, xs = slice(by, min(by + h, Y)), slice(bx, min(bx + w, X))
        wall[ys, xs, 0] = t
        wall[ys, xs, 1] = 0
        wall[ys, xs, 2] = 0
        wall[ys, xs, 3] = int(rng.integers(0, 120))
        base[ys, xs, 0:2] = 0.0
        base[ys, xs, 3] = 298.15 if t == 2 else 1000.0
        water[ys, xs, 0] = 1002.0 if t == 2 else 1001.0
        water[ys, xs, 1] = 0.0
        water[ys, xs, 2] = float(rng.integers(0, 60))
        water[ys, xs, 3] = float(rng.integers(0, 3)) * 6.0
    # Only the per-pass dumps up to the boundary pass are usable: the boundary pass grows walls by one row (fill rules),
    # after which they are no longer quad aligned and SwiftShader's advection output is garbage.
    out = run_fixture("randwalls64", X, Y, base, water, wall, None, u, niter=1, dump_iters=[], perpass_iter=0, precip=False,
                      iter0=100, keep=())
    path = os.path.join(OUT_DIR, "randwalls64.npz")
    keep = {k: v for k, v in out.items() if not k.startswith("pp_") or k.split("_")[1] in ("velocity", "curl", "vort", "boundary")}
    np.savez_compressed(path, **keep)
    print(f"[randwalls64] trimmed to the passes before advection: {os.path.getsize(path) / 1024:.0f} KiB")
    return keep


# ------------------------------------------------------------------------------------------------
# round-2 fixtures: every pass drawn as GL_POINTS (harness.js `points`), which side-steps SwiftShader's mixed-quad bug, so
# walls no longer have to sit on even x / even y
# ------------------------------------------------------------------------------------------------
def fx_save100raw():
    """BASELINE configs[0]: the reference's UNMODIFIED save (one-row sea, 20 x 2 island: nothing quad aligned), 1000
    iterations, sun fixed at the saved angle, precipitation off. Early dumps carry everything, late ones the three
    state textures."""
    sf = pkg.codec.load(REF_SAVE)
    gui = pkg.params.merge_settings(sf.settings)
    u = pkg.params.uniforms_from_gui(gui, sf.Y)
    return run_fixture("save100raw", sf.X, sf.Y, sf.base, sf.water, sf.wall, None, u, niter=1000, dump_iters=[1, 10, 50, 200, 1000],
                       perpass_iter=0, precip=False, keep=("base_cur", "water_cur", "wall_cur"), points=True, timeout=1800.0)


def fx_randwalls64p():
    """Random 1-cell-granular wall blocks of every type (floating islands, overhangs, caves, one-cell gaps, sea next to air,
    walls in the top rows) through the WHOLE iteration for 12 iterations from iterNum = 95 (crosses % 100 and % 20): bilerpWall
    next to irregular walls, the wall branch of advection, pressure and lighting on real terrain."""
    rng = np.random.default_rng(777)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    for _ in range(90):
        bx, by = int(rng.integers(0, X)), int(rng.integers(2, Y))
        w, h = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        t = int(rng.integers(0, 7))
        ys, xs = slice(by, min(by + h, Y)), slice(bx, min(bx + w, X))
        wall[ys, xs, 0] = t
        wall[ys, xs, 1] = 0
        wall[ys, xs, 2] = 0
        wall[ys, xs, 3] = int(rng.integers(0, 120))
        base[ys, xs, 0:2] = 0.0
        base[ys, xs, 3] = 298.15 if t == 2 else 1000.0
        water[ys, xs, 0] = 1002.0 if t == 2 else 1001.0
        water[ys, xs, 1] = 0.0
        water[ys, xs, 2] = float(rng.integers(0, 60))
        water[ys, xs, 3] = float(rng.integers(0, 3)) * 6.0
    return run_fixture("randwalls64p", X, Y, base, water, wall, None, u, niter=12, dump_iters=[1, 2, 6, 12], perpass_iter=0, precip=False,
                       iter0=95, keep=("base_cur", "water_cur", "wall_cur", "light_0", "light_1", "water_0"), points=True,
                       dump_emitted=True)


def _emitted_scene(seed):
    """synth64's terrain (every wall type; irregular blocks would grow into the sky within 60 iterations) plus smoke plumes dense enough
    to glow (lightingShader.frag:143-148: opacity > 0.8 <=> smoke > 4)."""
    rng = np.random.default_rng(seed)
    X, Y = 64, 64  # (Y >= 50: below that the top row passes boundaryShader.frag:192's `texCoord.y < 0.99` and the sky fills with wall)
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    air = wall[..., 1] != 0
    for _ in range(12):  # smoke plumes, 0 .. 14 g/m3
        cx, cy, r = rng.integers(0, X), rng.integers(4, Y - 4), rng.integers(2, 6)
        yy, xx = np.mgrid[0:Y, 0:X]
        d = np.hypot(xx - cx, yy - cy)
        water[..., 3] = np.where(air & (d < r), np.maximum(water[..., 3], 14.0 * (1.0 - d / r)), water[..., 3]).astype(np.float32)
    return X, Y, gui, u, base, water, wall


def fx_emitted64():
    """The lighting pass's second render target (emittedLight, RGBA16F): daylight at -30 degrees for 80 iterations (sunlight has
    reached the ground: air scattering, cloud / precipitation reflection, ground reflection, smoke glow) and a sun 2 degrees above
    the horizon for 6 iterations (red sunlight colour, urban / industrial / runway night glow). Dumped together with the light
    textures one iteration earlier, so the pass can be checked on the reference's own inputs."""
    X, Y, gui, u, base, water, wall = _emitted_scene(31337)
    keep = ("base_cur", "water_cur", "wall_cur", "light_0", "light_1")
    u_day = dict(u)
    u_day["sunAngle"] = float(np.deg2rad(-30.0))
    day = run_fixture("emitted64_day", X, Y, base, water, wall, None, u_day, niter=80, dump_iters=[1, 40, 79, 80], precip=False, iter0=7,
                      keep=keep, points=True, dump_emitted=True)
    u_night = dict(u)
    u_night["sunAngle"] = float(np.deg2rad(88.0))
    night = run_fixture("emitted64_night", X, Y, base, water, wall, None, u_night, niter=6, dump_iters=[5, 6], precip=False, iter0=7,
                        keep=keep, points=True, dump_emitted=True)
    return day, night


def fx_lightning64():
    """Lightning: a cold, very dense cloud deck (cloud + precipitation > 2.5 below 0 C) and a large pool of inactive droplets,
    from iterNum = 40 (> the 30-iteration lock-out of a fresh lightning texture), so that precipitationShader.vert:121-140
    requests strikes and lightningLocationShader.frag:24-38 accepts single ones, rejects double ones and the lock-out holds.
    EVERY iteration is dumped (post-advection base / water, droplets, feedback, lightning texture): the oracle's particle pass is
    checked per iteration on the reference's own inputs, because the strike decision hashes temperature BITS."""
    rng = np.random.default_rng(2024)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    yy = np.arange(Y)[:, None]
    air = wall[..., 1] != 0
    deck = air & (yy >= 22) & (yy < 44)  # rows where the real temperature is below freezing
    water[..., 1] = np.where(deck, 4.0 + 5.0 * rng.random((Y, X)), water[..., 1]).astype(np.float32)
    water[..., 0] = np.where(deck, water[..., 0] + water[..., 1], water[..., 0]).astype(np.float32)
    water[..., 2] = np.where(deck, 1.0 * rng.random((Y, X)), water[..., 2]).astype(np.float32)
    u["spawnChanceMult"] = 0.02
    u["enablePrecipitation"] = 1
    n = 1024
    drops = np.zeros((n, 5), np.float32)
    drops[:, 0] = rng.random(n)
    drops[:, 1] = rng.random(n)
    drops[:, 2] = -10.0 + rng.random(n)
    drops[:, 3] = rng.random(n)
    drops[:, 4] = rng.random(n)
    u["inactiveDroplets"] = float(n)
    niter = 48
    return run_fixture("lightning64", X, Y, base, water, wall, drops, u, niter=niter, dump_iters=list(range(1, niter + 1)), precip=True,
                       iter0=40, keep=("base_disp", "water_cur"), points=True)


AIRPLANE_CASES = [
    # airplaneValues = (x, y, -, mode): mode < 0 dumps water (advectionShader.frag:436-439), mode > 0.9 is a crash (:441-456)
    ("airplane_dump", (0.4, 0.5, 0.7, -1.0)), ("airplane_crash_air", (0.6, 0.6, 0.0, 1.0)), ("airplane_crash_ground", (0.2, 0.09, 0.0, 1.0)),
]


def fx_airplane64():
    """Airplane inputs of advectionShader.frag:415-457 on the synth64 terrain, 2 iterations each."""
    rng = np.random.default_rng(1234)
    X, Y = 64, 48
    gui, u, base, water, wall = synth_terrain(X, Y, rng)
    probe = run_harness({"X": X, "Y": Y, "probe": True, "n_drops": 0, "points": True})
    out = {"X": X, "Y": Y, "in_base": base, "in_water": water, "in_wall": wall, "initial_T": np.asarray(u["initial_T"], np.float32),
           "varyings": _dec(probe["probe"], np.float32).reshape(Y, X, 4), "cases": json.dumps([c[0] for c in AIRPLANE_CASES]), "points": 1}
    tmp = tempfile.mkdtemp(prefix="wxgold_")
    base.tofile(os.path.join(tmp, "base.f32"))
    water.tofile(os.path.join(tmp, "water.f32"))
    wall.tofile(os.path.join(tmp, "wall.i8"))
    nit = 2
    for name, av in AIRPLANE_CASES:
        uu = dict(u, userInputType=-1, airplaneValues=av)
        job = {"X": X, "Y": Y, "n_drops": 0, "dir": "file://" + tmp + "/", "uniforms": js_uniforms(uu), "points": True,
               "initial_T": [float(v) for v in u["initial_T"]], "niter": nit, "dump_iters": [nit], "precip": False, "iter0": 0}
        res = run_harness(job)
        d = res["dumps"][str(nit)]
        out[f"{name}_uniforms"] = json.dumps(js_uniforms(uu))
        out[f"{name}_niter"] = nit
        out[f"{name}_base"] = _dec(d["base_cur"], np.float32).reshape(Y, X, 4)
        out[f"{name}_water"] = _dec(d["water_cur"], np.float32).reshape(Y, X, 4)
        out[f"{name}_wall"] = _dec(d["wall_cur"], np.int8).reshape(Y, X, 4)
        print(f"[airplane64] {name}: err={res['err']}, wall cells changed vs input: {int((out[f'{name}_wall'] != wall).any(-1).sum())}")
    path = os.path.join(OUT_DIR, "airplane64.npz")
    np.savez_compressed(path, **out)
    print(f"[airplane64] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def fx_setup256():
    """The setup draw of a new simulation (setupShader.frag:36-92; uniforms app.js:5479-5485, 5732-5734) at 4096 x 96 (the noise
    terrain needs a few thousand columns to rise out of the sea) with the defaults synth.terrain_grid documents (seed 0.5,
    heightMult 0.3). Stored compactly: per-column wall height / type / vegetation / snow, per-row air state."""
    X, Y = 4096, 96
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y)
    job = {"X": X, "Y": Y, "n_drops": 0, "points": True, "initial_T": [float(v) for v in u["initial_T"]],
           "setup": {"seed": 0.5, "heightMult": 0.3, "simHeight": float(gui["simHeight"]), "dryLapse": float(u["dryLapse"])}}
    res = run_harness(job)
    out = {"X": X, "Y": Y, "seed": 0.5, "heightMult": 0.3, "simHeight": float(gui["simHeight"]), "dryLapse": float(u["dryLapse"]),
           "initial_T": np.asarray(u["initial_T"], np.float32), "renderer": res["renderer"],
           "base": _dec(res["base"], np.float32).reshape(Y, X, 4), "water": _dec(res["water"], np.float32).reshape(Y, X, 4),
           "wall": _dec(res["wall"], np.int8).reshape(Y, X, 4)}
    path = os.path.join(OUT_DIR, "setup256.npz")
    # the output is column / row structured: keep the full wall texture (int8, compresses well) and float planes as they are
    np.savez_compressed(path, **out)
    print(f"[setup256] err={res['err']} wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); wall cells: {int((out['wall'][..., 1] == 0).sum())}")
    return out


FIXTURES = {
    "randwalls64": fx_randwalls64,
    "brush64": fx_brush64,
    "save100qa": lambda: fx_save100(False),
    "save100qa_precip": lambda: fx_save100(True),
    "synth64": fx_synth64,
    "sounding64": fx_sounding64,
    "precip64": fx_precip64,
    "save100raw": fx_save100raw,
    "randwalls64p": fx_randwalls64p,
    "lightning64": fx_lightning64,
    "emitted64": fx_emitted64,
    "airplane64": fx_airplane64,
    "setup256": fx_setup256,
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(FIXTURES)
    for nm in names:
        FIXTURES[nm]()
