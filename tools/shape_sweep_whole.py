#!/usr/bin/env python3
"""Launch-shape sweep of the wet marching kernel on the WHOLE metric grid (16384 x 2048): WX_WET_* environment settings, one handle per
configuration, interleaved repetitions; kernel time from the engine's HIP events. Usage: python tools/shape_sweep_whole.py [X Y]"""
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
import os as _os
_dbg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so")
if "WXSIM_LIB" not in _os.environ and _os.path.exists(_dbg):
    _os.environ["WXSIM_LIB"] = _dbg
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools  # noqa: E402

X = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
if len(sys.argv) > 3:
    CONFIGS_OVERRIDE = [{}] + [{"WX_WET_SPEC": spec} for spec in sys.argv[3:]]
CONFIGS = [{}, {"WX_WET_SPEC": "4x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "3x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "2x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "2x1,1x0.6,1x0.3,1x0.15"},
           {"WX_WET_SPEC": "3x1,1x0.6,1x0.3,1x0.15"}, {"WX_WET_SPEC": "5x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "1x1,1x0.5,1x0.25,1x0.125"},
           {"WX_WET_SPEC": "4x1,1x0.5,1x0.25,1x0.125"}, {"WX_WET_SPEC": "3x1,2x0.5,1x0.25"}]


def make(env):
    for k in list(os.environ):
        if k.startswith("WX_WET_"):
            del os.environ[k]
    os.environ.update(env)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.15)
    h.step(20)
    h.sync()
    return h


def timed(h, steps=100):
    h.profile(True)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(steps // 10):
        h.step(10)
    h.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    p = h.profile_read()
    h.profile(False)
    ms, n = p["march_wet_full_iteration"]
    return dt, ms / n


if len(sys.argv) > 3:
    CONFIGS = CONFIGS_OVERRIDE
hs = [make(c) for c in CONFIGS]
res = [[] for _ in CONFIGS]
for rep in range(3):
    for i, h in enumerate(hs):
        res[i].append(timed(h))
for c, r in zip(CONFIGS, res):
    print(f"{X}x{Y}  step {' '.join(f'{t[0]:.4f}' for t in r)}  kernel {' '.join(f'{t[1]:.4f}' for t in r)}  {c}")
