#!/usr/bin/env python3
"""Launch-shape sweep of the wet marching kernel on ONE slab shape (default 2144 x 2048 = a configs[2] slab of an 8-GPU run).
Each configuration = environment settings read by wet_launch_shape (WX_WET_*); a fresh handle per configuration, interleaved repetitions.
Usage: python tools/slab_sweep.py [X_owned Y halo]"""
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

XO = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
HALO = int(sys.argv[3]) if len(sys.argv) > 3 else 48
X = XO * 8
CONFIGS = [{}, {"WX_WET_ROUNDS": "3"}, {"WX_WET_ROUNDS": "2"}, {"WX_WET_MINROWS": "32"}, {"WX_WET_MINROWS": "40"},
           {"WX_WET_SPEC": "7x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "5x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "4x1,1x0.5,1x0.25,1x0.125"},
           {"WX_WET_SPEC": "3x1,1x0.5,1x0.25"}, {"WX_WET_SPEC": "6x1,2x0.5,2x0.25"}, {"WX_WET_BANDS": "0"}, {"WX_WET_BANDS": "0", "WX_WET_ROUNDS": "2"}]


def make(env):
    for k in list(os.environ):
        if k.startswith("WX_WET_"):
            del os.environ[k]
    os.environ.update(env)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(XO, Y, 0, X_global=X, x0=3 * XO, halo=HALO)
    h.setup_columns(pkg.synth.terrain_columns(X, Y, cols=(3 * XO - HALO, XO + 2 * HALO)))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2, Xg=X, x0=3 * XO)
    h.step(20)  # (the launch shape is fixed by the first step: the environment is read there)
    h.sync()
    return h


def timed(h, steps=200):
    h.sync()
    t0 = time.perf_counter()
    for _ in range(steps // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / steps * 1e3


hs = [make(c) for c in CONFIGS]
res = [[] for _ in CONFIGS]
for rep in range(3):
    for i, h in enumerate(hs):
        res[i].append(timed(h))
for c, r in zip(CONFIGS, res):
    print(f"{XO + 2 * HALO}x{Y}  {' '.join(f'{t:.4f}' for t in r)}  {c}")
