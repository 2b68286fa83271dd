#!/usr/bin/env python3
"""Launch-shape sweep of the wet marching kernel on ONE handle (one placement of the planes: the allocation lottery of
shape_sweep_whole.py -- a handle per configuration, +-7 % from the placement alone -- drops out). The debug build re-reads WX_WET_*
whenever the cached shape is invalidated, which WX_OPT_ROW_BANDS does.
Usage: python tools/shape_sweep_one_handle.py [X Y [halo]] spec...   (spec = WX_WET_SPEC list, or KEY=VALUE[,KEY=VALUE], or "-" for the default)"""
import os as _os
_dbg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so")
if "WXSIM_LIB" not in _os.environ and _os.path.exists(_dbg):
    _os.environ["WXSIM_LIB"] = _dbg
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools  # noqa: E402

args = sys.argv[1:]
nums = []
while args and args[0].isdigit():
    nums.append(int(args.pop(0)))
X = nums[0] if len(nums) > 0 else 16384
Y = nums[1] if len(nums) > 1 else 2048
HALO = nums[2] if len(nums) > 2 else 0
REPS = int(os.environ.get("REPS", "3"))
TUNE = int(os.environ.get("TUNE", "4"))
specs = args or ["-"]


def env_of(spec):
    if spec == "-":
        return {}
    if "=" in spec:
        return dict(kv.split("=", 1) for kv in spec.split(";"))
    return {"WX_WET_SPEC": spec}


gui = pkg.params.merge_settings(None)
gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
u["enablePrecipitation"] = 0
if HALO:
    h = pkg.engine.Handle(X, Y, 0, X_global=8 * X, x0=3 * X, halo=HALO)  # (X = owned columns of one of eight slabs)
    h.setup_columns(pkg.synth.terrain_columns(8 * X, Y, cols=(3 * X - HALO, X + 2 * HALO)))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2, Xg=8 * X, x0=3 * X)
else:
    h = pkg.engine.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.15)
h.step(40)
h.sync()
if TUNE and not HALO:
    print("placement:", h.tune_placement(TUNE), flush=True)


def apply(env):
    for k in list(os.environ):
        if k.startswith("WX_WET_"):
            del os.environ[k]
    os.environ.update(env)
    h.set_option(pkg.engine.Handle.OPT_ROW_BANDS, 1)


def timed(steps=100):
    h.step(40)  # new shape, clocks
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


res = {s: [] for s in specs}
for rep in range(REPS):
    for s in specs:
        apply(env_of(s))
        res[s].append(timed())
for s in specs:
    print(f"{X}x{Y} halo {HALO}  kernel " + " ".join(f"{k:.4f}" for _, k in res[s]) + "  step " + " ".join(f"{d:.4f}" for d, _ in res[s]) + f"  {s}")
