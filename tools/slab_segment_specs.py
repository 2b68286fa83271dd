#!/usr/bin/env python3
"""Segment shapes of the marching wet kernel on the metric's SLAB (16384 x 2048 over 8 GPUs: 2048 + 2 x 42 columns x 2048 rows = 39 strips,
8 row bands of 256 rows): ONE handle, the shapes interleaved (WX_WET_SPEC of the -DWX_DEBUG library, "count x weight, ..." per band in
dispatch order), ms per plain iteration. A workgroup is four strips: 10 workgroups per segment and XCD, 32 CUs per XCD, at most 4
workgroups per CU -- the number of segments decides how evenly the CUs are loaded. Usage: WXSIM_LIB=.../libwxsim_debug.so python tools/slab_segment_specs.py [X_owned Y halo]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package(); E = pkg.engine
from weather_sandbox_amd import devtools
XO = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
HALO = int(sys.argv[3]) if len(sys.argv) > 3 else 42
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 0
h = E.Handle(XO, Y, 0, X_global=XO, x0=0, halo=HALO)
h.setup_columns(pkg.synth.terrain_columns(XO, Y, cols=(XO - HALO, XO + 2 * HALO)))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
devtools.seed_flow(h, 0.2, Xg=XO, x0=0)
h.slab_set_vx_bound(1.0)
SPECS = sys.argv[4].split(";") if len(sys.argv) > 4 else ["", "8x1", "9x1", "10x1", "11x1", "12x1", "7x1,1x0.5,1x0.25,1x0.125", "8x1,1x0.5,1x0.25", "9x1,1x0.5,1x0.25,1x0.125",
                                                          "10x1,2x0.5", "11x1,1x0.5", "6x1,2x0.5,2x0.25", "12x1,1x0.5,1x0.25"]


def timed(spec, n=300):
    if spec:
        os.environ["WX_WET_SPEC"] = spec
    else:
        os.environ.pop("WX_WET_SPEC", None)
    h.set_option(h.OPT_ROW_BANDS, 1)  # (invalidates the cached launch shape)
    h.step(20); h.sync()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / n * 1e3


if int(os.environ.get("TUNE", "0")):  # (the optimum may depend on the placement level: search first, like bench.py does)
    print("placement:", h.tune_placement(int(os.environ["TUNE"])))
res = {s: [] for s in SPECS}
for s in SPECS:
    timed(s, 40)
for _ in range(3):
    for s in SPECS:
        res[s].append(timed(s))
print(f"wet slab {XO} + 2 x {HALO} columns x {Y} rows: ms per iteration (frames of 10), 3 interleaved repetitions")
for s, v in res.items():
    print(f"  {s or '(shipped rule)':34s} " + " ".join(f"{x:.4f}" for x in v) + f"   median {np.median(v):.4f}")
