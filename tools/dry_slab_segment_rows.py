#!/usr/bin/env python3
"""Unit segment height of the two-iterations-per-launch dry kernel (WX_MARCH2_BAND_SEG of the -DWX_DEBUG library) on the north star's slab
(32768 x 4096 over 8 GPUs: 4096 + 2 x 42 columns) and on the whole grid: ONE handle per size, heights interleaved, ms per iteration.
Usage: WXSIM_LIB=.../libwxsim_debug.so python tools/dry_slab_segment_rows.py [X_owned Y halo]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package(); E = pkg.engine
from weather_sandbox_amd import devtools
XO = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
HALO = int(sys.argv[3]) if len(sys.argv) > 3 else 42
ROWS = [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,48,64,80,96,112,128".split(","))]
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY); u["enablePrecipitation"] = 0
h = E.Handle(XO, Y, 0, X_global=XO, x0=0, halo=HALO)
h.upload(*pkg.synth.dry_grid(XO, Y, cols=(XO - HALO, XO + 2 * HALO)))
h.slab_assert_water_free(True)
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
devtools.seed_flow(h, 0.2, Xg=XO, x0=0)
h.slab_set_vx_bound(1.0)
if int(os.environ.get("TUNE", "0")):
    print("placement:", h.tune_placement(int(os.environ["TUNE"])))


def timed(rows, n=300):
    if rows:
        os.environ["WX_MARCH2_BAND_SEG"] = str(rows)
    else:
        os.environ.pop("WX_MARCH2_BAND_SEG", None)
    h.step(20); h.sync()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / n * 1e3


res = {r: [] for r in ROWS}
for r in ROWS:
    timed(r, 40)
for _ in range(3):
    for r in ROWS:
        res[r].append(timed(r))
print(f"dry stencil, {XO} + 2 x {HALO} columns x {Y} rows, two iterations per launch: ms per iteration (frames of 10), 3 interleaved repetitions")
for r, v in res.items():
    print(f"  {('unit segment %d rows' % r) if r else '(shipped rule)':28s} " + " ".join(f"{x:.4f}" for x in v) + f"   median {np.median(v):.4f}")
