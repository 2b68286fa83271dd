#!/usr/bin/env python3
"""One small wet grid, 2000 iterations in frames of 10: wall time per iteration next to the kernel durations rocprofv3 sees (run under
`rocprofv3 --kernel-trace --stats`): how much of a small grid's iteration is launch overhead, how much the wave-serial march.
Usage: python tools/small_grid_trace.py [X Y]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wxpkg
pkg = wxpkg.load_package()
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 100)
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
u["enablePrecipitation"] = 0
h = pkg.engine.Handle(X, Y, 0)
h.setup_terrain(pkg.synth.sounding_rows(Y), sim_height=float(gui["simHeight"]))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
h.step(300); h.sync()
t0 = time.perf_counter()
for _ in range(200):
    h.step(10)
h.sync()
print(f"{X}x{Y}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per iteration (wall, frames of 10)")
h.close()
