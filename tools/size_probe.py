#!/usr/bin/env python3
"""Small and mid-size grids (the reference's own sizes): us per wet iteration of the marching kernel against the per-pass kernel set, frames
of 10 iterations. With the -DWX_DEBUG library WX_WET_MINROWS=<n> pins the minimum segment height (the default adapts it to the fill of the
chip: wet_launch_shape). Usage: python tools/size_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wxpkg
pkg = wxpkg.load_package()
gui = pkg.params.merge_settings(None)
for X, Y in ((100, 100), (256, 96), (512, 256), (1024, 512), (2048, 512), (4096, 1024), (2144, 2048)):
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    out = []
    for ks in (1, 0):
        h = pkg.engine.Handle(X, Y, 0)
        h.set_option(h.OPT_KERNEL_SET, ks)
        h.setup_terrain(pkg.synth.sounding_rows(Y), sim_height=float(gui["simHeight"]))
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
        n = 1000 if X * Y < 1e6 else 300
        h.step(300); h.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n // 10):
                h.step(10)
            h.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
        out.append(best)
        h.close()
    print(f"{X}x{Y}: marching {out[0]:.1f} us/iteration, per-pass {out[1]:.1f}   MINROWS={os.environ.get('WX_WET_MINROWS')}", flush=True)
