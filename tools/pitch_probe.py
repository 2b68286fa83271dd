#!/usr/bin/env python3
"""Does the power-of-two row pitch of the metric's grid (16384 columns x 16 B = 256 KiB per row) cost the marching kernels anything?
Whole-domain handles of neighbouring widths (same rows, same terrain generator, same seeded flow, each with its own placement search), timed
interleaved; reported per cell so that widths compare. Usage: python tools/pitch_probe.py [wet|dry] [Y] X1 X2 ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "wet"
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
XS = [int(v) for v in sys.argv[3:]] or [16384, 16408, 16440, 16328]
REPS = int(os.environ.get("REPS", "3"))
KEY = {"wet": "march_wet_full_iteration", "dry": "march_dry2_two_iterations_per_launch"}[kind]


def make(X):
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    if kind == "dry":
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    else:
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    if kind == "dry":
        h.upload(*pkg.synth.dry_grid(X, Y))
        h.set_option(h.OPT_DRY_PAIRS, 1)
    else:
        h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2)
    h.step(40)  # (the implicit placement search runs inside the first step)
    h.sync()
    return h


def timed(h, steps=200):
    h.profile(True)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(steps // 10):
        h.step(10)
    h.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    p = h.profile_read()
    h.profile(False)
    ms, n = p.get(KEY, (0.0, 1))
    return dt, ms / max(n, 1)


res = {}
for X in XS:  # one handle alive at a time (each keeps its placement candidates' memory only during the search)
    h = make(X)
    r = [timed(h) for _ in range(REPS)]
    info = h.placement_info() if hasattr(h, "placement_info") else None
    h.close()
    del h
    best = min(t[0] for t in r)
    res[X] = best
    print(f"{kind} {X}x{Y}: ms/iteration {' '.join(f'{t[0]:.4f}' for t in r)}  kernel/launch {' '.join(f'{t[1]:.4f}' for t in r)}"
          f"  -> {best * 1e6 / (X * Y):.4f} ns/cell-step = {X * Y / best / 1e6:.1f} Gcell-steps/s   placement {info}", flush=True)
