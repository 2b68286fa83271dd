"""GPU box: the reference's own grid sizes (index.html:325, 335) against the minimum segment height of the marching wet kernel, with the
launch shape the shipped rule picks for them (row bands below 128 strips) and the other one; ONE handle per size (one placement),
interleaved, best of three. Needs the debug build (make -C 2d-weather-sandbox_amd/csrc debug): WX_WET_MINROWS pins the minimum there.
Usage: ref_sizes_minrows.py -> table on stdout (us per iteration, frames of 10, moving fluid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WXSIM_LIB", os.path.join(ROOT, "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so"))
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
gui = pkg.params.merge_settings(None)
gui["sunAngle"] = 50.0
ROWS = [int(r) for r in os.environ.get("ROWS", "0 2 3 4 5 6 8 10 13 19").split()]  # 0 = the shipped rule
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "2500x300 5000x400 8000x500 16000x500 100x100").split()]
for X, Y in SIZES:
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2)
    res = {}
    for rep in range(3):
        for mode in (1, 2, 0):  # the shipped rule / row bands forced / column blocks forced
            for r in ROWS:
                if r:
                    os.environ["WX_WET_MINROWS"] = str(r)
                else:
                    os.environ.pop("WX_WET_MINROWS", None)
                h.set_option(h.OPT_ROW_BANDS, mode)  # (drops the cached launch shape)
                for _ in range(10):
                    h.step(10)
                h.sync()
                t0 = time.perf_counter()
                for _ in range(40):
                    h.step(10)
                h.sync()
                us = (time.perf_counter() - t0) / 400 * 1e6
                res[(mode, r)] = min(res.get((mode, r), 1e9), us)
    os.environ.pop("WX_WET_MINROWS", None)
    print(f"{X}x{Y}  (us per iteration; minrows 0 = the shipped rule)   shipped: {res[(1, 0)]:.1f} us = {104 * X * Y / res[(1, 0)] / 8e6:.3f} of the roofline")
    print("  minrows        " + " ".join(f"{r:6d}" for r in ROWS))
    for mode, name in ((1, "default mode  "), (2, "row bands     "), (0, "column blocks ")):
        print("  " + name + " " + " ".join(f"{res[(mode, r)]:6.1f}" for r in ROWS))
    best = min(res, key=res.get)
    print(f"  best: mode {best[0]} minrows {best[1]}: {res[best]:.1f} us = {104 * X * Y / res[best] / 8e6:.3f}", flush=True)
    h.close()
