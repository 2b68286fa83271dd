"""GPU box: what a handful of FAST cells (|v| >= 0.9 cells / iteration) costs the dry pair kernel (round-5 verdict, item 1).
The same state -- eddies of sigma 0.2 + N seeded vortices of PEAK cells / iteration -- through (a) one iteration per launch, (b) pairs with
the cell-granular exact path (k_dry2_fix, round 6), (c) pairs whose every recorded cell repeats the WHOLE pair (WX_MARCH2_FIX_CAP=0 in the
debug build: round 5's behaviour). Usage: dry_pair_cliff.py [X Y] ; N=8 PEAK=1.3 RADIUS=6 TUNE=6"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WXSIM_LIB", os.path.join(ROOT, "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so"))
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 4096)
N, PEAK, RADIUS = int(os.environ.get("N", "8")), float(os.environ.get("PEAK", "1.3")), float(os.environ.get("RADIUS", "20"))
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
u["enablePrecipitation"] = 0


GRID = pkg.synth.dry_grid(X, Y)


def handle(vortices, cap=None):
    if cap is None:
        os.environ.pop("WX_MARCH2_FIX_CAP", None)
    else:
        os.environ["WX_MARCH2_FIX_CAP"] = str(cap)
    h = pkg.engine.Handle(X, Y, 0)
    h.upload(*GRID)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2)
    if vortices:
        devtools.seed_vortices(h, N, PEAK, RADIUS)
    return h


def run(h, n=100):
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / n * 1e3


print(f"{X}x{Y} dry stencil, frames of 10, eddies sigma 0.2 (+ {N} vortices of {PEAK} cells / iteration, radius {RADIUS})", flush=True)
for label, vort, pairs, cap in (("clean flow, one iteration per launch", False, 0, None), ("clean flow, pairs", False, 1, None),
                                ("fast cells, one iteration per launch", True, 0, None), ("fast cells, pairs + k_dry2_fix (round 6)", True, 1, None),
                                ("fast cells, pairs, every recorded cell repeats the whole pair (round 5)", True, 1, 0)):
    h = handle(vort, cap)
    h.set_option(h.OPT_DRY_PAIRS, pairs)
    if int(os.environ.get("TUNE", "6")):
        h.tune_placement(int(os.environ.get("TUNE", "6")), 20)
    for _ in range(6):
        h.step(10)
    h.pair_stats()
    ms = []
    for k in range(3):  # (fresh vortices before every run: they decay)
        if vort and k:
            devtools.seed_vortices(h, N, PEAK, RADIUS, seed=20 + k)
        if k == 0:
            f0 = devtools.flow_stats(h)
        ms.append(run(h))
    fixed, repeated = h.pair_stats()
    f1 = devtools.flow_stats(h)
    print(f"{label:<76}: {min(ms):.4f} .. {max(ms):.4f} ms / iteration  {X * Y / max(ms) / 1e3:9.0f} Mcell-steps/s | 300 iterations: cells recomputed {fixed}, pairs repeated whole {repeated}"
          f" | cells >= 0.9: {f0['cells_component_ge_0.9']} -> {f1['cells_component_ge_0.9']}, max |v| {f0['max_v']:.2f} -> {f1['max_v']:.2f}", flush=True)
    h.close()
