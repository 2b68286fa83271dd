"""GPU box: the dry north-star stencil one iteration per launch (k_march_dry) against two (k_march_dry2, WX_OPT_DRY_PAIRS) on ONE handle
(one placement), interleaved; with the debug build also the pair kernel's unit segment height (WX_MARCH2_BAND_SEG).
Usage: dry_pairs_ab.py [X Y] ; SEGS="24 32 48 64" REPS=2 TUNE=8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WXSIM_LIB", os.path.join(ROOT, "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so"))
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 4096)
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
u["enablePrecipitation"] = 0
h = pkg.engine.Handle(X, Y, 0)
h.upload(*pkg.synth.dry_grid(X, Y))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
if float(os.environ.get("FLOW", "0.2")) > 0:
    devtools.seed_flow(h, float(os.environ.get("FLOW", "0.2")))
if int(os.environ.get("TUNE", "8")):
    print("placement:", h.tune_placement(int(os.environ.get("TUNE", "8")), 30), flush=True)


def run(n=200):
    for _ in range(8):
        h.step(10)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / n * 1e3


print(f"{X}x{Y} dry stencil, frames of 10, 200 iterations per entry, moving fluid; {devtools.flow_stats(h)}")
for rep in range(int(os.environ.get("REPS", "2"))):
    h.set_option(h.OPT_DRY_PAIRS, 0)
    ms = run()
    print(f"one iteration per launch           : {ms:.4f} ms / iteration  {X * Y / ms / 1e3:9.0f} Mcell-steps/s  frac(36 B/cell-step) {36 * X * Y / (ms * 1e-3) / 8e12:.3f}", flush=True)
    h.set_option(h.OPT_DRY_PAIRS, 1)
    for R in os.environ.get("SEGS", "24 32 48 64").split():
        os.environ["WX_MARCH2_BAND_SEG"] = R
        ms = run()
        print(f"two per launch, {R:>3}-row segments   : {ms:.4f} ms / iteration  {X * Y / ms / 1e3:9.0f} Mcell-steps/s  frac(36 B/cell-step) {36 * X * Y / (ms * 1e-3) / 8e12:.3f}", flush=True)
h.sync()
print("flow at the end:", devtools.flow_stats(h))
