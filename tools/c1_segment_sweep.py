"""GPU box: BASELINE configs[1] (4096 x 1024 dry stencil) against the unit segment height of the dry marching kernel, ONE handle
(one placement), interleaved. Needs the debug build (make -C 2d-weather-sandbox_amd/csrc debug; WXSIM_LIB points at it): the
shipped library has one launch shape. Usage: c1_segment_sweep.py [X Y] -> table on stdout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WXSIM_LIB", os.path.join(ROOT, "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so"))
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 1024)
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
u["enablePrecipitation"] = 0
h = pkg.engine.Handle(X, Y, 0)
h.upload(*pkg.synth.dry_grid(X, Y))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
if os.environ.get("FLOW", "0") != "0":
    devtools.seed_flow(h, float(os.environ["FLOW"]))


def run(n=1000):
    for _ in range(30):
        h.step(10)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n // 10):
        h.step(10)
    h.sync()
    return (time.perf_counter() - t0) / n * 1e3


os.environ["WX_MARCH_DEBUG"] = "1"
print(f"{X}x{Y} dry stencil, frames of 10, 1000 iterations per entry; default rule first")
os.environ.pop("WX_MARCH_BAND_SEG", None)
for rep in range(2):
    ms = run()
    print(f"default rule: {ms * 1e3:7.2f} us / iteration  {X * Y / ms / 1e3:9.0f} Mcell-steps/s  frac {36 * X * Y / (ms * 1e-3) / 8e12:.3f}")
for rep in range(2):
    for R in os.environ.get("SEGS", "24 20 16 12 8 32").split():
        os.environ["WX_MARCH_BAND_SEG"] = R
        ms = run()
        print(f"band segment {R:>3} rows: {ms * 1e3:7.2f} us / iteration  {X * Y / ms / 1e3:9.0f} Mcell-steps/s  frac {36 * X * Y / (ms * 1e-3) / 8e12:.3f}", flush=True)
