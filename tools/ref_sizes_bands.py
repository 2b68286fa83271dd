"""GPU box: the reference's own grid sizes (index.html:325, 335: 2500 x 300 default ... 16000 x 500) with the wet marching kernel's two
launch shapes -- column blocks per XCD (the default below 512 rows) against row bands per XCD (WX_OPT_ROW_BANDS 2) -- on ONE handle each,
interleaved; us per iteration, frames of 10, moving fluid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
gui = pkg.params.merge_settings(None)
gui["sunAngle"] = 50.0
for X, Y in ((2500, 300), (5000, 400), (8000, 500), (16000, 500), (16000, 300), (4096, 1024)):
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2)
    res = {1: [], 2: [], 0: []}
    for rep in range(3):
        for mode in (1, 2, 0):
            h.set_option(h.OPT_ROW_BANDS, mode)
            for _ in range(20):
                h.step(10)
            h.sync()
            t0 = time.perf_counter()
            for _ in range(50):
                h.step(10)
            h.sync()
            res[mode].append((time.perf_counter() - t0) / 500 * 1e6)
    print(f"{X}x{Y}: default rule {min(res[1]):.1f} us | row bands forced {min(res[2]):.1f} us | column blocks forced {min(res[0]):.1f} us   ({X * Y / min(min(res[1]), min(res[2])) / 1e3:.0f} Mcell-steps/s best)", flush=True)
    h.close()
