"""GPU box (run under rocprofv3 --kernel-trace --stats): the dry pair kernel with sustained fast cells -- which kernel pays what.
Usage: dry_pair_trace.py [X Y] ; N=8 PEAK=1.3 RADIUS=20"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wxpkg
pkg = wxpkg.load_package()
from weather_sandbox_amd import devtools
X, Y = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 4096)
N, PEAK, RADIUS = int(os.environ.get("N", "8")), float(os.environ.get("PEAK", "1.3")), float(os.environ.get("RADIUS", "20"))
gui = pkg.params.merge_settings(None)
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
u["enablePrecipitation"] = 0
h = pkg.engine.Handle(X, Y, 0)
h.set_option(h.OPT_PLACEMENT_SEARCH, 0)
h.upload(*pkg.synth.dry_grid(X, Y))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
devtools.seed_flow(h, 0.2)
for k in range(3):
    devtools.seed_vortices(h, N, PEAK, RADIUS, seed=20 + k)
    for _ in range(10):
        h.step(10)
    print(h.pair_stats(), devtools.flow_stats(h), flush=True)
h.close()
