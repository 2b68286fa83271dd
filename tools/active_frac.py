import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import wxpkg
pkg = wxpkg.load_package()
X, Y, N = 16384, 2048, 1048576
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
uw = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); uw["enablePrecipitation"] = 1
h = pkg.engine.Handle(X, Y, N)
h.setup_columns(pkg.synth.terrain_columns(X, Y, cloud_deck=True), pkg.synth.init_rain_drops(N))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), uw), uw["initial_T"])
for n in (200, 200, 600, 1000, 1000):
    for _ in range(n // 10): h.step(10)
    d = h.read_particles()
    print(f"after {h.iter} iterations: active fraction {(d[:, 2] >= 0).mean():.3f}", flush=True)
