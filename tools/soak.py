import sys, os, numpy as np, time
sys.path.insert(0, "/root/repo")
import wxpkg
pkg = wxpkg.load_package()
X, Y, N = 16384, 2048, 1048576
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 1
h = pkg.engine.Handle(X, Y, N)
h.setup_columns(pkg.synth.terrain_columns(X, Y, cloud_deck=True), pkg.synth.init_rain_drops(N))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
t0 = time.time()
for k in range(6):
    h.step(1000); h.sync()
    b = h.read_rect("BASE_CUR", 0, 0, X, 256); w = h.read_rect("WATER_CUR", 0, 512, X, 256); d = h.read_particles()
    print(f"iter {h.iter}: finite base {np.isfinite(b).all()} water {np.isfinite(w).all()} drops {np.isfinite(d).all()}; |v|max {np.abs(b[..., :2]).max():.3f}; active droplets {(d[:, 2] >= 0).sum()}; lightning {h.read_rect('LIGHTNING')}; {time.time() - t0:.1f} s", flush=True)
