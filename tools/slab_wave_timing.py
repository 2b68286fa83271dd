"""GPU box: per-segment start / end / duration of the marching wet kernel's waves (the -DWX_WET_TIMING variant dumps them at iteration 40)
on a slab or the whole grid. Usage: WXSIM_LIB=.../variants/libwxsim_timing.so python tools/slab_wave_timing.py X_owned Y halo"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package(); E = pkg.engine
from weather_sandbox_amd import devtools
XO, Y, HALO = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 0
if HALO:
    h = E.Handle(XO, Y, 0, X_global=XO, x0=0, halo=HALO)
    h.setup_columns(pkg.synth.terrain_columns(XO, Y, cols=(XO - HALO, XO + 2 * HALO)))
else:
    h = E.Handle(XO, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(XO, Y))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
devtools.seed_flow(h, 0.2, Xg=XO, x0=0)
if HALO:
    h.slab_set_vx_bound(1.0)
for _ in range(5):
    h.step(10)
h.sync()
