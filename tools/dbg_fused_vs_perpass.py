import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package()
E = pkg.engine
X, Y = int(sys.argv[1]), int(sys.argv[2])
base, water, wall = pkg.synth.terrain_grid(X, Y)
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 0
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
hs = []
for f in (1, 0):
    os.environ["WX_FUSED"] = str(f)
    h = E.Handle(X, Y, 0); h.upload(base, water, wall); h.set_params(p, u["initial_T"]); hs.append(h)
for it in range(1, 5):
    for h in hs: h.step(1)
    for fld in ("BASE_CUR", "WATER_CUR", "WATER_0", "WALL_CUR", "LIGHT_0", "LIGHT_1", "BASE_DISP", "CURL"):
        a, b = hs[0].read_rect(fld), hs[1].read_rect(fld)
        d = np.argwhere(a != b)
        if len(d):
            print(f"it {it} {fld}: {len(d)} diffs; first {d[:6].tolist()} x%64 {sorted(set((d[:,1]%64).tolist()))[:20]} y%16 {sorted(set((d[:,0]%16).tolist()))[:20]} ch {sorted(set(d[:,2].tolist()))}")
            y, x, c = d[0]
            print("   fused", a[y, x], "perpass", b[y, x], "wall", wall[y, x], "wall below", wall[y-1, x])
        else:
            print(f"it {it} {fld}: identical")
