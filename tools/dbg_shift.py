import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package()
E = pkg.engine
X, Y, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
base, water, wall = pkg.synth.terrain_grid(X, Y)
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 0
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
k = 80 * 37
res = {}
for f in (1, 0):
    os.environ["WX_FUSED"] = str(f)
    for sh in (0, k):
        h = E.Handle(X, Y, 0); h.upload(np.roll(base, sh, 1), np.roll(water, sh, 1), np.roll(wall, sh, 1)); h.set_params(p, u["initial_T"])
        h.step(n)
        res[(f, sh)] = (h.read_rect("BASE_CUR"), h.read_rect("WATER_CUR"), h.read_rect("WALL_CUR"))
        h.close()
for i, nm in enumerate(("BASE", "WATER", "WALL")):
    print(nm, "fused vs perpass (unshifted):", np.count_nonzero(res[(1, 0)][i] != res[(0, 0)][i]),
          " fused vs perpass (shifted):", np.count_nonzero(res[(1, k)][i] != res[(0, k)][i]))
    for f in (1, 0):
        a, b = np.roll(res[(f, 0)][i], k, 1), res[(f, k)][i]
        d = np.argwhere(a != b)
        print("   mode", f, "shift-equivariance diffs:", len(d), d[:5].tolist())
        if len(d):
            y, x, c = d[0]; print("     ", a[y, x], b[y, x], "maxabs", np.abs(a.astype(np.float64) - b).max())
print("max |v|", np.abs(res[(0,0)][0][..., :2]).max())
