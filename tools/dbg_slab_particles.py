"""GPU-box debug: slab particles vs whole domain, mismatch statistics per exchange period."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg
pkg = wxpkg.load_package()
E = pkg.engine
nslab = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, Y, halo, n_iter, N = 512, 128, 64, 30, 6000
base, water, wall = pkg.synth.terrain_grid(X, Y)
pkg.synth.add_cloud_deck(water, wall)
rng = np.random.default_rng(4)
air = wall[..., 1] != 0
base[..., 0] += np.where(air, rng.normal(0, 0.3, (Y, X)), 0).astype(np.float32)
drops = pkg.synth.init_rain_drops(N)
na = 2500
px = rng.uniform(-1, 1, na)
xo = X // nslab
for k, e in enumerate(np.arange(nslab) * xo):
    px[k * 200:(k + 1) * 200] = (e + rng.uniform(-8, 8, 200)) / X * 2 - 1
px[1000:1200] = np.where(rng.random(200) < 0.5, -1 + rng.uniform(0, 7, 200) * 2 / X, 1 - rng.uniform(0, 7, 200) * 2 / X)
drops[:na, 0] = ((px + 1) % 2 - 1).astype(np.float32)
drops[:na, 1] = rng.uniform(-0.6, 0.2, na).astype(np.float32)
drops[:na, 2] = rng.uniform(0.1, 1.0, na).astype(np.float32)
drops[:na, 3] = np.where(rng.random(na) < 0.3, rng.uniform(0.1, 0.5, na), 0).astype(np.float32)
drops[:na, 4] = 1.0
gui = pkg.params.merge_settings(None)
gui["sunAngle"] = 35.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
u["enablePrecipitation"] = 1
u["inactiveDroplets"] = float(N - na)
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
whole = E.Handle(X, Y, N); whole.upload(base, water, wall, drops); whole.set_params(p, u["initial_T"])
slabs, bufs = [], []
for r in range(nslab):
    h = E.Handle(xo, Y, N, X_global=X, x0=r * xo, halo=halo); h.slab_set_rank(r)
    idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
    h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
    h.set_params(p, u["initial_T"]); slabs.append(h)
    bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
keys = [torch.zeros(N, dtype=torch.int32, device="cuda") for _ in range(nslab)]
state = [torch.zeros(5 * N, dtype=torch.float32, device="cuda") for _ in range(nslab)]
done = 0
while done < n_iter:
    k = min(per, n_iter - done)
    for h in slabs: h.step(k)
    whole.step(k)
    done += k
    for r, h in enumerate(slabs):
        h.halo_pack(0, bufs[r][0].data_ptr()); h.halo_pack(1, bufs[r][1].data_ptr())
    for h in slabs: h.sync()
    for r, h in enumerate(slabs):
        h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr()); h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
    for r, h in enumerate(slabs):
        h.particles_keys(keys[r].data_ptr()); h.sync()
    K = torch.stack(keys)
    win = K.max(0).values.contiguous()
    nclaim = (K == win[None]).sum(0)
    for r, h in enumerate(slabs):
        h.particles_contribute(win.data_ptr(), state[r].data_ptr()); h.sync()
    total = torch.stack(state).sum(0).contiguous()
    for h in slabs:
        h.particles_adopt(win.data_ptr(), total.data_ptr(), False); h.slab_period_begin(); h.sync()
    d_ref = whole.read_particles(); d = slabs[0].read_particles()
    bad = np.nonzero((d[:, 2] >= 0) != (d_ref[:, 2] >= 0))[0]
    diff = np.abs(d - d_ref).max(1)
    print(f"iter {done}: unclaimed {int((win == 0).sum())}, multi-claim {int((nclaim > 1).sum())}, active-mismatch {len(bad)}, |d|>1e-4: {(diff > 1e-4).sum()}, max {diff.max():.3g}")
    for i in bad[:6]:
        print("   ", i, "ref", d_ref[i], "slab", d[i], "x_ref_col", (d_ref[i, 0] + 1) / 2 * X, "x_slab_col", (d[i, 0] + 1) / 2 * X)
    big = np.nonzero(diff > 1e-4)[0]
    for i in big[:4]:
        print("  big", i, "ref", d_ref[i], "slab", d[i], "col", (d_ref[i, 0] + 1) / 2 * X)
    for f in ("PRECIP_FB", "BASE_CUR"):
        ref = whole.read_rect(f)
        for r, h in enumerate(slabs):
            a, b = h.read_rect(f, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]
            dd = np.abs(a - b).max(-1)
            if f == "PRECIP_FB" and r == 0: dd[0, :2] = 0
            yy, xx = np.unravel_index(dd.argmax(), dd.shape)
            print(f"    {f} slab {r}: max diff {dd.max():.3g} at local col {xx} row {yy}; cols with diff>1e-5: {np.nonzero((dd > 1e-5).any(0))[0][:12]}")
