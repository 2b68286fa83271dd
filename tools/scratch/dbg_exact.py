import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import wxpkg
pkg = wxpkg.load_package()
E = pkg.engine
from test_gpu_parity import _pool_exact_iteration, _exact_period_end
nslab = 4
X, Y, halo, N = 512, 128, 64, 8000
base, water, wall = pkg.synth.terrain_grid(X, Y)
pkg.synth.add_cloud_deck(water, wall)
rng = np.random.default_rng(11)
air = wall[..., 1] != 0
base[..., 0] += np.where(air, rng.normal(0, 0.3, (Y, X)), 0).astype(np.float32)
deck = air & (water[..., 1] > 0)
water[..., 1] += np.where(deck, 2.5, 0).astype(np.float32)
water[..., 0] += np.where(deck, 2.5, 0).astype(np.float32)
base[..., 3] -= np.where(deck, 25.0, 0).astype(np.float32)
drops = pkg.synth.init_rain_drops(N)
na = 1500
drops[:na, 0] = rng.uniform(-1, 1, na).astype(np.float32)
drops[:na, 1] = rng.uniform(-0.8, 0.3, na).astype(np.float32)
drops[:na, 2] = rng.uniform(0.03, 0.2, na).astype(np.float32)
drops[:na, 3] = 0
drops[:na, 4] = 1.0
gui = pkg.params.merge_settings(None); gui["sunAngle"] = 35.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
u["enablePrecipitation"] = 1; u["inactiveDroplets"] = float(N - na); u["spawnChanceMult"] = 2e-3
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
xo, per, it0 = X // nslab, 1 + (halo - 12) // 9, 585
whole = E.Handle(X, Y, N); whole.upload(base, water, wall, drops); whole.set_params(p, u["initial_T"]); whole.set_option(1, 1); whole.iter = it0
slabs, bufs = [], []
for r in range(nslab):
    h = E.Handle(xo, Y, N, X_global=X, x0=r * xo, halo=halo); h.slab_set_rank(r)
    idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
    h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
    h.set_params(p, u["initial_T"]); h.set_option(1, 1); h.set_option(7, 1); h.iter = it0
    slabs.append(h); bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
ev = [torch.zeros(h.pool_event_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
pl = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
pr = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
prev = drops.copy()
for period in range(6):
    for j in range(per):
        _pool_exact_iteration(slabs, nslab, ev)
        whole.step(1)
        dref = whole.read_particles()
        d = [h.read_particles() for h in slabs]
        f = np.stack([h.pool_flags() for h in slabs])
        owners = (f == 2).sum(0)
        act_ref = dref[:, 2] >= 0
        bad = np.nonzero((owners != act_ref.astype(int)))[0]
        # owned records must equal the reference
        wrong = [i for i in np.nonzero(owners == 1)[0] if not np.array_equal(d[int(np.argmax(f[:, i] == 2))][i], dref[i])]
        if len(bad) or len(wrong):
            print("period", period, "iteration", j, "owner-count mismatches", len(bad), "wrong owned records", len(wrong))
            for i in list(bad[:4]) + wrong[:4]:
                gx = (dref[i, 0] / 2 + 0.5) * X
                print(" droplet", i, "flags", f[:, i], "ref", dref[i], "ref col", gx, "prev ref", prev[i], "prev col", (prev[i, 0] / 2 + 0.5) * X)
                for r in range(nslab):
                    print("   rank", r, d[r][i])
            sys.exit(0)
        if period == 0 or period == 3:
            for fld in ("BASE_CUR", "WATER_CUR", "BASE_DISP", "PRECIP_FB"):
                hb, wb = slabs[0].read_rect(fld), whole.read_rect(fld)
                idx = (0 * xo - halo + np.arange(xo + 2 * halo)) % X
                if fld == "PRECIP_FB":
                    wb = wb.copy(); wb[0, :2] = 0; hb = hb.copy(); hb[0, halo:halo + 2] = 0
                eq = np.nonzero(np.abs(hb - wb[:, idx]).max(axis=(0, 2)) == 0)[0]
                print("  period", period, "j", j, fld, "rank 0 equal columns", (int(eq.min()), int(eq.max())) if len(eq) else None, "claimed valid", (6 * (j + 1), xo + 2 * halo - 1 - 6 * (j + 1)))
        for r in range(nslab):
            ph = np.nonzero((f[r] >= 2) & (d[r][:, 2] >= 0) & (dref[:, 2] < 0))[0]
            if len(ph):
                b = whole.read_rect("BASE_CUR")
                print("PHANTOM period", period, "iteration j", j, "iter", whole.iter, "rank", r, "droplets", ph[:5], "flags", f[:, ph[0]], "rec", d[r][ph[0]], "col", (d[r][ph[0], 0] / 2 + 0.5) * X,
                      "max|v|", float(np.abs(b[..., :2]).max()))
                hb, wb = slabs[r].read_rect("BASE_CUR"), whole.read_rect("BASE_CUR")
                hw, ww = slabs[r].read_rect("WATER_CUR"), whole.read_rect("WATER_CUR")
                idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
                db = np.abs(hb - wb[:, idx]).max(axis=(0, 2)); dw = np.abs(hw - ww[:, idx]).max(axis=(0, 2))
                print("  local columns where rank's BASE differs from the whole domain:", np.nonzero(db > 0)[0][[0, -1]] if (db > 0).any() else None, " WATER:", np.nonzero(dw > 0)[0][[0, -1]] if (dw > 0).any() else None, "valid lo/hi", halo - (halo - 6 * (j + 1)), xo + halo + (halo - 6 * (j + 1)))
                sys.exit(0)
        prev = dref.copy()
    _exact_period_end(slabs, nslab, bufs, pl, pr)
    d = [h.read_particles() for h in slabs]
    f = np.stack([h.pool_flags() for h in slabs])
    owners = (f == 2).sum(0)
    off = np.nonzero((owners == 0) & ((f != 1).any(0)))[0]
    print("period", period, "after edges: offenders", len(off))
    for i in off[:6]:
        print(" droplet", i, "flags", f[:, i], "ref", dref[i], "ref col", (dref[i, 0] / 2 + 0.5) * X)
        for r in range(nslab):
            print("   rank", r, d[r][i])
    if len(off):
        sys.exit(0)
