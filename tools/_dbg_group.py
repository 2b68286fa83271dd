import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import fuzz_parity as F
import wxpkg
pkg = wxpkg.load_package(); E = pkg.engine
E.lib().wx_set_option(None, E.Handle.OPT_PLACEMENT_SEARCH, 0)
seed, K = int(sys.argv[1]), int(sys.argv[2])
over = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
rng = np.random.default_rng(seed)
for k in range(K + 1):
    c = F.draw_group(rng, F.draw_case(rng, 600000))
c.update(over)
print(json.dumps(c))
X, Y = c["X"], c["Y"]
base, water, wall, u, drops = F.build_case(pkg, c)
nd = len(drops)
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
g = E.Group(c["nslab"], X, Y, halo=c["halo"], devices=[0] * c["nslab"], transport=E.TRANSPORT_LOCAL, n_droplets=nd)
whole = E.Handle(X, Y, nd)
g.upload(base, water, wall, drops); whole.upload(base, water, wall, drops)
g.set_params(p, u["initial_T"]); whole.set_params(p, u["initial_T"])
for hh in g.slabs + [whole]: hh.iter = c["iter0"]
for k, v in [(E.Handle.OPT_DRY_PAIRS, c["pairs"]), (E.Handle.OPT_ROW_BANDS, c["bands"]), (E.Handle.OPT_SPLAT_ORDER, 1)]:
    g.set_option(k, v); whole.set_option(k, v)
g.set_option(E.Handle.OPT_EXCHANGE_OVERLAP, c["overlap"]); g.set_option(E.Handle.OPT_SPLIT_LAUNCH, c["split"]); g.set_option(E.Handle.OPT_POOL_EXACT, 1)
for it in range(sum(c["steps"])):
    g.step(1); whole.step(1)
    g.exchange(); g.sync()
    d, r = g.particles(), whole.read_particles()
    ne = np.nonzero((d != r).any(1))[0]
    msg = []
    for f in ("BASE_CUR", "WATER_CUR", "PRECIP_FB", "PRECIP_DEP"):
        a, b = g.read(f), whole.read_rect(f)
        if f == "PRECIP_FB": a[0, :2] = 0; b[0, :2] = 0
        if not np.array_equal(a, b):
            ys, xs = np.nonzero((a != b).any(-1)); msg.append((f, len(ys), int(xs[0]), int(ys[0])))
    print("iter", it + 1, "droplets differing", len(ne), ne[:6].tolist(), msg, "active", int((r[:, 2] >= 0).sum()))
    for i in ne[:4]:
        print("   ", i, "group", d[i].tolist(), "whole", r[i].tolist(), "x_px", (r[i, 0] + 1) * 0.5 * X, "y_px", (r[i, 1] + 1) * 0.5 * Y)
    if len(ne) or msg:
        for i in ne[:2]:
            xp, yp = int((r[i, 0] + 1) * 0.5 * X) % X, min(Y - 1, int((r[i, 1] + 1) * 0.5 * Y))
            for f in ("WATER_CUR", "BASE_CUR", "WALL_CUR", "BASE_DISP", "PRECIP_FB", "PRECIP_DEP"):
                a, b = g.read(f), whole.read_rect(f)
                print("      ", f, "at", (xp, yp), "group", a[yp, xp].tolist(), "whole", b[yp, xp].tolist(), "| one row up", b[min(Y - 1, yp + 1), xp].tolist())
            print("       lightning group", [h.lightning().tolist() for h in g.slabs][:2], "whole", whole.lightning().tolist())
            print("       slab geometry: xo", g.xo, "halo", g.halo, "owner slab", xp // g.xo)
        fl = np.stack([h.pool_flags() for h in g.slabs])
        for i in ne[:4]: print("    flags", i, fl[:, i].tolist())
        break
