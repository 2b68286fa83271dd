"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle and the golden vectors.

Bars: every grid field BIT-EXACT against the oracle (the grid passes use only exactly-rounded fp32
operations in a fixed order, see csrc/wx_cells.h); particle feedback to fp32 summation-order tolerance;
against the SwiftShader goldens the calibrated envelope of tests/test_oracle_golden.py (the goldens were
rendered with SwiftShader's own fragCoord interpolation, which the product does not imitate).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ULP_T = 3.0518e-05
GRID_FIELDS = ["BASE_CUR", "BASE_DISP", "WATER_0", "WATER_CUR", "WALL_CUR", "WALL_DISP", "LIGHT_0", "LIGHT_1"]


@pytest.fixture(scope="module")
def E(pkg):
    from weather_sandbox_amd import engine
    engine.build()
    return engine


@pytest.fixture(params=[2, 0], ids=["march", "perpass"])
def fused(request):
    """Both kernel sets -- the row-marching single kernel (default) and one kernel per reference pass, the independent
    cross-check -- go through every parity test: WX_FUSED is read by wx_create."""
    import os
    old = os.environ.get("WX_FUSED")
    os.environ["WX_FUSED"] = str(request.param)
    yield request.param
    if old is None:
        os.environ.pop("WX_FUSED", None)
    else:
        os.environ["WX_FUSED"] = old


def _make_pair(pkg, oracle, E, X, Y, base, water, wall, u, drops=None, iter0=0):
    nd = 0 if drops is None else len(drops)
    h = E.Handle(X, Y, nd)
    h.upload(base, water, wall, drops)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    h.set_params(p, u["initial_T"], u.get("sounding_T"), u.get("sounding_W"), u.get("sounding_Vel"))
    h.iter = iter0
    o = oracle.OracleSim(X, Y, nd)
    o.upload(base, water, wall, drops)
    o.set_params(u)
    o.iter = iter0
    return h, o


def _assert_grid_equal(h, o, fields=GRID_FIELDS, emitted=True):
    for f in fields:
        a, b = h.read_rect(f), o.field(f)
        assert np.array_equal(a, b), f"{f}: {np.count_nonzero(a != b)} of {a.size} values differ, max |d| = {np.abs(a.astype(np.float64) - b).max()}"
    if emitted and (o.pass_mask & 32):  # the lighting pass's second render target (RGBA16F): the oracle's fp32 values, rounded once
        a, b = h.read_rect("EMITTED"), o.field("EMITTED").astype(np.float16)
        assert a.dtype == np.float16 and np.array_equal(a, b), f"EMITTED: {np.count_nonzero(a != b)} of {a.size} values differ"


@pytest.mark.parametrize("name", ["save100qa", "synth64", "randwalls64", "sounding64", "save100raw", "randwalls64p", "emitted64_day", "emitted64_night"])
@pytest.mark.parametrize("quad_scale", [0, 1])
def test_bit_exact_vs_oracle_on_golden_inputs(pkg, oracle, golden, E, name, quad_scale, fused):
    g, u = golden(name)
    u = dict(u, quad_scale=quad_scale, enablePrecipitation=0)
    X, Y = int(g["X"]), int(g["Y"])
    h, o = _make_pair(pkg, oracle, E, X, Y, g["in_base"], g["in_water"], g["in_wall"], u, iter0=int(g["iter0"]))
    done = 0
    for it in (1, 2, 10, 50):
        h.step(it - done)
        o.step(it - done)
        done = it
        _assert_grid_equal(h, o)
    assert h.iter == o.iter == int(g["iter0"]) + 50


def test_vs_swiftshader_goldens(pkg, golden, E, fused):
    """HIP (quad_scale=1) straight against the reference's own output, 50 iterations of the save."""
    g, u = golden("save100qa")
    u = dict(u, quad_scale=1, enablePrecipitation=0)
    X, Y = int(g["X"]), int(g["Y"])
    h = E.Handle(X, Y, 0)
    h.upload(g["in_base"], g["in_water"], g["in_wall"])
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    done = 0
    # tolerance: SwiftShader interpolates fragCoord differently (a few 1e-6 cells), plus pow() ulps
    tol = {1: (5e-7, 4 * ULP_T, 1e-4), 10: (5e-6, 2e-3, 5e-4), 50: (3e-5, 5e-3, 2e-3)}
    for it in (1, 10, 50):
        h.step(it - done)
        done = it
        tv, tT, tw = tol[it]
        assert np.array_equal(h.read_rect("WALL_CUR"), g[f"it{it}_wall_cur"]), "wall masks must be bit-exact"
        b, rb = h.read_rect("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :3] - rb[..., :3]).max() <= tv
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= tT
        assert np.abs(h.read_rect("WATER_CUR") - g[f"it{it}_water_cur"]).max() <= tw


def test_reference_raw_save_1000_iterations(pkg, golden, E, fused):
    """BASELINE configs[0]: the reference's UNMODIFIED `100 X 100 Test` save through the HIP path (quad_scale = 1: the
    reference's own fragment coordinates), 1000 iterations, straight against the reference's output (fixture save100raw, drawn
    as GL_POINTS). Wall / cell-type masks bit-exact at every dump; v, P, T, water inside the calibrated 1-ulp perturbation
    envelope (tests/golden/envelope_save100raw.json, oracle/golden/calibrate_envelope.py)."""
    import json, os
    g, u = golden("save100raw")
    with open(os.path.join(os.path.dirname(__file__), "golden", "envelope_save100raw.json")) as f:
        env = {int(k): v for k, v in json.load(f)["envelope"].items()}
    u = dict(u, quad_scale=1, enablePrecipitation=0)
    X, Y = int(g["X"]), int(g["Y"])
    h = E.Handle(X, Y, 0)
    h.upload(g["in_base"], g["in_water"], g["in_wall"])
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    # the envelope is what two EXACT runs differ by when their inputs differ in the last bit: the parity build must be inside it; the opt-in
    # tolerance build (tests/test_fast_arith.py runs this test on it), whose every operation may round differently, inside a small multiple
    factor = float(os.environ.get("WX_TEST_ENVELOPE_FACTOR", "1")) if E.lib().wx_arith() == 1 else 1.0
    done = 0
    for it in (1, 10, 50, 200, 1000):
        h.step(it - done)
        done = it
        e = {k: v * factor for k, v in env[it].items()}
        assert np.array_equal(h.read_rect("WALL_CUR"), g[f"it{it}_wall_cur"]), f"wall masks must be bit-exact (iteration {it})"
        b, rb = h.read_rect("BASE_CUR"), g[f"it{it}_base_cur"]
        assert np.abs(b[..., :2] - rb[..., :2]).max() <= e["v"], it
        assert np.abs(b[..., 2] - rb[..., 2]).max() <= e["P"], it
        assert np.abs(b[..., 3] - rb[..., 3]).max() <= e["T"], it
        assert np.abs(h.read_rect("WATER_CUR") - g[f"it{it}_water_cur"]).max() <= e["water"], it
    assert h.iter == 1000
    h.close()


def test_lightning_vs_reference(pkg, golden, E):
    """Lightning (precipitationShader.vert:121-140, lightningLocationShader.frag:24-38) through the HIP path against the
    reference, iteration by iteration on the reference's own inputs (fixture lightning64; pass_mask = precipitation only, the
    post-advection textures uploaded as the state): the rejected double strike, the accepted single strike, the 30-iteration
    lock-out and the multi-strikes after it. The strike decision hashes temperature bits, hence no free-running comparison."""
    g, u = golden("lightning64")
    X, Y, n = int(g["X"]), int(g["Y"]), len(g["in_drops"])
    iter0, niter = int(g["iter0"]), int(g["niter"])
    u = dict(u, quad_scale=1, enablePrecipitation=1, pass_mask=pkg.params.PASS_PRECIPITATION)
    h = E.Handle(X, Y, n)
    # several strike requests of one iteration are SUMMED in texel (1,0): with fp32 atomics the order -- and the last bit -- varies from
    # run to run; the deterministic splat order adds them in droplet-index order, the order the reference's blend unit draws them in
    h.set_option(h.OPT_SPLAT_ORDER, 1)
    drops, light = g["in_drops"].copy(), np.zeros(4, np.float32)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    flipped = accepted = 0
    for k in range(1, niter + 1):
        h.upload(g[f"it{k}_base_disp"], g[f"it{k}_water_cur"], g["in_wall"], drops)
        h.set_params(p, u["initial_T"])
        h.set_lightning(light)
        h.iter = iter0 + k - 1
        h.step(1)
        d, rd, rfb, rl = h.read_particles(), g[f"it{k}_drops"], g[f"it{k}_precip_fb"], g[f"it{k}_lightning"]
        flip = np.abs(d - rd).max(1) > 2.5e-7  # spawn threshold fract(pow(cloud * 10, 2)): one driver-pow ulp flips ~1 droplet in 1000
        # (the opt-in tolerance build -- tests/test_fast_arith.py runs this test on it -- rounds twice as many operations differently: 4
        # per iteration; the run's total stays below one droplet-iteration in 1000 for both builds, asserted at the end)
        assert flip.sum() <= (4 if E.lib().wx_arith() == 1 else 2), (k, int(flip.sum()))
        flipped += int(flip.sum())
        assert np.array_equal((d[:, 2] >= 0)[~flip], (rd[:, 2] >= 0)[~flip]), k
        fb = h.read_rect("PRECIP_FB", 0, 0, 2, 1)
        assert abs(fb[0, 0, 0] - rfb[0, 0, 0]) <= flip.sum(), k
        if not flip.any():  # (1,0) is the SUM of that iteration's strike requests: exact for one, atomics-order rounding for several
            assert np.abs(fb[0, 1] - rfb[0, 1]).max() <= 2e-7 * max(1.0, np.abs(rfb[0, 1]).max()), (k, fb[0, 1], rfb[0, 1])
            if rfb[0, 1, 2] <= iter0 + k - 1:
                assert np.array_equal(fb[0, 1], rfb[0, 1]), (k, fb[0, 1], rfb[0, 1])
        assert np.array_equal(h.read_rect("LIGHTNING"), rl), (k, h.read_rect("LIGHTNING"), rl)
        accepted += int(rl[2] != light[2])
        drops, light = rd.copy(), rl.copy()
    assert accepted >= 1 and flipped <= 0.001 * n * niter
    h.close()


@pytest.mark.parametrize("X,Y", [(512, 128), (192, 96), (130, 50), (4100, 20)])
def test_bit_exact_terrain_grid(pkg, oracle, E, X, Y, fused):
    """setupShader-style terrain with default settings, all grid passes + lighting; ragged sizes included."""
    S = pkg.synth
    base, water, wall = S.terrain_grid(X, Y)
    rng = np.random.default_rng(5)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.05, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.05, (Y, X)), 0).astype(np.float32)
    water[..., 0] *= np.where(air, 1.0 + 0.4 * rng.random((Y, X)), 1.0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 40.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u, iter0=90)
    for _ in range(3):
        h.step(7)
        o.step(7)
        _assert_grid_equal(h, o)


@pytest.mark.parametrize("X,Y", [(20, 12), (63, 17), (71, 24), (72, 23), (127, 15), (2, 4)])
def test_bit_exact_tiny_and_ragged_grids(pkg, oracle, E, fused, X, Y):
    """Grids smaller than a tile, one column past a tile, halo wrapping more than once (the SMALL kernel variants,
    every lane of a strip wraps): all passes and the dry mask, every kernel set."""
    base, water, wall = pkg.synth.terrain_grid(X, Y) if Y >= 12 else pkg.synth.dry_grid(X, Y)
    rng = np.random.default_rng(X * 100 + Y)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 40.0
    for mask in (pkg.params.PASS_ALL, pkg.params.PASS_DRY):
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=mask)
        u["enablePrecipitation"] = 0
        h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
        h.step(9)
        o.step(9)
        _assert_grid_equal(h, o)
        h.close()


@pytest.mark.parametrize("moist", [False, True], ids=["dry", "moist"])
@pytest.mark.parametrize("X,Y", [(256, 128), (130, 50)])
def test_dry_config_pass_mask(pkg, oracle, E, fused, X, Y, moist):
    """BASELINE config 1 (pass_mask = velocity|advection|pressure): the single fused dry kernel (with and without a
    water texture to carry) and the per-pass kernels, bit-exact vs the oracle; fast velocities exercise the
    out-of-tile back-trace path."""
    base, water, wall = pkg.synth.dry_grid(X, Y)
    base[Y // 3:Y // 2, X // 3:X // 2, 3] += 3.0  # a warm bubble so something moves
    rng = np.random.default_rng(3)
    base[..., 2] += rng.normal(0, 1e-3, (Y, X)).astype(np.float32)
    base[1:, :, 0] += rng.normal(0, 0.35, (Y - 1, X)).astype(np.float32)  # some |v| > 0.9
    if moist:
        water[1:, :, 0] = 3.0 + rng.random((Y - 1, X)).astype(np.float32)
        water[1:, :, 3] = 0.1
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, pass_mask=pkg.params.PASS_DRY)
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    for _ in range(3):
        h.step(7)
        o.step(7)
        _assert_grid_equal(h, o, ["BASE_CUR", "BASE_DISP", "WATER_CUR", "WATER_0", "WALL_CUR"])
    assert np.abs(h.read_rect("BASE_CUR")[..., 0]).max() > 1e-4


@pytest.mark.parametrize("X,Y", [(256, 128), (130, 50), (64, 8), (1000, 70), (4100, 20)])
@pytest.mark.parametrize("pairs", [0, 1])
def test_dry_marching_kernel(pkg, oracle, E, X, Y, monkeypatch, pairs):
    """The row-marching wavefront kernel (wx_march.h, WX_DRY_MARCH=1) on the water-free dry state: bit-exact vs the
    oracle, strips and row segments that do not divide the grid, |v| > 0.9 through the out-of-line path. pairs = WX_OPT_DRY_PAIRS: 1
    (the default since round 5) runs two iterations per launch (wx_march2.h) -- with these velocities nearly every pair's second
    iteration meets a back-trace it has no exact path for and is repeated by the predicated one-iteration launches: still the oracle,
    bit for bit."""
    monkeypatch.setenv("WX_DRY_MARCH", "1")
    monkeypatch.setenv("WX_FUSED", "1")
    base, water, wall = pkg.synth.dry_grid(X, Y)
    base[Y // 3:Y // 2, X // 3:X // 2, 3] += 3.0
    rng = np.random.default_rng(11)
    base[..., 2] += rng.normal(0, 1e-3, (Y, X)).astype(np.float32)
    base[1:, :, 0] += rng.normal(0, 0.35, (Y - 1, X)).astype(np.float32)
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, pass_mask=pkg.params.PASS_DRY)
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    h.set_option(E.Handle.OPT_DRY_PAIRS, pairs)
    h.profile(True)
    for _ in range(3):
        h.step(7)
        o.step(7)
        _assert_grid_equal(h, o, ["BASE_CUR", "BASE_DISP", "WATER_CUR", "WATER_0", "WALL_CUR"])
    prof = h.profile_read()
    if pairs and Y >= 16:  # three pairs and the odd one out per call
        assert prof["march_dry2_two_iterations_per_launch"][1] == 9 and prof["march_dry_vel_advect_pressure"][1] == 3, prof
    else:
        assert prof["march_dry_vel_advect_pressure"][1] == 21, prof  # the marching kernel is what ran


@pytest.mark.parametrize("pairs", [0, 1])
@pytest.mark.parametrize("X,Y", [(700, 160), (130, 96)])
def test_dry_marching_kernel_with_obstacles(pkg, oracle, E, X, Y, monkeypatch, pairs):
    """Wall blocks inside the dry domain: the marching dry kernel switches between its free-air instantiation of the advection stage
    (rows whose footprints reach no wall cell: one vote per input row, three rows of history) and the wall-aware one, several times
    per segment, next to strip borders and the periodic seam. Bit-exact vs the oracle."""
    monkeypatch.setenv("WX_DRY_MARCH", "1")
    monkeypatch.setenv("WX_FUSED", "1")
    base, water, wall = pkg.synth.dry_grid(X, Y)
    rng = np.random.default_rng(23)
    for x0, x1, y0, y1 in [(10, 40, 20, 22), (55, 70, 30, 47), (X - 6, X, 50, 58), (0, 5, 50, 58), (X // 2, X // 2 + 1, 70, 71),
                           (X // 3, X // 3 + 90, Y - 30, Y - 29), (120 % X, 125 % X, 3, 4)]:
        wall[y0:y1, x0:x1, 1] = 0  # distance 0 = wall
        wall[y0:y1, x0:x1, 0] = 0  # inert type
    base[..., 2] += rng.normal(0, 1e-3, (Y, X)).astype(np.float32)
    base[1:, :, 0] += rng.normal(0, 0.3, (Y - 1, X)).astype(np.float32)
    base[1:, :, 1] += rng.normal(0, 0.2, (Y - 1, X)).astype(np.float32)
    base[wall[..., 1] == 0, 0:2] = 0.0
    water[wall[..., 1] == 0, 0] = 1001.0  # the marker the advection pass gives wall cells (advectionShader.frag:403-409): "water-free" state
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, pass_mask=pkg.params.PASS_DRY)
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    h.set_option(E.Handle.OPT_DRY_PAIRS, pairs)
    h.profile(True)
    for _ in range(3):
        h.step(5)
        o.step(5)
        _assert_grid_equal(h, o, ["BASE_CUR", "BASE_DISP", "WATER_CUR", "WALL_CUR"])
    prof = h.profile_read()
    if pairs:  # two pairs and the odd one out per call
        assert prof["march_dry2_two_iterations_per_launch"][1] == 6 and prof["march_dry_vel_advect_pressure"][1] == 3, prof
    else:
        assert prof["march_dry_vel_advect_pressure"][1] == 15, prof  # the marching kernel is what ran
    h.close()


def test_dry_water_free_flag_is_dropped_when_water_appears(pkg, oracle, E, monkeypatch):
    """The water-free specialisation of the dry iteration (NO_WATER marching / tiled kernel) is only valid while the water
    texture is identically zero in air. A moisture brush during a dry step, or a full step in between, puts water there:
    the following brush-free dry steps must advect and condense it like the oracle does."""
    monkeypatch.setenv("WX_FUSED", "1")
    X, Y = 256, 96
    base, water, wall = pkg.synth.dry_grid(X, Y)
    rng = np.random.default_rng(13)
    base[1:, :, 0] += rng.normal(0, 0.1, (Y - 1, X)).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    for scenario in ("brush", "full_step"):
        u_dry = pkg.params.uniforms_from_gui(gui, Y, pass_mask=pkg.params.PASS_DRY)
        h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u_dry)
        h.profile(True)
        h.step(2)
        o.step(2)
        assert h.profile_read()["march_dry2_two_iterations_per_launch"][1] == 1  # water-free so far: the marching kernel ran (one pair)
        if scenario == "brush":
            u2 = dict(u_dry, userInputType=2, userInputValues=(0.5, 0.5, 0.8, 20.0))  # TOOL_WATER: adds vapour (and cloud)
        else:
            u2 = pkg.params.uniforms_from_gui(gui, Y, pass_mask=pkg.params.PASS_ALL)
            u2["enablePrecipitation"] = 0
        for sim_u in (u2, u_dry):  # two steps with the change, then brush-free dry steps again
            h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), sim_u), sim_u["initial_T"])
            o.set_params(sim_u)
            h.step(3)
            o.step(3)
            _assert_grid_equal(h, o, ["BASE_CUR", "WATER_CUR", "WALL_CUR"])
        if scenario == "brush":
            assert h.read_rect("WATER_CUR")[Y // 2, X // 2, 0] > 0.1
        names = set(h.profile_read())
        assert "march_dry_vel_advect_pressure" not in names and "march_dry2_two_iterations_per_launch" not in names  # the water-carrying kernels took over
        h.close()


def test_more_to_come_pieces_equal_one_step(pkg, oracle, E):
    """(ABI 10, WX_OVERLAP_MORE_TO_COME) A step cut into pieces -- what a slab host does at every exchange period -- whose pieces but the
    last skip the display-side stores: state AND display-side fields after the last piece equal one undivided step and the oracle."""
    X, Y = 700, 160
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.Generator(np.random.Philox(8))
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 40.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    for pieces in ((3, 2), (1, 1, 4), (6,)):
        for i, k in enumerate(pieces):
            h.step(k, 4 if i + 1 < len(pieces) else 0)
        o.step(sum(pieces))
        _assert_grid_equal(h, o)
        assert np.array_equal(h.read_rect("CURL"), o.field("CURL"))
    h.close()


def test_water0_on_demand_equals_stored(pkg, oracle, E):
    """(ABI 10, WX_OPT_WATER0_ON_DEMAND) waterTexture_0 -- what a save stores, no display pass samples it -- is made when asked for: the
    per-pass kernels on the inputs the step's last iteration left behind, with THAT iteration's parameters. Against a handle whose
    iterations store it themselves (option 0) and against the oracle: after odd and even iteration counts, with the light textures
    switched to RGBA by a reader in between, after wx_set_params replaced the parameters AND the initial_T row, across a placement
    search, and twice in a row."""
    X, Y = 700, 160
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.Generator(np.random.Philox(5))
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 30.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    gui2 = dict(gui, sunAngle=70.0, dryLapseRate=9.0)
    u2 = pkg.params.uniforms_from_gui(gui2, Y, quad_scale=0)
    u2["enablePrecipitation"] = 0
    assert not np.array_equal(np.asarray(u["initial_T"]), np.asarray(u2["initial_T"]))
    lazy, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    stored = E.Handle(X, Y, 0)
    stored.set_option(stored.OPT_WATER0_ON_DEMAND, 0)
    stored.upload(base, water, wall)
    stored.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    lazy.profile(True)
    cur = u
    for k, how in ((3, "plain"), (2, "light first"), (5, "params"), (4, "tune"), (1, "twice")):
        for h in (lazy, stored):
            h.step(k)
        o.step(k)
        if how == "light first":  # a reader switches the light textures to RGBA before anybody asks for waterTexture_0
            assert np.array_equal(lazy.read_rect("LIGHT_0"), o.field("LIGHT_0"))
        if how == "params":  # the NEXT frame's parameters arrive before the save asks: waterTexture_0 belongs to the iteration that ran
            cur = u2 if cur is u else u
            for h in (lazy, stored):
                h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), cur), cur["initial_T"])
            o.set_params(cur)
        if how == "tune":
            lazy.tune_placement(2, 3)
        w0 = lazy.read_rect("WATER_0")
        assert np.array_equal(w0, stored.read_rect("WATER_0")), how
        assert np.array_equal(w0, o.field("WATER_0")), how
        if how == "twice":
            assert np.array_equal(lazy.read_rect("WATER_0"), w0)
        _assert_grid_equal(lazy, o)
    names = lazy.profile_read()
    assert names["boundary"][1] >= 4, names  # the on-demand passes ran (once per question, not per frame)
    lazy.close()
    stored.close()


@pytest.mark.parametrize("X,Y,bands", [(700, 200, None), (1100, 260, "2"), (130, 50, None)])
def test_fast_cells_exact_path_vs_oracle(pkg, oracle, E, monkeypatch, X, Y, bands):
    """Cells whose back-trace leaves the 3 x 3 neighbourhood (|v| >= 0.9 cells / iteration) go through the fix pass of the marching
    wet kernel (k_wet_fix: an 8 x 8 post-boundary patch per output cell): an updraft core of contiguous fast cells (up to 1.8),
    scattered single ones, cores that straddle strip / segment / band borders and the periodic seam, next to terrain -- and a few
    cells beyond 2 cells / iteration, whose footprints leave the patch (fully general fallback). Bit-exact against the oracle."""
    monkeypatch.setenv("WX_FUSED", "2")
    if bands:
        monkeypatch.setenv("WX_WET_BANDS", bands)
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.Generator(np.random.Philox(91))
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.15, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    yy, xx = np.mgrid[0:Y, 0:X]
    for cx, cy, rx, ry, vx, vy in [(50, Y // 2, 9, 14, 0.3, 1.6), (56 * 3 + 2, Y // 3, 7, 7, -1.4, 0.8), (X - 3, Y // 2, 8, 6, 1.2, -1.3),
                                   (X // 2, Y - 20, 12, 10, 0.9, 1.1), (X // 3, 30, 6, 20, -1.0, 1.7)]:
        blob = np.exp(-(((xx - cx + X // 2) % X - X // 2) / rx) ** 2 - ((yy - cy) / ry) ** 2)
        base[..., 0] += np.where(air, vx * blob, 0).astype(np.float32)
        base[..., 1] += np.where(air, vy * blob, 0).astype(np.float32)
    for k in range(12):  # single cells far beyond the patch's reach
        x, y = int(rng.integers(0, X)), int(rng.integers(Y // 2, Y - 2))
        base[y, x, :2] = (2.6, -2.2) if k % 2 else (-3.1, 2.4)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 30.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    assert (np.abs(base[..., :2]).max(-1) >= 0.9).sum() > 300
    assert h.fastest_velocity() == 0.0  # nothing ran yet
    for n in (1, 2, 4):
        h.step(n)
        o.step(n)
        _assert_grid_equal(h, o)
        if n == 1:  # wx_fastest_velocity: the planted fast cells went through the exact path (post-boundary values: the pressure step has already taken the edge off the 3.1-cell spikes)
            assert 1.2 < h.fastest_velocity() < 4.5
            assert h.fastest_velocity() == 0.0  # ... and reading it resets it
    h.sync()
    h.close()
    calm, _, _ = pkg.synth.terrain_grid(X, Y)
    c = E.Handle(X, Y, 0)
    c.upload(calm, water, wall)
    c.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    c.step(5)
    assert c.fastest_velocity() == 0.0  # a fluid at rest never enters the exact path
    c.close()


def test_fix_pass_sized_for_an_empty_list_meets_fast_cells(pkg, oracle, E, monkeypatch):
    """The fix pass is launched with a few workgroups while the last list the host has heard of was empty (WetFixList::hint). A calm
    state first (the hint drops to 0), then a state full of fast cells uploaded into the SAME handle: the small launch has to work
    through a long list -- any grid is correct (grid-stride loop). Bit-exact against the oracle."""
    monkeypatch.setenv("WX_FUSED", "2")
    X, Y = 700, 200
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 30.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u)
    h.step(3)  # nothing moves: empty lists, and the host reads the hint at the next enqueue
    h.sync()
    rng = np.random.Generator(np.random.Philox(17))
    air = wall[..., 1] != 0
    base = base.copy()
    base[..., 0] += np.where(air, rng.normal(0, 0.33, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.33, (Y, X)), 0).astype(np.float32)
    assert (np.abs(base[..., :2]).max(-1) >= 0.9).sum() > 1000  # (a few times that many list entries; the list holds 65 536)
    h.upload(base, water, wall)
    o.upload(base, water, wall)
    h.iter = o.iter = 0
    for n in (1, 2):
        h.step(n)
        o.step(n)
        _assert_grid_equal(h, o)
    h.sync()
    h.close()


def test_exact_path_overflow_is_reported(pkg, E, monkeypatch):
    """More fast cells in one iteration than the exact-path list holds: WX_E_STATE from the next blocking call, not silence."""
    monkeypatch.setenv("WX_FUSED", "2")
    monkeypatch.setenv("WX_WET_FIX_CAP", "64")
    X, Y = 256, 96
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    base[40:80, :, 1] = 1.5
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = E.Handle(X, Y, 0)
    h.upload(base, water, wall)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(1)
    with pytest.raises(E.WxError) as ei:
        h.sync()
    assert ei.value.code == -5 and "0.9" in str(ei.value)
    h.close()


def test_exact_path_overflow_never_consumes_unwritten_list_slots(pkg, E):
    """An appender whose 1-3 entries straddle the end of the exact-path list writes none of them, so the slots just below the capacity may
    hold whatever the allocation held before; the fix pass must not take coordinates from there (tools/fuzz_parity.py found a memory access
    fault a few cases after a handle whose list had overflowed). Device memory is poisoned first; capacities that 1-3-entry appends straddle."""
    import torch
    X, Y = 384, 128
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    base[40:110, :, 1] = 1.5
    base[40:110, :, 0] = -1.2
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    for cap in (1000, 1001, 1002, 4099):
        poison = [torch.full((1 << 20,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(16)]
        torch.cuda.synchronize()
        del poison
        torch.cuda.empty_cache()  # (back to the driver: the handle's hipMalloc may now be handed these pages)
        h = E.Handle(X, Y, 0)
        h.set_option(h.OPT_KERNEL_SET, 1)
        h.set_option(h.OPT_FIX_CAP, cap)
        h.upload(base, water, wall)
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
        h.step(2)
        with pytest.raises(E.WxError) as ei:
            h.sync()
        assert ei.value.code == -5
        h.close()
    t = torch.zeros(16, device="cuda") + 1.0  # the device is still there
    assert float(t.sum()) == 16.0


def test_particles_vs_oracle(pkg, oracle, golden, E, fused):
    g, u = golden("precip64")
    u = dict(u, quad_scale=0, enablePrecipitation=1)
    X, Y = int(g["X"]), int(g["Y"])
    h, o = _make_pair(pkg, oracle, E, X, Y, g["in_base"], g["in_water"], g["in_wall"], u, drops=g["in_drops"])
    h.step(1)
    o.step(1)
    # particle state: per-droplet arithmetic is order independent -> bit-exact
    assert np.array_equal(h.read_particles(), o.field("DROPS"))
    # splats: atomic adds in arbitrary order vs droplet-index order
    fb, ofb = h.read_rect("PRECIP_FB"), o.field("PRECIP_FB")
    assert np.abs(fb - ofb).max() <= 1e-7 * max(1.0, np.abs(ofb).max())
    assert round(float(fb[0, 0, 0])) == round(float(ofb[0, 0, 0])) > 10  # inactive-droplet count in texel (0,0)
    assert np.abs(h.read_rect("PRECIP_DEP") - o.field("PRECIP_DEP")).max() <= 1e-7
    assert np.array_equal(h.read_rect("LIGHTNING"), o.field("LIGHTNING"))
    # grid fields of that iteration did not depend on feedback yet -> still bit-exact
    _assert_grid_equal(h, o)
    h.step(3)
    o.step(3)
    d, od = h.read_particles(), o.field("DROPS")
    assert np.array_equal(d[:, 2] >= 0, od[:, 2] >= 0)
    assert np.abs(d - od).max() <= 1e-5  # feedback summation order feeds back into T, cloud
    assert np.abs(h.read_rect("BASE_CUR")[..., 3] - o.field("BASE_CUR")[..., 3]).max() <= 1e-3


def test_particles_deterministic_order_coupled_run_bit_exact(pkg, oracle, golden, E, fused):
    """WX_OPT_SPLAT_ORDER 1 (deposit records sorted by anchor texel, each texel's deposits added in droplet-index order, order-free
    box trees) against the oracle's matching summation tree (splat_order 1): the COUPLED run -- droplets feed the grid through the
    feedback textures, the grid feeds the droplets -- stays bit for bit equal, iteration after iteration."""
    g, u = golden("precip64")
    u = dict(u, quad_scale=0, enablePrecipitation=1, splat_order=1)
    X, Y = int(g["X"]), int(g["Y"])
    h, o = _make_pair(pkg, oracle, E, X, Y, g["in_base"], g["in_water"], g["in_wall"], u, drops=g["in_drops"])
    h.set_option(h.OPT_SPLAT_ORDER, 1)
    for n in (1, 3, 8):
        h.step(n)
        o.step(n)
        assert np.array_equal(h.read_particles(), o.field("DROPS"))
        assert np.array_equal(h.read_rect("PRECIP_FB"), o.field("PRECIP_FB"))
        assert np.array_equal(h.read_rect("PRECIP_DEP"), o.field("PRECIP_DEP"))
        assert np.array_equal(h.read_rect("LIGHTNING"), o.field("LIGHTNING"))
        _assert_grid_equal(h, o)
    assert np.abs(h.read_rect("PRECIP_FB")).max() > 0 and np.abs(h.read_rect("PRECIP_DEP")).max() > 0
    # two runs of the deterministic mode are identical; the atomic mode agrees with it to summation order
    h2 = E.Handle(X, Y, len(g["in_drops"]))
    h2.upload(g["in_base"], g["in_water"], g["in_wall"], g["in_drops"])
    h2.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h2.step(12)
    assert np.abs(h2.read_particles() - h.read_particles()).max() <= 1e-5
    with pytest.raises(E.WxError):
        h.set_option(99, 1)
    h.set_option(h.OPT_CHECK_LAUNCHES, 1)  # synchronise-and-check after every launch: same results
    h.step(1)
    o.step(1)
    assert np.array_equal(h.read_particles(), o.field("DROPS"))
    h.close()
    h2.close()


@pytest.mark.parametrize("mode", ["wet", "dry", "particles"])
def test_tune_placement_leaves_the_state_untouched(pkg, oracle, E, mode):
    """wx_tune_placement re-allocates the handle's planes (several candidate arenas, each timed with the handle's own iteration on a
    COPY of the state) and moves in with the fastest: every field, the droplet pool and the iteration counter are what they were, and
    the run continues bit for bit like an untuned handle / the oracle."""
    X, Y = 700, 200
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.Generator(np.random.Philox(17))
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    drops = None
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 40.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, **({"pass_mask": pkg.params.PASS_DRY} if mode == "dry" else {}))
    u["enablePrecipitation"] = 0
    if mode == "dry":
        base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.2)
    if mode == "particles":
        pkg.synth.add_cloud_deck(water, wall)
        drops = pkg.synth.init_rain_drops(3000)
        drops[:1000, 0] = rng.uniform(-1, 1, 1000).astype(np.float32)
        drops[:1000, 1] = rng.uniform(-0.6, 0.2, 1000).astype(np.float32)
        drops[:1000, 2] = rng.uniform(0.1, 1.0, 1000).astype(np.float32)
        drops[:1000, 3] = 0.0
        drops[:1000, 4] = 1.0
        u.update(enablePrecipitation=1, inactiveDroplets=2000.0, splat_order=1)
    h, o = _make_pair(pkg, oracle, E, X, Y, base, water, wall, u, drops=drops, iter0=3)
    if mode == "particles":
        h.set_option(h.OPT_SPLAT_ORDER, 1)
    fields = ["BASE_CUR", "WATER_CUR", "WALL_CUR"] + ([] if mode == "dry" else ["LIGHT_0", "LIGHT_1", "BASE_DISP", "WATER_0"])
    h.step(3)
    o.step(3)
    before = {f: h.read_rect(f) for f in fields + ["CURL", "PRECIP_FB"]}
    ms0, ms1 = h.tune_placement(tries=3, iters_per_try=3)
    assert ms1 <= ms0 and ms1 > 0
    assert h.iter == 6
    for f, a in before.items():
        assert np.array_equal(h.read_rect(f), a), f
    if drops is not None:
        assert np.array_equal(h.read_particles(), o.field("DROPS"))
    _assert_grid_equal(h, o, fields, emitted=mode != "dry")
    h.step(5)
    o.step(5)
    _assert_grid_equal(h, o, fields, emitted=mode != "dry")
    if drops is not None:
        assert np.array_equal(h.read_particles(), o.field("DROPS"))
        assert np.array_equal(h.read_rect("PRECIP_FB"), o.field("PRECIP_FB"))
    h.close()


def test_read_rect_contract(pkg, golden, E):
    g, u = golden("synth64")
    X, Y = int(g["X"]), int(g["Y"])
    h = E.Handle(X, Y, 0)
    h.upload(g["in_base"], g["in_water"], g["in_wall"])
    assert np.array_equal(h.read_rect("BASE_CUR", 3, 5, 7, 2), g["in_base"][5:7, 3:10])
    assert np.array_equal(h.read_rect("BASE_DISP"), g["in_base"])  # upload fills both ping-pong copies
    assert np.array_equal(h.read_rect("WATER_0", 0, 0, X, 1), g["in_water"][:1])
    w8 = h.read_rect("WALL_DISP", 10, 0, 1, Y)
    w32 = h.read_rect("WALL_DISP", 10, 0, 1, Y, int32=True)
    assert w8.dtype == np.int8 and w32.dtype == np.int32 and np.array_equal(w8, w32)
    assert np.array_equal(w8[:, 0], g["in_wall"][:, 10])
    with pytest.raises(E.WxError) as ei:  # no wrap, like readPixels
        h.read_rect("BASE_CUR", X - 2, 0, 4, 1)
    assert ei.value.code == -4
    with pytest.raises(E.WxError) as ei:
        h.step(1)  # no params yet
    assert ei.value.code == -5
    with pytest.raises(E.WxError):
        h.read_particles(0, 1)


def test_full_size_invariants(pkg, E):
    """BASELINE size (16384 x 2048): the two independently written kernel sets (row-marching single kernel vs one kernel
    per reference pass) agree BIT FOR BIT, runs are deterministic, fields stay finite / non-negative, and the
    solution is periodic in x (shift equivariance)."""
    import os
    X, Y = 16384, 2048
    S = pkg.synth
    base, water, wall = S.terrain_grid(X, Y)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    fields = ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1", "WATER_0", "BASE_DISP", "CURL")

    def run(b, w, wl, n, fused):
        os.environ["WX_FUSED"] = str(fused)
        try:
            h = E.Handle(X, Y, 0)
        finally:
            os.environ.pop("WX_FUSED", None)
        h.upload(b, w, wl)
        h.set_params(p, u["initial_T"])
        h.step(n)
        out = {f: h.read_rect(f) for f in fields}
        h.close()
        return out

    n = 12
    r_f = run(base, water, wall, n, 2)
    for mode in (0,):
        r_p = run(base, water, wall, n, mode)
        for f in fields:
            assert np.array_equal(r_f[f], r_p[f]), f"single-kernel iteration and kernel set {mode} differ in {f}"
        del r_p
    b0, w0, wl0 = r_f["BASE_CUR"], r_f["WATER_CUR"], r_f["WALL_CUR"]
    assert np.isfinite(b0).all() and np.isfinite(w0).all() and np.isfinite(r_f["LIGHT_0"]).all()
    assert (w0[..., 0] >= 0).all() and (w0[..., 1] >= 0).all()
    assert np.abs(b0[..., :2]).max() > 1e-3  # the flow developed
    # determinism
    r_2 = run(base, water, wall, n, 2)
    assert all(np.array_equal(r_f[f], r_2[f]) for f in fields)
    del r_2
    # periodic in x: shifting the input by a multiple of 80 columns (industrial stacks use x % 80) shifts the
    # output. Masks and water are exactly equivariant; the back-trace `fragCoord - v` is evaluated at the
    # absolute x like in the reference, so its rounding depends on x (ulp(16384.5) = 1e-3 cell): velocity,
    # pressure and temperature are equivariant to rounding noise (99 % of the values bit-equal), not bit for bit.
    k = 80 * 37
    r_s = run(np.roll(base, k, 1), np.roll(water, k, 1), np.roll(wall, k, 1), n, 2)
    assert np.array_equal(np.roll(wl0, k, 1), r_s["WALL_CUR"])
    assert np.abs(np.roll(w0, k, 1) - r_s["WATER_CUR"]).max() <= 1e-5
    d = np.abs(np.roll(b0, k, 1) - r_s["BASE_CUR"])
    assert d[..., :3].max() <= 1e-4 and d[..., 3].max() <= 1e-3
    assert (d == 0).mean() > 0.99


def test_full_size_dry_marching_equals_tiled(pkg, E, monkeypatch):
    """BASELINE configs[1] at 16384 x 2048: the row-marching kernel and the LDS-tiled kernel (independent data paths:
    wave-private row ring + shuffles vs 64x16 tiles) agree bit for bit after 25 iterations."""
    X, Y = 16384, 2048
    base, water, wall = pkg.synth.dry_grid(X, Y)
    rng = np.random.default_rng(17)
    base[1:, :, 0] += rng.normal(0, 0.2, (Y - 1, X)).astype(np.float32)  # includes |v| > 0.9 cells (exact out-of-line path)
    base[1:, :, 2] += rng.normal(0, 1e-3, (Y - 1, X)).astype(np.float32)
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, pass_mask=pkg.params.PASS_DRY)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    out = {}
    for march in (1, 0):
        monkeypatch.setenv("WX_DRY_MARCH", str(march))
        h = E.Handle(X, Y, 0)
        h.upload(base, water, wall)
        h.set_params(p, u["initial_T"])
        h.profile(True)
        h.step(25)
        out[march] = {f: h.read_rect(f) for f in ("BASE_CUR", "BASE_DISP", "WALL_CUR")}
        assert ("march_dry_vel_advect_pressure" in h.profile_read()) == bool(march)
        h.close()
    for f in out[1]:
        assert np.array_equal(out[1][f], out[0][f]), f
    assert np.isfinite(out[1]["BASE_CUR"]).all() and np.abs(out[1]["BASE_CUR"][..., 1]).max() > 1e-4


def test_full_size_long_run_stays_physical(pkg, E):
    """600 iterations of the full wet iteration at 16384 x 2048 (day side, default sliders): no NaN/Inf, water
    non-negative, velocities inside the shaders' documented -1..1 cell/iteration range, air temperature physical,
    wall masks unchanged where nothing can build or erode them."""
    X, Y = 16384, 2048
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = E.Handle(X, Y, 0)
    h.upload(base, water, wall)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(600)
    b, w, wl = h.read_rect("BASE_CUR"), h.read_rect("WATER_CUR"), h.read_rect("WALL_CUR")
    assert h.iter == 600
    assert np.isfinite(b).all() and np.isfinite(w).all() and np.isfinite(h.read_rect("LIGHT_0")).all()
    air = wl[..., 1] != 0
    assert (w[..., 0][air] >= 0).all() and (w[..., 1][air] >= 0).all()
    assert np.abs(b[..., :2]).max() < 1.0
    realT = b[..., 3] - ((np.arange(Y, dtype=np.float32)[:, None] + 0.5) / Y) * np.float32(u["dryLapse"])
    assert realT[air].min() > 150.0 and realT[air].max() < 400.0
    assert np.array_equal(wl[..., 0], wall[..., 0])  # wall types: no brush, no fire -> unchanged
    h.close()


@pytest.mark.parametrize("nslab,halo", [(2, 12), (4, 6)])
def test_slab_handles_equal_whole_domain(pkg, E, fused, nslab, halo):
    """wx_create_slab + wx_halo_pack/unpack: N slab handles on one GPU, halos copied device-to-device in ring
    order, against the undecomposed handle -- bit for bit. (The multi-process RCCL path uses the same calls;
    its exchange logic is covered by tests/test_slab_gloo.py.)"""
    import torch
    X, Y, n_iter = 256, 64, 9
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.default_rng(2)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=1)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(p, u["initial_T"])
    xo = X // nslab
    slabs, bufs = [], []
    for r in range(nslab):
        h = E.Handle(xo, Y, 0, X_global=X, x0=r * xo, halo=halo)
        idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
        h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
        h.set_params(p, u["initial_T"])
        assert h.halo_bytes() == halo * Y * 68
        slabs.append(h)
        bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
    per = halo // 6
    done = 0
    while done < n_iter:
        k = min(per, n_iter - done)
        for h in slabs:
            h.step(k)
        done += k
        for r, h in enumerate(slabs):
            h.halo_pack(0, bufs[r][0].data_ptr())
            h.halo_pack(1, bufs[r][1].data_ptr())
        for h in slabs:
            h.sync()
        for r, h in enumerate(slabs):
            h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr())  # left ghosts <- left neighbour's right edge
            h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
        for h in slabs:
            h.sync()
    whole.step(n_iter)
    # (EMITTED is computed on demand from the last lighting pass's inputs incl. one ghost column per side: valid after the exchange)
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1", "EMITTED"):
        ref = whole.read_rect(f)
        for r, h in enumerate(slabs):
            assert np.array_equal(h.read_rect(f, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]), (f, r)


def _run_overlapped(E, slabs, comm, bufs, per, n_iter):
    """n_iter iterations on N slab handles of one GPU with the halo exchange on per-handle comm streams (device-to-device copies
    stand in for send / recv), edge / interior split launches, no host synchronisation between the steps."""
    import torch
    nslab = len(slabs)
    done, exchanged = 0, False
    while done < n_iter:
        k = min(per, n_iter - done)
        flags = (E.Handle.OVERLAP_EDGES_LAST if exchanged else 0) | (E.Handle.OVERLAP_EDGES_FIRST if k == per else 0)
        for h in slabs:
            h.step(k, flags)
        done += k
        if k < per:
            break
        packed = []
        for r, h in enumerate(slabs):  # pack on each handle's comm stream (the library waits for the edge-strip event only)
            h.halo_pack(0, bufs[r][0].data_ptr())
            h.halo_pack(1, bufs[r][1].data_ptr())
            ev = torch.cuda.Event()
            ev.record(comm[r])
            packed.append(ev)
        for r, h in enumerate(slabs):  # "recv": a handle's unpack may start once BOTH neighbours have packed
            comm[r].wait_event(packed[(r - 1) % nslab])
            comm[r].wait_event(packed[(r + 1) % nslab])
            h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr())
            h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
        # a neighbour's NEXT pack overwrites the buffers this handle's unpack reads: fence the comm streams among themselves
        unpacked = []
        for r in range(nslab):
            ev = torch.cuda.Event()
            ev.record(comm[r])
            unpacked.append(ev)
        for r in range(nslab):
            comm[r].wait_event(unpacked[(r - 1) % nslab])
            comm[r].wait_event(unpacked[(r + 1) % nslab])
        exchanged = True


@pytest.mark.parametrize("nslab,halo,X,bands,split", [(2, 12, 1024, None, 1), (4, 24, 4096, None, 1), (2, 6, 128, None, 1), (2, 12, 1024, "2", 1),
                                                      (4, 24, 4096, "2", 1), (2, 12, 1024, None, 0), (4, 24, 4096, "2", 0)])
def test_slab_overlapped_exchange_equals_whole_domain(pkg, E, monkeypatch, nslab, halo, X, bands, split):
    """The exchange / compute overlap (wx_set_comm_stream + wx_step_overlap): N slab handles on one GPU, each with its own compute
    stream and its own comm stream, nothing synchronised on the host between the steps -- the edge strips of the last iteration
    run first, pack + copy + unpack proceed on the comm streams while the interior strips compute, the edge strips of the next
    iteration wait for the unpack. Ten exchange periods, bit for bit the undecomposed handle. (A slab too narrow to have
    interior strips, the third case, degrades to the in-order exchange through the same calls.) `split` = WX_OPT_SPLIT_LAUNCH: 1 (the
    default since round 5) runs a split iteration as ONE launch whose dispatch order puts the edge strips first / last, with device-side
    hand-offs (arrival word + gate kernel, epoch word); 0 = the two launch groups on two streams of rounds 2-4. The velocities
    (sigma 0.2) put a few dozen cells per iteration on the exact path, also in edge strips: their own list, consumed on the comm stream."""
    import torch
    monkeypatch.setenv("WX_FUSED", "2")
    if bands:  # the row-band launch shape of wide slabs (e.g. two ranks on 32768 columns), forced onto this small grid
        monkeypatch.setenv("WX_WET_BANDS", bands)
    Y = 64
    per, n_iter = halo // 6, 10 * (halo // 6) + 1
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.default_rng(2)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=1)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    xo = X // nslab
    slabs, bufs, main, comm = [], [], [], []
    for r in range(nslab):
        h = E.Handle(xo, Y, 0, X_global=X, x0=r * xo, halo=halo)
        idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
        h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
        h.set_params(p, u["initial_T"])
        main.append(torch.cuda.Stream())
        comm.append(torch.cuda.Stream())
        h.set_stream(main[r].cuda_stream)
        h.set_comm_stream(comm[r].cuda_stream)
        h.set_option(E.Handle.OPT_SPLIT_LAUNCH, split)
        slabs.append(h)
        bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
    _run_overlapped(E, slabs, comm, bufs, per, n_iter)
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(p, u["initial_T"])
    whole.step(n_iter)
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1"):
        ref = whole.read_rect(f)
        for r, h in enumerate(slabs):
            assert np.array_equal(h.read_rect(f, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]), (f, r)
    for h in slabs:
        h.close()


def _dry_uniforms(pkg, Y):
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    u["enablePrecipitation"] = 0
    return u


def _dry_slabs(pkg, E, X, Y, nslab, halo, base, water, wall, u, assert_free):
    import torch
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    xo = X // nslab
    slabs, bufs, main, comm = [], [], [], []
    for r in range(nslab):
        h = E.Handle(xo, Y, 0, X_global=X, x0=r * xo, halo=halo)
        idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
        h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
        h.set_params(p, u["initial_T"])
        main.append(torch.cuda.Stream())
        comm.append(torch.cuda.Stream())
        h.set_stream(main[r].cuda_stream)
        h.set_comm_stream(comm[r].cuda_stream)
        slabs.append(h)
        bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
    free = [h.water_free() for h in slabs]
    agreed = all(free) if assert_free is None else assert_free  # (slab.py: all-reduce MIN over the ranks)
    for h in slabs:
        h.slab_assert_water_free(agreed)
    return slabs, comm, bufs, free


@pytest.mark.parametrize("nslab,halo,X,wet_rank,Y,split", [(2, 12, 1024, None, 96, 1), (4, 24, 4096, None, 96, 1), (2, 12, 1024, 1, 96, 1), (4, 12, 2048, 2, 96, 1),
                                                           (4, 24, 4096, None, 576, 1), (2, 12, 1024, None, 96, 0), (4, 24, 4096, None, 576, 0)])
def test_dry_slab_overlapped_exchange_equals_whole_domain(pkg, E, nslab, halo, X, wet_rank, Y, split):
    """BASELINE's north-star stencil (pass_mask DRY) on slabs with the exchange overlapped: the water-free row-marching kernel takes
    strip ranges like the wet one (edge strips / interior). It may only run when NO slab carries water -- the hosts agree on that
    once per upload (wx_water_free -> all-reduce MIN -> wx_slab_assert_water_free), so no step ever reads a flag back. With water in
    ONE rank's slab (wet_rank) every handle falls back to the water-carrying kernel; both ways the owned columns equal the
    undecomposed handle bit for bit over ten exchange periods. Y = 576 gives the row-band launch shape (72-row bands), whose split
    iterations are ONE launch with the edge strips first / last in every XCD's dispatch order (`split` = WX_OPT_SPLIT_LAUNCH, see the wet test)."""
    per, n_iter = halo // 6, 10 * (halo // 6) + 1
    base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.2 if Y < 200 else 0.12)  # (the taller grid's eddies are faster: |v| < 1 is the precondition of the fixed period)
    rng = np.random.Generator(np.random.Philox(77))
    base[1:, :, 2] += rng.normal(0, 1e-3, (Y - 1, X)).astype(np.float32)
    xo = X // nslab
    if wet_rank is not None:  # a humid blob inside ONE rank's slab (outside its neighbour's ghost columns), drifting towards the edge
        x0 = wet_rank * xo + halo + 2
        water[20:60, x0:x0 + 40, 0] = 0.004
        base[20:60, x0 - 20:x0 + 60, 0] = -0.7
    u = _dry_uniforms(pkg, Y)
    slabs, comm, bufs, free = _dry_slabs(pkg, E, X, Y, nslab, halo, base, water, wall, u, None)
    assert free == [r != wet_rank for r in range(nslab)]
    for h in slabs:
        h.set_option(E.Handle.OPT_SPLIT_LAUNCH, split)
        h.profile(True)
    _run_overlapped(E, slabs, comm, bufs, per, n_iter)
    for h in slabs:
        names = set(h.profile_read())
        assert ("march_dry_vel_advect_pressure" in names) == (wet_rank is None), names
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    whole.step(n_iter)
    assert np.abs(whole.read_rect("BASE_CUR")[..., :2]).max() < 1.0  # precondition of the 6-column cone
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR"):
        ref = whole.read_rect(f)
        for r, h in enumerate(slabs):
            assert np.array_equal(h.read_rect(f, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]), (f, r)
    for h in slabs:
        h.close()
    whole.close()


def test_dry_slab_wrong_water_free_assertion_is_reported(pkg, E):
    """A host that asserts "every slab is water-free" although one slab carries water gets WX_E_STATE from that slab's next wx_halo_pack
    (ABI 10: agreed slabs exchange the base texture alone, and the slab that knows better refuses to) -- never a silent divergence."""
    X, Y, nslab, halo = 1024, 64, 2, 12
    base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.05)
    x0 = X // 2 + halo + 1  # inside rank 1's slab, just outside rank 0's ghost columns, drifting left at 0.8 cells / iteration
    water[20:40, x0:x0 + 30, 0] = 0.004
    base[20:40, x0 - 20:x0 + 40, 0] = -0.8
    u = _dry_uniforms(pkg, Y)
    slabs, comm, bufs, free = _dry_slabs(pkg, E, X, Y, nslab, halo, base, water, wall, u, True)  # wrong on purpose
    assert free == [True, False]
    assert [h.halo_message_bytes() for h in slabs] == [halo * Y * 16] * 2  # (by the host's word: the two agree about the message size)
    with pytest.raises(E.WxError) as ei:
        _run_overlapped(E, slabs, comm, bufs, halo // 6, 3 * (halo // 6))
    assert ei.value.code == -5 and "THIS slab is not water-free" in str(ei.value)
    for h in slabs:
        h.close()


def _pool_exchange(slabs, nslab, ev, pl, pr):
    """The droplet-pool part of slab.SlabSim.exchange() with the all-gather and the send / recv spelled out (handles of one process)."""
    import torch
    for r, h in enumerate(slabs):
        h.pool_events_pack(ev[r].data_ptr())
        h.sync()
    most = max(int(e[:4].view(torch.int32)[0]) for e in ev)  # slab.py: the counts travel first, then only the filled part
    stride = min(len(ev[0]), (16 + most * 32 + 4095) // 4096 * 4096)
    gathered = torch.cat([e[:stride] for e in ev]).contiguous()
    for h in slabs:
        h.pool_events_apply(gathered.data_ptr(), nslab, stride)
    for r, h in enumerate(slabs):
        h.pool_edges_pack(pl[r].data_ptr(), pr[r].data_ptr(), False)
        h.sync()
    for r, h in enumerate(slabs):
        h.pool_edges_apply(pr[(r - 1) % nslab].data_ptr())  # my left neighbour's right-edge droplets
        h.pool_edges_apply(pl[(r + 1) % nslab].data_ptr())
        h.sync()
    best = max((h.lightning() for h in slabs), key=lambda v: float(v[2]))
    for h in slabs:
        h.set_lightning(best)
        h.slab_period_begin()


def _assemble_pool(slabs):
    """The global pool from the partitioned one: every active droplet is owned (flag 2) by exactly one handle, every other droplet is
    inactive with the same record on every handle."""
    d = [h.read_particles() for h in slabs]
    f = np.stack([h.pool_flags() for h in slabs])
    owners = (f == 2).sum(0)
    assert owners.max() <= 1, "an active droplet is owned by two ranks"
    inactive = owners == 0
    assert (f[:, inactive] == 1).all(), "a droplet nobody owns must be inactive on every rank"
    for k in range(1, len(slabs)):
        assert np.array_equal(d[k][inactive], d[0][inactive]), "inactive records differ between ranks"
    out = d[0].copy()
    for k, h in enumerate(slabs):
        out[f[k] == 2] = d[k][f[k] == 2]
    return out, f


@pytest.mark.parametrize("nslab,order", [(2, 0), (4, 0), (4, 1)])
def test_slab_particles_equal_whole_domain(pkg, E, nslab, order):
    """Particles on column slabs: the PARTITIONED droplet pool (owner = slab containing the droplet, ghost copies within `halo`
    columns of an edge, static inactive records everywhere, status-flip events at the exchange) -- N slab handles on one GPU, the
    all-gather and the neighbour send / recv of slab.py emulated with torch ops, against the undecomposed handle. Droplets sit on
    slab edges, in ghost zones and within a sprite radius of the domain edge (sprites are clipped there, never wrapped) and drift
    across the slab edges. With the deterministic splat order (order 1) the whole coupled run is BIT-IDENTICAL to the undecomposed
    one; with atomics the fields agree to summation order."""
    import torch
    X, Y, halo, n_iter, N = 512, 128, 64, 25, 6000
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    pkg.synth.add_cloud_deck(water, wall)
    rng = np.random.default_rng(4)
    air = wall[..., 1] != 0
    # droplets drift across slab edges (|vx| clipped below one cell / iteration: these tests drive wx_step / wx_halo_* with the FIXED periods of
    # the 6-column cone, which the handles now hold their hosts to -- tests/test_group_transport.py::test_group_is_exact_at_any_speed covers faster flow)
    base[..., 0] += np.where(air, np.clip(rng.normal(0, 0.3, (Y, X)), -0.8, 0.8), 0).astype(np.float32)
    drops = pkg.synth.init_rain_drops(N)
    na = 2500  # active droplets: everywhere, plus clusters on the slab edges and on the domain edge
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
    whole = E.Handle(X, Y, N)
    whole.upload(base, water, wall, drops)
    whole.set_params(p, u["initial_T"])
    whole.set_option(whole.OPT_SPLAT_ORDER, order)
    slabs, bufs = [], []
    for r in range(nslab):
        h = E.Handle(xo, Y, N, X_global=X, x0=r * xo, halo=halo)
        h.slab_set_rank(r)
        idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
        h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
        h.set_params(p, u["initial_T"])
        h.set_option(h.OPT_SPLAT_ORDER, order)
        assert h.halo_bytes() == halo * Y * 92  # + feedback 16 + deposition 8
        slabs.append(h)
        bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
    ev = [torch.zeros(h.pool_event_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
    pl = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
    pr = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
    f0 = np.stack([h.pool_flags() for h in slabs])  # after the upload: every active droplet has one owner, the far ranks dropped it
    assert ((f0 == 2).sum(0) == (drops[:, 2] >= 0)).all() and (f0 == 0).sum() > 0
    per = 1 + (halo - 12) // 9  # WX_SLAB_PERIOD_PARTICLES: 6 columns for the first iteration of a period, 9 for every further one
    with pytest.raises(E.WxError):  # more iterations than the ghost columns allow: refused before any iteration runs
        slabs[0].step(per + 1)
    assert slabs[0].iter == 0
    done = 0
    while done < n_iter:
        k = min(per, n_iter - done)
        for h in slabs:
            h.step(k)
        done += k
        for r, h in enumerate(slabs):
            h.halo_pack(0, bufs[r][0].data_ptr())
            h.halo_pack(1, bufs[r][1].data_ptr())
        for h in slabs:
            h.sync()
        for r, h in enumerate(slabs):
            h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr())
            h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
        _pool_exchange(slabs, nslab, ev, pl, pr)
    whole.step(n_iter)
    d_ref = whole.read_particles()
    assert (d_ref[:, 2] >= 0).sum() > 500 and (d_ref[:, 2] < 0).sum() > 500
    d, f = _assemble_pool(slabs)
    assert np.array_equal(d[:, 2] >= 0, d_ref[:, 2] >= 0), "same droplets active"
    assert (f == 3).sum() > 0, "ghost copies exist"
    if order == 1:
        assert np.array_equal(d, d_ref)
    else:
        assert np.abs(d - d_ref).max() <= 1e-6
    tol = {"PRECIP_FB": 1e-6, "PRECIP_DEP": 1e-6, "BASE_CUR": 1e-6, "WATER_CUR": 1e-6}
    for fld, t in tol.items():
        ref = whole.read_rect(fld)
        for r, h in enumerate(slabs):
            a, b = h.read_rect(fld, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]
            if fld == "PRECIP_FB" and r == 0:  # texels (0,0) / (1,0) are the reference's mailboxes, not kept on slabs
                a, b = a.copy(), b.copy()
                a[0, :2], b[0, :2] = 0, 0
            if order == 1:
                assert np.array_equal(a, b), (fld, r)
            else:
                assert np.abs(a - b).max() <= t * max(1.0, np.abs(b).max()), (fld, r, np.abs(a - b).max())
    assert np.abs(whole.read_rect("PRECIP_FB")).max() > 0
    for r, h in enumerate(slabs):
        assert np.array_equal(h.read_rect("WALL_CUR", halo, 0, xo, Y), whole.read_rect("WALL_CUR")[:, r * xo:(r + 1) * xo])
    for h in slabs:
        h.close()
    whole.close()


def _pool_exact_iteration(slabs, nslab, ev):
    """WX_OPT_POOL_EXACT: one iteration on every slab, then the per-iteration all-gather of status flips + iteration records."""
    import torch
    for h in slabs:
        h.step(1)
    for r, h in enumerate(slabs):
        h.pool_events_pack(ev[r].data_ptr())
        h.sync()
    most = max(int(e[:4].view(torch.int32)[0]) for e in ev)
    stride = min(len(ev[0]), (16 + most * 32 + 4095) // 4096 * 4096)
    gathered = torch.cat([e[:stride] for e in ev]).contiguous()
    for h in slabs:
        h.pool_events_apply(gathered.data_ptr(), nslab, stride)
    return most


def _exact_period_end(slabs, nslab, bufs, pl, pr):
    """... and once per period: grid halos + edge droplets (no events, no lightning reconciliation: both are current already)."""
    for r, h in enumerate(slabs):
        h.halo_pack(0, bufs[r][0].data_ptr())
        h.halo_pack(1, bufs[r][1].data_ptr())
    for h in slabs:
        h.sync()
    for r, h in enumerate(slabs):
        h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr())
        h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
    for r, h in enumerate(slabs):
        h.pool_edges_pack(pl[r].data_ptr(), pr[r].data_ptr(), False)
        h.sync()
    for r, h in enumerate(slabs):
        h.pool_edges_apply(pr[(r - 1) % nslab].data_ptr())
        h.pool_edges_apply(pl[(r + 1) % nslab].data_ptr())
        h.sync()
        h.slab_period_begin()


@pytest.mark.parametrize("nslab", [2, 4])
def test_slab_particles_exact_mode_is_bit_identical(pkg, E, nslab):
    """WX_OPT_POOL_EXACT (SURVEY 8e's determinism check with particles): status flips, lightning requests and the 600-iteration inactive
    count exchanged after EVERY iteration. A spawn-heavy scene -- a cold dense cloud deck, most of the pool inactive, spawn chance raised so
    that droplets spawn, retire and re-spawn within one exchange period, lightning strikes -- run across iteration 600 with the
    deterministic splat order: pool, feedback, deposition, lightning state and every grid field equal the undecomposed handle bit for bit
    after every period; the default (per-period) protocol on the same scene does not."""
    import torch
    X, Y, halo, N = 512, 128, 64, 8000
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    pkg.synth.add_cloud_deck(water, wall)
    rng = np.random.default_rng(11)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, np.clip(rng.normal(0, 0.3, (Y, X)), -0.8, 0.8), 0).astype(np.float32)  # (fixed 6-column periods: |vx| < 1)
    deck = air & (water[..., 1] > 0)
    water[..., 1] += np.where(deck, 2.5, 0).astype(np.float32)  # dense: lightning requests
    water[..., 0] += np.where(deck, 2.5, 0).astype(np.float32)
    base[..., 3] -= np.where(deck, 25.0, 0).astype(np.float32)   # cold: snow spawns
    drops = pkg.synth.init_rain_drops(N)
    na = 1500
    drops[:na, 0] = rng.uniform(-1, 1, na).astype(np.float32)
    drops[:na, 1] = rng.uniform(-0.8, 0.3, na).astype(np.float32)
    drops[:na, 2] = rng.uniform(0.03, 0.2, na).astype(np.float32)  # light: many evaporate (retire) soon
    drops[:na, 3] = 0
    drops[:na, 4] = 1.0
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1
    u["inactiveDroplets"] = float(N - na)
    u["spawnChanceMult"] = 2e-3
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    xo, per, it0 = X // nslab, 1 + (halo - 12) // 9, 585
    results = {}
    for exact in (1, 0):
        whole = E.Handle(X, Y, N)
        whole.upload(base, water, wall, drops)
        whole.set_params(p, u["initial_T"])
        whole.set_option(whole.OPT_SPLAT_ORDER, 1)
        whole.iter = it0
        slabs, bufs = [], []
        for r in range(nslab):
            h = E.Handle(xo, Y, N, X_global=X, x0=r * xo, halo=halo)
            h.slab_set_rank(r)
            idx = (r * xo - halo + np.arange(xo + 2 * halo)) % X
            h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
            h.set_params(p, u["initial_T"])
            h.set_option(h.OPT_SPLAT_ORDER, 1)
            h.set_option(h.OPT_POOL_EXACT, exact)
            h.iter = it0
            slabs.append(h)
            bufs.append([torch.empty(h.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in range(2)])
        ev = [torch.zeros(h.pool_event_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
        pl = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
        pr = [torch.zeros(h.pool_edge_bytes(), dtype=torch.uint8, device="cuda") for h in slabs]
        if exact:
            with pytest.raises(E.WxError):  # one iteration per call in exact mode
                slabs[0].step(2)
        identical, flips, strikes = True, 0, set()
        for period in range(6):  # 36 iterations: 585 .. 620, across the refresh of iteration 600
            if exact:
                for _ in range(per):
                    flips += _pool_exact_iteration(slabs, nslab, ev)
                    strikes.add(float(slabs[0].lightning()[2]))
                _exact_period_end(slabs, nslab, bufs, pl, pr)
            else:
                for h in slabs:
                    h.step(per)
                for r, h in enumerate(slabs):
                    h.halo_pack(0, bufs[r][0].data_ptr())
                    h.halo_pack(1, bufs[r][1].data_ptr())
                for h in slabs:
                    h.sync()
                for r, h in enumerate(slabs):
                    h.halo_unpack(0, bufs[(r - 1) % nslab][1].data_ptr())
                    h.halo_unpack(1, bufs[(r + 1) % nslab][0].data_ptr())
                _pool_exchange(slabs, nslab, ev, pl, pr)
            whole.step(per)
            d_ref = whole.read_particles()
            if exact:
                d, f = _assemble_pool(slabs)  # (asserts: one owner per active droplet, inactive records identical on every rank)
            else:  # the default protocol may leave a stale record behind until a later exchange: assemble without those assertions
                dd, f = [h.read_particles() for h in slabs], np.stack([h.pool_flags() for h in slabs])
                d = dd[0].copy()
                for k in range(nslab):
                    d[f[k] == 2] = dd[k][f[k] == 2]
            same = np.array_equal(d, d_ref) and all(np.array_equal(h.lightning(), whole.lightning()) for h in slabs)
            for fld in ("PRECIP_FB", "PRECIP_DEP", "BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1"):
                ref = whole.read_rect(fld)
                for r, h in enumerate(slabs):
                    a, b = h.read_rect(fld, halo, 0, xo, Y), ref[:, r * xo:(r + 1) * xo]
                    if fld == "PRECIP_FB" and r == 0:
                        a, b = a.copy(), b.copy()
                        a[0, :2], b[0, :2] = 0, 0
                    same = same and np.array_equal(a, b)
            if exact:
                assert same, f"period {period}: exact mode differs from the undecomposed handle"
            identical = identical and same
        results[exact] = identical
        if exact:
            spawned = int(((drops[:, 2] < 0) & (d_ref[:, 2] >= 0)).sum())
            print(f"exact mode: {flips} status-flip events in 36 iterations, {spawned} of the initially inactive droplets active at the end, strike times seen {sorted(strikes)}")
            assert flips > 400, "the scene must exercise the protocol"
            assert max(strikes) > 0, "no lightning strike: the scene does not test the request path"
        for h in slabs:
            h.close()
        whole.close()
    assert results[1] and not results[0], results


@pytest.mark.parametrize("X,Y,cols", [(512, 128, None), (130, 50, None), (4096, 256, (4000, 300))])
def test_device_side_setup_equals_uploaded_grid(pkg, E, X, Y, cols):
    """wx_setup_columns: the textures filled on the device from the 1-D descriptors are bit-identical
    to the arrays synth.terrain_grid builds from the same descriptors and uploads; so is the run that follows."""
    S = pkg.synth
    desc = S.terrain_columns(X, Y, cols=cols, cloud_deck=True)
    base, water, wall = S.terrain_grid(X, Y, cols=cols)
    S.add_cloud_deck(water, wall)
    Xl = X if cols is None else cols[1]
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 40.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    kw = {} if cols is None else dict(X_global=X, x0=cols[0] + 12, halo=12)
    a = E.Handle(Xl if cols is None else Xl - 24, Y, 0, **kw)
    b = E.Handle(Xl if cols is None else Xl - 24, Y, 0, **kw)
    a.upload(base, water, wall)
    b.setup_columns(desc)
    for h in (a, b):
        h.set_params(p, u["initial_T"])
    for f in ("BASE_CUR", "WATER_CUR", "WATER_0", "WALL_CUR", "LIGHT_0"):
        assert np.array_equal(a.read_rect(f), b.read_rect(f)), f
    a.step(6)
    b.step(6)
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1"):
        assert np.array_equal(a.read_rect(f), b.read_rect(f)), f
    bad = dict(desc, wall_rows=desc["wall_rows"].copy())
    bad["wall_rows"][3] = Y + 1
    with pytest.raises(E.WxError):
        b.setup_columns(bad)
    with pytest.raises(ValueError):
        b.setup_columns(dict(desc, snow=desc["snow"][:-1]))


@pytest.mark.parametrize("X,Y,cols,seed,mult,snap", [(4096, 256, None, 0.5, 0.3, 2), (1000, 600, None, 0.8371, 0.45, 1), (4096, 256, (4000, 300), 0.5, 0.3, 2),
                                                     (512, 64, None, 0.2, 0.07, 2), (512, 64, None, 0.2, 0.0, 2), (16384, 512, (16000, 1024), 0.123, 0.9, 4)])
def test_device_side_terrain_equals_host_generator(pkg, E, X, Y, cols, seed, mult, snap):
    """wx_setup_terrain (setupShader.frag:26-92 entirely on the device): rand / noise / the octave sum in double on the device
    against the host generator's descriptors -- wall rows, sea / land, snow and the vegetation byte of every column. The only source of a
    difference is the last bit of the device's sin() at a column whose height sits exactly on a row boundary, so the textures must be
    identical on (at least) all but a handful of columns; in practice they are identical, and then so is the run that follows.
    A slab handle generates its own window of global columns (wrap included)."""
    S = pkg.synth
    desc = S.terrain_columns(X, Y, seed=seed, height_mult=mult, snap=snap, cols=cols)
    Xl = X if cols is None else cols[1]
    kw = {} if cols is None else dict(X_global=X, x0=cols[0] + 12, halo=12)
    a = E.Handle(Xl if cols is None else Xl - 24, Y, 0, **kw)
    b = E.Handle(Xl if cols is None else Xl - 24, Y, 0, **kw)
    a.setup_columns(desc)
    gui = pkg.params.merge_settings(None)
    b.setup_terrain(S.sounding_rows(Y), seed=seed, height_mult=mult, snap=snap, sim_height=float(gui["simHeight"]))
    fields = ("BASE_CUR", "WATER_CUR", "WATER_0", "WALL_CUR")
    fa, fb = {f: a.read_rect(f) for f in fields}, {f: b.read_rect(f) for f in fields}
    differing = np.zeros(Xl, bool)
    for f in fields:
        differing |= (fa[f] != fb[f]).any(axis=(0, 2))
    assert differing.sum() <= max(2, Xl // 2000), (int(differing.sum()), np.flatnonzero(differing)[:10])
    if mult >= 0.10:
        assert len(np.unique(desc["wall_rows"])) >= 3  # hills, not a flat line
    if not differing.any():
        gui["sunAngle"] = 40.0
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
        u["enablePrecipitation"] = 0
        p = pkg.params.fill_struct(pkg.params.WxParams(), u)
        for h in (a, b):
            h.set_params(p, u["initial_T"])
            h.step(5)
        for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0"):
            assert np.array_equal(a.read_rect(f), b.read_rect(f)), f
    with pytest.raises(E.WxError):
        b.setup_terrain(S.sounding_rows(Y), snap=0)
    with pytest.raises(ValueError):
        b.setup_terrain({k: v[:-1] for k, v in S.sounding_rows(Y).items()})


def test_device_side_droplet_pool(pkg, E):
    """wx_init_droplets (initRainDrops app.js:4901-4913 on the device): the pool is a pure function of the seed through
    the shaders' integer hash -- bit-identical to the numpy restatement, identical on a slab handle, all droplets inactive with seeds in
    [0, 1), different seeds give different pools, and a run started from it equals a run started from the uploaded restatement."""
    S = pkg.synth
    X, Y, N = 512, 128, 5000
    ref = S.init_rain_drops_hashed(N, seed=77)
    assert ref.shape == (N, 5) and (ref[:, 2] < -9.0).all() and (ref[:, 2] >= -10.0).all()
    assert (ref[:, [0, 1, 3, 4]] >= 0).all() and (ref[:, [0, 1, 3, 4]] < 1).all()
    assert abs(float(ref[:, 0].mean()) - 0.5) < 0.02 and len(np.unique(ref[:, 0])) > 0.99 * N
    base, water, wall = S.terrain_grid(X, Y)
    S.add_cloud_deck(water, wall)
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1
    u["inactiveDroplets"] = float(N)
    u["spawnChanceMult"] = 5e-3
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    a, b = E.Handle(X, Y, N), E.Handle(X, Y, N)
    a.upload(base, water, wall, ref)
    b.upload(base, water, wall, S.init_rain_drops(N))
    b.init_droplets(77)
    assert np.array_equal(b.read_particles(), ref)
    for h in (a, b):
        h.set_params(p, u["initial_T"])
        h.set_option(h.OPT_SPLAT_ORDER, 1)
        h.step(12)
    da, db = a.read_particles(), b.read_particles()
    assert (da[:, 2] >= 0).sum() > 20  # droplets spawned from the hashed seeds
    assert np.array_equal(da, db) and np.array_equal(a.read_rect("BASE_CUR"), b.read_rect("BASE_CUR"))
    b.init_droplets(78)
    assert not np.array_equal(b.read_particles(), ref)
    sl = E.Handle(256, Y, N, X_global=X, x0=64, halo=64)
    sl.upload(base[:, :384], water[:, :384], wall[:, :384], None)
    sl.init_droplets(77)
    assert np.array_equal(sl.read_particles(), ref)  # (all inactive: every slab holds every inactive record)
    g = E.Handle(X, Y, 0)
    with pytest.raises(E.WxError):
        g.init_droplets(1)


def test_display_field_streaming(pkg, golden, E):
    """wx_stream_frame: the six display fields of a viewport arrive in one pinned buffer, hold the state
    at the time of the call even though more iterations are enqueued right behind it, and do not disturb the run."""
    g, u = golden("precip64")
    u = dict(u, quad_scale=0, enablePrecipitation=1)
    X, Y = int(g["X"]), int(g["Y"])

    def make():
        h = E.Handle(X, Y, len(g["in_drops"]))
        h.upload(g["in_base"], g["in_water"], g["in_wall"], g["in_drops"])
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
        return h

    h, ref = make(), make()
    rect = (8, 4, 40, 30)
    h.step(5)
    ref.step(5)
    want = {name: h.read_rect(name, *rect) for name, _, _ in E.Handle.STREAM_FIELDS}  # the state at the time of the call
    h.stream_frame(*rect)
    h.step(3)  # enqueued behind the copies: must not leak into the frame
    frame = h.stream_wait()
    for name, _, ch in E.Handle.STREAM_FIELDS:
        assert frame[name].shape == (rect[3], rect[2], ch)
        assert np.array_equal(frame[name].reshape(want[name].shape), want[name]), name
    assert np.abs(frame["PRECIP_FB"]).max() > 0
    ref.step(3)
    assert np.array_equal(h.read_rect("WALL_CUR"), ref.read_rect("WALL_CUR"))
    for f in ("BASE_CUR", "WATER_CUR"):  # two runs with particle feedback agree to the order of the splat atomics
        assert np.abs(h.read_rect(f) - ref.read_rect(f)).max() <= 1e-5, f
    h.stream_frame()  # whole grid, buffer re-allocated for the new size
    full = h.stream_wait()
    assert np.array_equal(full["WATER_CUR"], h.read_rect("WATER_CUR"))
    with pytest.raises(E.WxError):
        h.stream_frame(X - 4, 0, 8, 8)  # no wrap


def test_emitted_light_field(pkg, oracle, golden, E, fused):
    """WX_FIELD_EMITTED (the lighting pass's RGBA16F second render target, computed on demand): zero before the first lighting
    pass, the reference's values after 80 iterations, rectangle reads == whole-grid reads, both dtypes, and -- like the
    texture -- not affected by a uniform change that no lighting pass has drawn with yet."""
    g, u = golden("emitted64_day")
    u = dict(u, quad_scale=1, enablePrecipitation=0)
    X, Y = int(g["X"]), int(g["Y"])
    h, o = _make_pair(pkg, oracle, E, X, Y, g["in_base"], g["in_water"], g["in_wall"], u, iter0=int(g["iter0"]))
    assert not h.read_rect("EMITTED").any()
    h.step(80)
    e = h.read_rect("EMITTED")
    r = g["it80_emitted"]
    assert e.dtype == np.float16 and e.shape == (Y, X, 4)
    assert (np.abs(e.astype(np.float32) - r) <= 2e-5 + 2.0 ** -11 * np.abs(r)).all()  # the run's drift + one rounding to binary16
    assert np.count_nonzero(e[..., :3].any(-1)) >= 0.9 * X * Y
    assert np.array_equal(h.read_rect("EMITTED", 5, 7, 33, 21), e[7:28, 5:38])
    f32 = np.zeros((Y, X, 4), np.float32)
    h._chk(E.lib().wx_read_rect(h._h, E.FIELD_IDS["EMITTED"], 0, 0, X, Y, f32.ctypes.data, E.DTYPE_F32))
    assert np.array_equal(f32, e.astype(np.float32))
    with pytest.raises(E.WxError):
        h._chk(E.lib().wx_read_rect(h._h, E.FIELD_IDS["EMITTED"], 0, 0, X, Y, f32.ctypes.data, E.DTYPE_I8))
    night = dict(u, sunAngle=float(np.deg2rad(88.0)))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), night), u["initial_T"])
    assert np.array_equal(h.read_rect("EMITTED"), e)
    h.step(1)
    e2 = h.read_rect("EMITTED")
    glow = (g["it80_wall_cur"][..., 2] == 1) & np.isin(g["it80_wall_cur"][..., 0], (4, 5, 6))
    assert glow.sum() >= 10 and (e2[glow][:, 0] >= 0.03).all() and not (e2 == e).all()
    h.upload(g["in_base"], g["in_water"], g["in_wall"])
    assert not h.read_rect("EMITTED").any()


def test_python_host_new_simulation_round_trip(pkg, tmp_path):
    """WeatherSim.new_simulation (setup on the device + initRainDrops) -> frames with the day/night driver -> save ->
    load -> identical state: the new-simulation path produces a valid .weathersandbox file."""
    sim = pkg.WeatherSim.new_simulation(200, 100, {"dayNightCycle": True, "IterPerFrame": 10})
    for _ in range(3):
        sim.step()
    sf = sim.to_save()
    assert sf.droplets.shape == (200 * 100 // 25, 5) and np.isfinite(sf.base).all() and (sf.wall[0, :, 1] == 0).all()
    path = str(tmp_path / "new.weathersandbox")
    pkg.codec.save(path, sf)
    back = pkg.codec.load(path)
    assert np.array_equal(back.base, sf.base) and np.array_equal(back.water, sf.water) and np.array_equal(back.wall, sf.wall)
    again = pkg.WeatherSim.from_save(back)
    assert np.array_equal(again.to_save().wall, sf.wall)


def test_brush_and_airplane_inputs_bit_exact(pkg, oracle, E, fused):
    """Next-row (f1): user brush (every tool of advectionShader.frag:229-401) and airplane inputs (:415-457):
    HIP == oracle bit for bit, on the inputs/uniforms of the reference goldens (tests/golden/brush64.npz)."""
    import json, os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "brush64.npz"))
    X, Y = int(g["X"]), int(g["Y"])
    cases = json.loads(str(g["cases"]))
    extra = {"airplane_dump": dict(airplaneValues=(0.4, 0.5, 0.7, -1.0)), "airplane_crash_air": dict(airplaneValues=(0.6, 0.6, 0.0, 1.0)),
             "airplane_crash_ground": dict(airplaneValues=(0.2, 0.09, 0.0, 1.0)), "nowrap_brush": dict(userInputType=1, userInputValues=(0.98, 0.5, 0.3, 6.0), wrapHorizontally=0)}
    for case in cases + list(extra):
        if case in extra:
            u = json.loads(str(g["temperature_uniforms"]))
            u.update(userInputType=-1)
            u.update(extra[case])
            n = 2
        else:
            u = json.loads(str(g[f"{case}_uniforms"]))
            n = int(g[f"{case}_niter"]) + 1  # one more iteration than the golden: the edited walls then feed back
        for k in ("userInputValues", "userInputMove", "airplaneValues"):
            u[k] = tuple(u[k])
        u["initial_T"] = g["initial_T"]
        u["enablePrecipitation"] = 0
        u["quad_scale"] = 0
        h, o = _make_pair(pkg, oracle, E, X, Y, g["in_base"], g["in_water"], g["in_wall"], u)
        h.step(n)
        o.step(n)
        for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "WATER_0", "LIGHT_1"):
            assert np.array_equal(h.read_rect(f), o.field(f)), (case, f)
        h.close()


@pytest.mark.gpu
def test_halo_pack_both_equals_two_single_side_calls(pkg, E):
    """wx_halo_pack_both / wx_halo_unpack_both (both sides in one launch, ABI 8) == wx_halo_pack / wx_halo_unpack per side: the same
    buffers byte for byte, the same ghost columns afterwards."""
    import torch
    X, Y, halo, xo = 512, 64, 12, 128
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.default_rng(11)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=1)
    u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    hs = []
    for _ in range(2):
        h = E.Handle(xo, Y, 0, X_global=X, x0=xo, halo=halo)
        idx = (xo - halo + np.arange(xo + 2 * halo)) % X
        h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
        h.set_params(p, u["initial_T"])
        h.step(2)
        hs.append(h)
    nb = hs[0].halo_bytes()
    single = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    both = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    hs[0].halo_pack(0, single[0].data_ptr())
    hs[0].halo_pack(1, single[1].data_ptr())
    hs[1].halo_pack_both(both[0].data_ptr(), both[1].data_ptr())
    hs[0].sync()
    hs[1].sync()
    assert torch.equal(single[0], both[0]) and torch.equal(single[1], both[1])
    assert int(single[0].to(torch.int64).sum()) != 0
    # unpack (crossed, as a slab whose neighbours are itself would): the ghost columns must agree
    hs[0].halo_unpack(0, single[1].data_ptr())
    hs[0].halo_unpack(1, single[0].data_ptr())
    hs[1].halo_unpack_both(both[1].data_ptr(), both[0].data_ptr())
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1"):
        a, b = hs[0].read_rect(f), hs[1].read_rect(f)
        assert np.array_equal(a, b), f
    for h in hs:
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("X,Y,sigma", [(4096, 1024, 0.1), (1000, 96, 0.1), (130, 50, 0.1), (8192, 512, 0.1), (4096, 1024, 0.3), (1000, 96, 0.35)])
def test_dry_pairs_equal_single_iterations(pkg, E, oracle, X, Y, sigma):
    """WX_OPT_DRY_PAIRS (round 5 prototype, csrc/wx_march2.h): two iterations of the water-free dry stencil per launch -- the second
    iteration's input row never leaves the wavefront. Bit for bit the one-iteration kernel at odd and even counts, with frames that end
    inside a pair (the display fields of the last iteration), on banded and unbanded launch shapes and a ragged grid, over terrain-free
    and wall-bottomed states -- and, on the small grid, the oracle. sigma 0.3 / 0.35: hundreds of cells beyond 0.9 cells / iteration, also
    in second iterations, which have no exact path inside the march: recorded and recomputed by k_dry2_fix (round 6; wx_pair_stats)."""
    base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=sigma)
    rng = np.random.Generator(np.random.Philox(9))
    base[1:, :, 2] += rng.normal(0, 1e-3, (Y - 1, X)).astype(np.float32)
    u = _dry_uniforms(pkg, Y)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    hs = []
    for pairs in (0, 1):
        h = E.Handle(X, Y, 0)
        h.upload(base, water, wall)
        h.set_params(p, u["initial_T"])
        h.set_option(E.Handle.OPT_DRY_PAIRS, pairs)
        hs.append(h)
    ref = None
    if X * Y <= 200000:
        ref = oracle.OracleSim(X, Y, 0)
        ref.upload(base, water, wall)
        ref.set_params(u)
    for k in (1, 4, 7, 10, 2):
        for h in hs:
            h.profile(True)
            h.step(k)
        names = [set(h.profile_read()) for h in hs]
        assert "march_dry2_two_iterations_per_launch" not in names[0]
        assert ("march_dry2_two_iterations_per_launch" in names[1]) == (k >= 2), names
        for f in ("BASE_CUR", "BASE_DISP", "WALL_CUR", "WATER_CUR"):
            assert np.array_equal(hs[0].read_rect(f), hs[1].read_rect(f)), (k, f)
        if ref is not None:
            ref.step(k)
            assert np.array_equal(hs[1].read_rect("BASE_CUR"), ref.field("BASE_CUR")), k
        assert hs[0].iter == hs[1].iter
    fixed, repeated = hs[1].pair_stats()
    if sigma >= 0.3:
        assert fixed > 0, "white noise of sigma 0.3 puts second-iteration cells beyond 0.9: the exact path must have run"
    assert hs[0].pair_stats() == (0, 0)
    for h in hs:
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("X,Y", [(1000, 96), (4096, 1024), (2132, 2048)])
def test_dry_pairs_fast_cells_take_the_cell_granular_exact_path(pkg, E, oracle, X, Y):
    """(round 6) The reference has no velocity clamp (advectionShader.frag:85-99). A second-iteration cell of the pair kernel whose
    back-trace is 0.9 cells or more is recorded and recomputed by k_dry2_fix (one wavefront per cell: iteration 1 on a patch from the
    pair's inputs, then the three outputs the cell feeds) -- round 5 repeated the whole grid twice for one such cell. Compact vortices of
    1.4 cells / iteration on strip borders (multiples of 56 columns), segment / band borders, the periodic seam, the floor and the top
    row: bit for bit the one-iteration kernel (and the oracle on the small grid), cells recomputed > 0, NO pair repeated whole. Then
    5 cells / iteration (second-iteration footprints leave the fix kernel's stage) and a list of 4 entries (overflow): the pair is
    repeated whole, still bit for bit."""
    base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.1)
    seg = max(8, min(128, ((3 * (Y // 8)) // 16) & ~3))  # (launch_march_dry2's unit segment height on banded shapes)
    centers = [(0.5, Y * 0.4), (56 * 5 + 0.5, Y - 3.0, -1), (56 * 9, 4.0), (X - 56 * 3 - 1, Y // 8 + 0.5, -1), (X // 2 + 28, float(seg)), (X // 3, Y // 2 - 0.5, -1)]
    n_fast = pkg.synth.add_vortices(base, wall, centers, 5.0, 1.4)
    assert n_fast > 100
    u = _dry_uniforms(pkg, Y)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)

    def handles(b, cap=None):
        hs = []
        for pairs in (0, 1):
            h = E.Handle(X, Y, 0)
            if cap is not None:
                h.set_option(E.Handle.OPT_FIX_CAP, cap)
            h.upload(b, water, wall)
            h.set_params(p, u["initial_T"])
            h.set_option(E.Handle.OPT_DRY_PAIRS, pairs)
            hs.append(h)
        return hs

    def compare(hs, ref, steps):
        for k in steps:
            for h in hs:
                h.step(k)
            for f in ("BASE_CUR", "BASE_DISP"):
                assert np.array_equal(hs[0].read_rect(f), hs[1].read_rect(f)), (k, f)
            if ref is not None:
                ref.step(k)
                assert np.array_equal(hs[1].read_rect("BASE_CUR"), ref.field("BASE_CUR")), k

    ref = None
    if X * Y <= 200000:
        ref = oracle.OracleSim(X, Y, 0)
        ref.upload(base, water, wall)
        ref.set_params(u)
    hs = handles(base)
    compare(hs, ref, (2, 4, 3, 10))
    fixed, repeated = hs[1].pair_stats()
    assert fixed > 0 and repeated == 0, (fixed, repeated)
    for h in hs:
        h.close()
    # second-iteration back-traces of three cells and more: the stage of the fix kernel is left -> the whole pair is repeated
    b3 = base.copy()
    pkg.synth.add_vortices(b3, wall, [(X // 2 + 100.5, Y // 2 + 0.5)], 10.0, 5.0)
    hs = handles(b3)
    compare(hs, None, (2, 4))
    fixed, repeated = hs[1].pair_stats()
    assert repeated > 0, (fixed, repeated)
    for h in hs:
        h.close()
    # a list too short for the recorded cells: the whole pair is repeated
    hs = handles(base, cap=4)
    compare(hs, None, (2, 2))
    fixed, repeated = hs[1].pair_stats()
    assert repeated > 0, (fixed, repeated)
    for h in hs:
        h.close()
