"""The halo exchange inside the library (wx_comm_* / wx_exchange / wx_slab_step, wx_group_*; include/wxsim.h): N slabs driven through
the C ABI alone -- from Python over ctypes and from Node over the N-API addon -- equal the undecomposed handle bit for bit."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


def _scene(pkg, X, Y, seed=2):
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.default_rng(seed)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    return base, water, wall, u


FIELDS = ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1")


def test_group_api_refuses_what_it_cannot_do(pkg):
    """(CPU) argument checks happen before any device work."""
    E = pkg.engine
    with pytest.raises(E.WxError):
        E.Group(3, 1000, 64, halo=12)       # width not divisible
    L = E.lib()
    import ctypes as C
    g = C.c_void_p()
    assert L.wx_group_create(2, None, 1024, 64, 3, 0, 0, C.byref(g)) != 0    # halo below the dependency cone
    assert L.wx_group_create(2, None, 1024, 64, 6, 100, 0, C.byref(g)) != 0   # slabs with particles need halo >= 12
    assert b"halo >= 12" in L.wx_group_last_error(None)
    assert L.wx_exchange(None) != 0 and L.wx_slab_step(None, 1) != 0 and L.wx_group_step(None, 1) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("nslab,halo,X,wet", [(2, 12, 1024, True), (4, 24, 4096, True), (4, 12, 2048, False), (3, 18, 1536, True)])
def test_group_local_transport_equals_whole_domain(pkg, nslab, halo, X, wet, monkeypatch):
    """wx_group_step on ONE GPU (several slabs per device: device-to-device copies between the slabs' buffers, event-fenced, every slab
    on compute and comm streams of its own, the host never waits): ten exchange periods + 1 iteration == the undecomposed handle."""
    E = pkg.engine
    Y = 64
    base, water, wall, u = _scene(pkg, X, Y)
    if not wet:
        u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
        u["enablePrecipitation"] = 0
        base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.1)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL)
    assert g.transport == E.TRANSPORT_LOCAL
    g.upload(base, water, wall)
    g.set_params(p, u["initial_T"])
    n_iter = 10 * (halo // 6) + 1
    for k in (1, 3, n_iter - 4):  # call boundaries anywhere relative to the exchange periods
        g.step(k)
    g.sync()
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(p, u["initial_T"])
    whole.step(n_iter)
    for f in FIELDS:
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    # the display-side fields are stored by the LAST piece of a step only (WX_OVERLAP_MORE_TO_COME on every other piece of wx_group_step's
    # period pieces): what the group shows after the call is what one handle shows
    for f in (("BASE_DISP", "WATER_0", "CURL") if wet else ("BASE_DISP",)):
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    assert g.slabs[0].iter == whole.iter
    # a re-upload restarts the exchange period (fresh ghost columns): same result again
    g.upload(base, water, wall)
    g.step(n_iter)
    whole.upload(base, water, wall)  # (the iteration counter is not reset by an upload, app.js:4628-4640: both continue from n_iter)
    whole.step(n_iter)
    for f in FIELDS:
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    g.close()
    whole.close()


@pytest.mark.gpu
def test_dry_slabs_exchange_the_base_texture_alone(pkg):
    """(ABI 10) Slabs that agreed on the water-free dry stencil send 16 of the 68 bytes per halo cell -- that iteration writes nothing but
    the base texture -- and run their periods in order, iterations in pairs. The ghost columns of water, wall and light must still be what
    the upload made them when the parameters leave the dry stencil: the WET iterations that follow (full-size messages again) equal the
    undecomposed handle bit for bit, which they cannot unless every ghost texture was valid when they started."""
    E = pkg.engine
    X, Y, halo, nslab = 2048, 64, 12, 4
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    ud = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    uw = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    ud["enablePrecipitation"] = uw["enablePrecipitation"] = 0
    base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.1)
    pd, pw = (pkg.params.fill_struct(pkg.params.WxParams(), u) for u in (ud, uw))
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL)
    whole = E.Handle(X, Y, 0)
    for h in (g, whole):
        h.upload(base, water, wall)
        h.set_params(pd, ud["initial_T"])
    assert all(h.halo_bytes() == halo * Y * 68 and h.halo_message_bytes() == halo * Y * 16 for h in g.slabs)
    g.slabs[0].profile(True)
    for k in (1, 4, 8):  # 13 iterations: six exchanges of the period 2, call boundaries anywhere
        g.step(k)
        whole.step(k)
    g.sync()
    # (a period of two iterations as ONE pair launch: with split edge / interior iterations -- B, then A -- there would be no pair at all)
    assert g.slabs[0].profile_read()["march_dry2_two_iterations_per_launch"][1] >= 4
    g.slabs[0].profile(False)
    for f in FIELDS:
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    assert all(h.halo_message_bytes() == halo * Y * 16 for h in g.slabs)
    for h in (g, whole):
        h.set_params(pw, uw["initial_T"])
    g.step(9)
    whole.step(9)
    g.sync()
    assert all(h.halo_message_bytes() == h.halo_bytes() for h in g.slabs)  # (dropped by the first wet step, on every slab alike)
    for f in FIELDS:
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    # ... and a slab that is given new contents on its own is refused until the slabs have agreed again
    for h in (g, whole):
        h.upload(base, water, wall)
        h.set_params(pd, ud["initial_T"])
    assert all(h.halo_message_bytes() == halo * Y * 16 for h in g.slabs)
    xo = X // nslab
    cols = (1 * xo - halo + np.arange(xo + 2 * halo)) % X
    wet_water = water[:, cols].copy()
    wet_water[40, 100, 0] = 0.01
    g.slabs[1].upload(base[:, cols].copy(), wet_water, wall[:, cols].copy())
    with pytest.raises(E.WxError, match="THIS slab is not water-free"):
        g.step(halo // 6)
        g.sync()
    g.close()
    whole.close()


def _jet_scene(pkg, X, Y, nslab, wet):
    """Jets of 1.5 .. 2.5 cells / iteration straddling EVERY slab edge (and the periodic seam), in both directions: faster than the
    shaders' documented range (common.glsl:40-41) but nothing the reference clamps (advectionShader.frag:85-99)."""
    if wet:
        base, water, wall, u = _scene(pkg, X, Y)
    else:
        u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
        u["enablePrecipitation"] = 0
        base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.1)
    xo = X // nslab
    for b in range(nslab):  # slab edge at column b * xo (b = 0: the periodic seam)
        cols = (b * xo + np.arange(-70, 70)) % X
        ramp = np.interp(np.arange(-70, 70), [-70, -40, 40, 70], [0.0, 1.0, 1.0, 0.0]).astype(np.float32)
        for rows, v in ((slice(34, 40), 1.8), (slice(44, 50), -1.6)):
            base[rows, cols, 0] = np.where(wall[rows, cols, 1] != 0, v * ramp, base[rows, cols, 0])
        core = (b * xo + np.arange(-3, 3)) % X  # a core beyond two cells / iteration: the exact path's general fallback
        base[36:38, core, 0] = np.where(wall[36:38, core, 1] != 0, 2.4, base[36:38, core, 0])
    return base, water, wall, u


@pytest.mark.gpu
@pytest.mark.parametrize("nslab,wet", [(2, True), (4, True), (2, False), (4, False)])
def test_group_is_exact_at_any_speed(pkg, nslab, wet):
    """SURVEY 8e's determinism check without the |v| < 1 caveat of rounds 1-4: one iteration's dependency cone is 6 + floor|vx| columns,
    the marching kernels measure |vx|, and the slabs size every exchange period by the bound they agree on (wx_slab_set_vx_bound; here
    through wx_group_step: bootstrap scan after the upload, then the maxima travel with every exchange). Jets of 1.5 .. 2.5 cells /
    iteration across every slab edge, wet iteration and the dry north-star stencil, in-library transport: bit for bit one handle."""
    E = pkg.engine
    X, Y, halo = 2048, 64, 42
    base, water, wall, u = _jet_scene(pkg, X, Y, nslab, wet)
    assert np.abs(base[..., 0]).max() > 2.0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL)
    g.upload(base, water, wall)
    g.set_params(p, u["initial_T"])
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(p, u["initial_T"])
    cones = []
    for k in (1, 5, 8, 3):  # call boundaries anywhere relative to the (now variable) exchange periods
        g.step(k)
        whole.step(k)
        cones.append(g.slabs[0].slab_cone)
    g.sync()
    assert cones[0] >= 6 + 3, cones  # the bootstrap scan saw the 2.4 core: 6 + floor(1.25 * 2.4 + 0.25) = 9 columns per iteration
    assert all(h.slab_cone == cones[-1] and h.slab_period == halo // cones[-1] for h in g.slabs)
    vmax = float(np.abs(whole.read_rect("BASE_CUR")[..., 0]).max())
    assert vmax > 1.0, vmax  # still faster than one cell per iteration at the end: the cone of rounds 1-4 would not have held
    for f in FIELDS[:3] if not wet else FIELDS:
        assert np.array_equal(g.read(f), whole.read_rect(f)), f
    g.close()
    whole.close()


@pytest.mark.gpu
def test_slab_outrunning_its_vx_bound_is_reported(pkg):
    """A host that drives wx_step_overlap / wx_halo_* itself and never tells the handle about fast flow (wx_slab_set_vx_bound) keeps the
    6-column cone -- and gets WX_E_STATE from the next blocking call once a |vx| reaches one cell per iteration: never a silent divergence."""
    E = pkg.engine
    X, Y, halo = 1024, 64, 12
    base, water, wall, u = _jet_scene(pkg, X, Y, 2, True)
    h = E.Handle(X // 2, Y, 0, X_global=X, x0=0, halo=halo)
    idx = (-halo + np.arange(X // 2 + 2 * halo)) % X
    h.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    assert h.slab_cone == 6 and h.slab_period == 2
    h.step(1)
    with pytest.raises(E.WxError) as ei:
        h.sync()
    assert ei.value.code == -5 and "|vx| reached" in str(ei.value)
    # told about the flow, the same handle sizes its period accordingly; 12 ghost columns are too few for 6 cells / iteration
    v = h.slab_vx_take()
    assert v > 1.5
    h.slab_set_vx_bound(v)
    assert h.slab_cone == 6 + int(1.25 * v + 0.25) >= 8 and h.slab_period == 1
    with pytest.raises(E.WxError) as ei:
        h.slab_set_vx_bound(6.0)
    assert "wider halo" in str(ei.value)
    h.slab_set_vx_bound(0.7)
    assert h.slab_cone == 7 and h.slab_period == 1
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wet", [True, False])
def test_interior_jet_faster_than_the_slab_can_follow_is_reported(pkg, wet):
    """The strips three halo widths and more from a slab's edges are not watched for |vx| -- they consume no ghost column -- until a
    velocity spans the two halo widths between them and the ghost columns (a state that has blown up: tools/fuzz_parity.py --mode group met
    2 600 cells / iteration under a brush). Such a jet is a violation like any other: WX_E_STATE, not a silent difference."""
    E = pkg.engine
    X, Y, halo, nslab = 2048, 64, 12, 2
    base, water, wall, u = _scene(pkg, X, Y)
    if not wet:
        u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
        u["enablePrecipitation"] = 0
        base, water, wall = pkg.synth.dry_grid(X, Y, flow_sigma=0.05)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    # a quiet flow passes ...
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL)
    calm = base.copy()
    calm[..., 0] *= 0.1
    g.upload(calm, water, wall)
    g.set_params(p, u["initial_T"])
    g.step(3)
    g.read("BASE_CUR")
    g.close()
    # ... a jet of 40 cells / iteration in the middle of slab 0 (500 columns from either of its edges; 2 * halo - 8 = 16) does not
    jet = calm.copy()
    jet[40:44, 500:506, 0] = np.where(wall[40:44, 500:506, 1] != 0, 40.0, 0.0)
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL)
    with pytest.raises(E.WxError) as ei:
        g.upload(jet, water, wall)
        g.set_params(p, u["initial_T"])
        g.step(3)
        g.sync()
        g.read("BASE_CUR")
    assert ei.value.code == -5, ei.value
    g.close()


def _particle_scene(pkg, X, Y, N, seed=4):
    """A cloud deck over terrain, droplets everywhere incl. on the slab edges and at the domain edge, a drift that carries them across."""
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    pkg.synth.add_cloud_deck(water, wall)
    rng = np.random.default_rng(seed)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.3, (Y, X)), 0).astype(np.float32)
    drops = pkg.synth.init_rain_drops(N)
    na = N // 3
    px = rng.uniform(-1, 1, na)
    px[:200] = np.where(rng.random(200) < 0.5, -1 + rng.uniform(0, 7, 200) * 2 / X, 1 - rng.uniform(0, 7, 200) * 2 / X)
    drops[:na, 0] = px.astype(np.float32)
    drops[:na, 1] = rng.uniform(-0.6, 0.2, na).astype(np.float32)
    drops[:na, 2] = rng.uniform(0.05, 1.0, na).astype(np.float32)
    drops[:na, 3] = np.where(rng.random(na) < 0.3, rng.uniform(0.1, 0.5, na), 0).astype(np.float32)
    drops[:na, 4] = 1.0
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1
    u["inactiveDroplets"] = float(N - na)
    u["spawnChanceMult"] = 5e-4
    return base, water, wall, drops, u


@pytest.mark.gpu
@pytest.mark.parametrize("nslab,exact", [(2, False), (4, False), (4, True)])
def test_group_with_particles_equals_whole_domain(pkg, nslab, exact):
    """Slabs with particles on the library's transport (wx_group_step: status flips + lightning all-gathered with a stride every rank derives from the previous counts, edge
    droplets in the halos' batch, no host round trip): with the deterministic splat order and WX_OPT_POOL_EXACT the group is
    BIT-IDENTICAL to one handle -- pool, feedback, deposition, lightning, every field, checked after calls that end inside and at the
    end of exchange periods; with the per-period protocol the same droplets are active at the end of whole periods and the fields agree
    to the few re-spawn probes a period can miss."""
    E = pkg.engine
    X, Y, halo, N = 512, 128, 64, 6000
    base, water, wall, drops, u = _particle_scene(pkg, X, Y, N)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL, n_droplets=N)
    g.upload(base, water, wall, drops)
    g.set_params(p, u["initial_T"])
    g.set_option(E.Handle.OPT_SPLAT_ORDER, 1)
    g.set_option(E.Handle.OPT_POOL_EXACT, 1 if exact else 0)
    whole = E.Handle(X, Y, N)
    whole.upload(base, water, wall, drops)
    whole.set_params(p, u["initial_T"])
    whole.set_option(whole.OPT_SPLAT_ORDER, 1)
    per = 1 + (halo - 12) // 9
    done = 0
    for k in (per, 2, per - 2, 2 * per + 3, per - 3):  # call boundaries anywhere; the run ends at a period boundary
        g.step(k)
        whole.step(k)
        done += k
        if exact:  # identical at any moment (the fields; the pool is assembled after an exchange below)
            for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "PRECIP_DEP", "LIGHT_0", "LIGHT_1"):
                assert np.array_equal(g.read(f), whole.read_rect(f)), (f, done)
            assert all(np.array_equal(h.lightning(), whole.lightning()) for h in g.slabs), done
    assert done % per == 0
    g.sync()
    d, d_ref = g.particles(), whole.read_particles()
    assert (d_ref[:, 2] >= 0).sum() > 500 and (d_ref[:, 2] < 0).sum() > 500
    fb, fb_ref = g.read("PRECIP_FB"), whole.read_rect("PRECIP_FB")
    fb[0, :2], fb_ref[0, :2] = 0, 0  # texels (0,0) / (1,0) are the reference's mailboxes, not kept on slabs
    if exact:
        assert np.array_equal(d, d_ref)
        assert np.array_equal(fb, fb_ref)
    else:
        same = (d[:, 2] >= 0) == (d_ref[:, 2] >= 0)
        assert same.mean() > 0.995, same.mean()
        assert np.abs(d - d_ref)[same].max() <= 1e-2
        assert np.array_equal(g.read("WALL_CUR"), whole.read_rect("WALL_CUR"))
        for f in ("BASE_CUR", "WATER_CUR"):
            a, b = g.read(f), whole.read_rect(f)
            assert ((a != b).any(-1)).mean() < 0.02 and np.abs(a - b).max() <= 1e-2 * max(1.0, float(np.abs(b).max())), f
    g.close()
    whole.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nslab,ndrops", [(4, 6000), (2, 6000), (4, 0)])
def test_group_exchange_overlap_changes_nothing(pkg, nslab, ndrops):
    """WX_OPT_EXCHANGE_OVERLAP: the exchange on the handles' side streams (grid-only: behind the interior strips before and after; with
    particles: grid halos, feedback / deposition texture and the droplet pool behind the interior strips of the NEXT iteration) against
    the same protocol in order on the compute stream -- every field, the pool and the lightning state bit for bit, at call boundaries
    inside and at the end of periods (deterministic splat order: the textures are then a pure function of the pool)."""
    E = pkg.engine
    X, Y, halo = 512, 128, 64
    base, water, wall, drops, u = _particle_scene(pkg, X, Y, 6000)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    gs = []
    for overlap in (1, 0):
        g = E.Group(nslab, X, Y, halo=halo, devices=[0] * nslab, transport=E.TRANSPORT_LOCAL, n_droplets=ndrops)
        g.upload(base, water, wall, drops if ndrops else None)
        g.set_params(p, u["initial_T"])
        if ndrops:
            g.set_option(E.Handle.OPT_SPLAT_ORDER, 1)
        g.set_option(E.Handle.OPT_EXCHANGE_OVERLAP, overlap)
        gs.append(g)
    per = (1 + (halo - 12) // 9) if ndrops else halo // 6
    fields = ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1") + (("PRECIP_DEP", "PRECIP_FB") if ndrops else ())
    done = 0
    for k in (per, 1, per - 1, 2, 3 * per + 1, per - 3, per):
        for g in gs:
            g.step(k)
        done += k
        for f in fields:
            assert np.array_equal(gs[0].read(f), gs[1].read(f)), (f, done)
    if ndrops:
        for a, b in zip(gs[0].slabs, gs[1].slabs):
            assert np.array_equal(a.lightning(), b.lightning())
        d0, d1 = gs[0].particles(), gs[1].particles()
        assert (d0[:, 2] >= 0).sum() > 500
        assert np.array_equal(d0, d1)
    for g in gs:
        g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
def test_group_start_up_burst_of_status_flips(pkg, exact):
    """The start-up burst: a large all-inactive pool over a cloud deck with a high spawn chance flips more droplets in its first period
    than the steady-state stride of the status-flip all-gather holds (65 536 events). The transport carries the whole event buffer until
    it has seen the counts of a period, then 4 x the largest count (pool_stride_update): nothing is lost -- in exact mode the two slabs
    stay bit-identical to one handle through the burst and after it; the default protocol ends without an overflow report."""
    E = pkg.engine
    X, Y, halo, N = 512, 128, 64, 800000
    base, water, wall, _, u = _particle_scene(pkg, X, Y, 6000)
    drops = pkg.synth.init_rain_drops_hashed(N, 9)
    u["inactiveDroplets"] = 1000.0  # (the normaliser of the spawn chance: every probe that lands in cloud spawns)
    u["spawnChanceMult"] = 2.0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    g = E.Group(2, X, Y, halo=halo, devices=[0, 0], transport=E.TRANSPORT_LOCAL, n_droplets=N)
    g.upload(base, water, wall, drops)
    g.set_params(p, u["initial_T"])
    g.set_option(E.Handle.OPT_SPLAT_ORDER, 1)
    g.set_option(E.Handle.OPT_POOL_EXACT, 1 if exact else 0)
    per = 1 + (halo - 12) // 9
    whole = None
    if exact:
        whole = E.Handle(X, Y, N)
        whole.upload(base, water, wall, drops)
        whole.set_params(p, u["initial_T"])
        whole.set_option(whole.OPT_SPLAT_ORDER, 1)
    for k in (1, per - 1, per, 2 * per):
        g.step(k)
        g.sync()  # (an overflow of an exchange buffer would be reported here)
        if whole is not None:
            whole.step(k)
            if k == 1:  # the premise: ONE iteration flips more droplets per slab than the steady-state stride holds
                flipped = int((whole.read_particles()[:, 2] >= 0).sum())
                assert flipped > 2 * 65536 * 1.2, flipped
    d = g.particles()
    active = int((d[:, 2] >= 0).sum())
    assert active > 70000, active  # more flips than the steady-state stride holds, most of them in the first period
    if whole is not None:
        assert np.array_equal(d, whole.read_particles())
        for f in ("BASE_CUR", "WATER_CUR", "PRECIP_DEP"):
            assert np.array_equal(g.read(f), whole.read_rect(f)), f
        whole.close()
    g.close()


@pytest.mark.gpu
def test_rccl_comm_of_one_rank(pkg):
    """wx_comm_unique_id / wx_comm_init with world = 1: RCCL itself is bound (dlopen) and a communicator created on the device -- as much
    of the one-rank-per-process path as a 1-GPU box can execute; wx_slab_step then is wx_step. A lone slab with ghost columns is refused."""
    E = pkg.engine
    X, Y = 1024, 64
    base, water, wall, u = _scene(pkg, X, Y, seed=5)
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    h = E.Handle(X, Y, 0)
    h.upload(base, water, wall)
    h.set_params(p, u["initial_T"])
    uid = E.Handle.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    h.comm_init(uid, 0, 1)
    h.slab_step(13)
    h.exchange()  # (nothing to exchange)
    h.sync()
    whole = E.Handle(X, Y, 0)
    whole.upload(base, water, wall)
    whole.set_params(p, u["initial_T"])
    whole.step(13)
    for f in FIELDS:
        assert np.array_equal(h.read_rect(f), whole.read_rect(f)), f
    s = E.Handle(512, Y, 0, X_global=X, x0=0, halo=12)
    with pytest.raises(E.WxError):
        s.comm_init(uid, 0, 1)
    for x in (h, whole, s):
        x.close()


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_node_host_four_slabs_equal_python_whole_domain(pkg, golden, tmp_path):
    """`node host/sim_host.js in.weathersandbox N out.weathersandbox --gpus 4`: JS -> N-API -> wx_group_* -> HIP, four slabs with the
    halo exchange inside the library == the same save run as ONE handle by the Python host, bit for bit (row g3 of the verdict)."""
    g, _ = golden("save100raw")
    X, Y = int(g["X"]), int(g["Y"])
    reps = 4  # 100 columns are too few for four slabs with ghost columns: tile the save four times in x (periodic, so still a valid state)
    tile = lambda a: np.ascontiguousarray(np.tile(a.reshape(Y, X, 4), (1, reps, 1)))
    settings = {"sunAngle": 67.45275198770811, "dayNightCycle": False, "enablePrecipitation": False, "sunIntensity": 1, "IterPerFrame": 7,
                "vorticity": 0.007, "dragMultiplier": 0.01, "wind": -0.0001, "globalDrying": 1e-05, "evapHeat": 1.9, "waterWeight": 0.5}
    drops = pkg.synth.init_rain_drops(pkg.codec.num_droplets(X * reps, Y))  # (the save format carries X*Y/25 droplets; precipitation is off)
    sf = pkg.codec.SaveFile(X * reps, Y, tile(g["in_base"]), tile(g["in_water"]), tile(g["in_wall"]), drops, [], settings)
    src, dst = str(tmp_path / "in.weathersandbox"), str(tmp_path / "out.weathersandbox")
    pkg.codec.save(src, sf)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    n = 45
    out = subprocess.check_output([NODE, os.path.join(ROOT, "host", "sim_host.js"), src, str(n), dst, "--sun-fixed", "--gpus", "4", "--halo", "12",
                                   "--transport", "local"])
    info = json.loads(out.decode().strip().splitlines()[-1])
    assert info["iterNum"] == n
    js = pkg.codec.load(dst)
    sim = pkg.WeatherSim.from_save(pkg.codec.load(src), sun_angle_deg=settings["sunAngle"])
    sim.step(n)
    py = sim.to_save()
    assert np.array_equal(js.base, py.base) and np.array_equal(js.water, py.water) and np.array_equal(js.wall, py.wall)
    assert np.array_equal(js.droplets, drops)  # carried through unchanged


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_node_host_four_slabs_with_particles_exact(pkg, tmp_path):
    """`node host/sim_host.js ... --gpus 4 --splat-order --exact`: the partitioned droplet pool through JS -> N-API -> wx_group_* with
    WX_OPT_POOL_EXACT == the Python host on ONE handle with the deterministic splat order, bit for bit (fields AND the droplets of the
    output save)."""
    X, Y = 512, 100  # (X * Y a multiple of 25: the save format's droplet section is only well defined then, app.js:1282, 1315, 6595)
    N = pkg.codec.num_droplets(X, Y)
    base, water, wall, drops, u = _particle_scene(pkg, X, Y, N)
    settings = {"sunAngle": 35.0, "dayNightCycle": False, "enablePrecipitation": True, "sunIntensity": 1, "IterPerFrame": 5, "spawnChance": 5e-4,
                "inactiveDroplets": float(N - N // 3)}
    sf = pkg.codec.SaveFile(X, Y, base, water, wall, drops, [], settings)
    src, dst = str(tmp_path / "in.weathersandbox"), str(tmp_path / "out.weathersandbox")
    pkg.codec.save(src, sf)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    n = 23
    out = subprocess.check_output([NODE, os.path.join(ROOT, "host", "sim_host.js"), src, str(n), dst, "--sun-fixed", "--gpus", "4", "--transport", "local",
                                   "--splat-order", "--exact"])
    info = json.loads(out.decode().strip().splitlines()[-1])
    assert info["iterNum"] == n
    js = pkg.codec.load(dst)
    sim = pkg.WeatherSim.from_save(pkg.codec.load(src), sun_angle_deg=settings["sunAngle"])
    sim.handle.set_option(sim.handle.OPT_SPLAT_ORDER, 1)
    left = n
    while left > 0:
        k = min(left, 5)
        sim.step(k)
        left -= k
    py = sim.to_save()
    assert (py.droplets[:, 2] >= 0).sum() > 300 and (py.droplets[:, 2] < 0).sum() > 300
    assert np.array_equal(js.base, py.base) and np.array_equal(js.water, py.water) and np.array_equal(js.wall, py.wall)
    assert np.array_equal(js.droplets, py.droplets)
