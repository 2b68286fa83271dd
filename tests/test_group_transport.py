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
    assert L.wx_group_create(2, None, 1024, 64, 12, 100, 0, C.byref(g)) != 0  # particles: host-driven exchange (slab.py)
    assert b"slab.py" in L.wx_group_last_error(None)
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
