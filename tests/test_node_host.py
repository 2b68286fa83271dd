"""Node.js host (host/sim_host.js + the N-API addon): JS-side logic on CPU, end-to-end on the GPU."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
HAVE_HEADERS = os.path.exists("/usr/include/node/node_api.h")

needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")


@needs_node
def test_js_host_logic_matches_python(pkg):
    out = json.loads(subprocess.check_output([NODE, os.path.join(ROOT, "host", "selftest.js")]))
    assert out["ok"]
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings({"sunAngle": 67.45275198770811, "sunIntensity": 1}), 100)
    assert np.array_equal(np.asarray(out["initial_T"], np.float32), u["initial_T"][:4])
    assert out["sunAngle"] == u["sunAngle"] and out["sunIntensity"] == u["sunIntensity"]
    # startSimulation()'s clock: the Python host and the JS Date arithmetic agree (incl. month roll-over)
    from weather_sandbox_amd.sim import initial_sim_datetime
    cases = [(6.65, 9.9, True), (6.65, 9.9, False), (6.67, 11.44416, True), (1.0, 0.0, True), (12.99, 23.99, True), (13.016, 5.5, True), (3.5, 12.25, False)]
    for c, js in zip(cases, out["clocks"]):
        t = initial_sim_datetime(*c)
        assert [t.year, t.month - 1, t.day, t.hour, t.minute, t.second] == js, (c, js, t)


@needs_node
@pytest.mark.skipif(not HAVE_HEADERS, reason="node headers missing")
def test_addon_builds_loads_and_fails_loudly_without_gpu(pkg):
    from weather_sandbox_amd import engine
    engine.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    js = ("const m=require('%s'); console.log(m.abiVersion());"
          "try{m.create(64,32,0); console.log('created')}catch(e){console.log('ERR '+e.message)}" % os.path.join(ROOT, "host", "wxsim_napi.node"))
    out = subprocess.check_output([NODE, "-e", js]).decode().split("\n")
    assert out[0] == "11"
    import torch
    if not torch.cuda.is_available():
        assert out[1].startswith("ERR") and "no CPU fallback" in out[1]


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("sun_fixed", [True, False], ids=["sun-fixed", "day-night-cycle"])
def test_node_host_end_to_end_equals_python_host(pkg, golden, tmp_path, sun_fixed):
    """save in -> N iterations through JS/N-API/C-ABI/HIP -> save out == the same run through the Python host; with the clock running
    (dayNightCycle: updateSunlight before every frame, app.js:6510-6561) both hosts push a new sun per frame and still end bit for bit alike."""
    g, u = golden("save100qa_precip")
    settings = {"vorticity": 0.007, "dragMultiplier": 0.01, "wind": -0.0001, "globalDrying": 1e-05, "evapHeat": 1.9, "meltingHeat": 0.6,
                "waterWeight": 0.5, "subZeroThreshold": 0.01, "spawnChance": 2e-05, "freezingRate": 0.0025, "meltingRate": 0.0025,
                "evapRate": 0.0005, "sunAngle": 67.45275198770811, "timeOfDay": 11.44416, "month": 6.67, "dayNightCycle": True,
                "enablePrecipitation": True, "wrapHorizontally": True, "IterPerFrame": 26, "sunIntensity": 1, "latitude": 45}
    sf = pkg.codec.SaveFile(int(g["X"]), int(g["Y"]), g["in_base"], g["in_water"], g["in_wall"], g["in_drops"], [], settings)
    src, dst = str(tmp_path / "in.weathersandbox"), str(tmp_path / "out.weathersandbox")
    pkg.codec.save(src, sf)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    n = 60
    out = subprocess.check_output([NODE, os.path.join(ROOT, "host", "sim_host.js"), src, str(n), dst] + (["--sun-fixed"] if sun_fixed else []))
    info = json.loads(out.decode().strip().split("\n")[-1])
    assert info["iterNum"] == n
    got = pkg.codec.load(dst)
    sim = pkg.WeatherSim.from_save(sf, sun_angle_deg=settings["sunAngle"]) if sun_fixed else pkg.WeatherSim.from_save(sf)
    left = n
    while left > 0:
        k = min(left, 26)
        sim.step(k)
        left -= k
    ref = sim.to_save()
    assert np.array_equal(got.wall, ref.wall)
    assert np.array_equal(got.base, ref.base) and np.array_equal(got.water, ref.water)
    assert got.droplets.shape == ref.droplets.shape
    assert got.settings["vorticity"] == 0.007


@needs_node
@pytest.mark.parametrize("X,Y", [(256, 96), (1000, 200)])
def test_js_terrain_columns_match_python(pkg, X, Y):
    """terrainColumns() of host/sim_host.js (new-simulation path) against synth.terrain_columns: same float64 math."""
    js = ("const H=require('%s'); const d=H.terrainColumns(%d,%d,H.mergeSettings(null));"
          "console.log(JSON.stringify({rows:Array.from(d.wallRows),sea:Array.from(d.sea),veg:Array.from(d.vegNoise),snow:Array.from(d.snow),"
          "T:Array.from(d.T_air),tot:Array.from(d.totalWater),cloud:Array.from(d.cloudWater)}))" % (os.path.join(ROOT, "host", "sim_host.js"), X, Y))
    d = json.loads(subprocess.check_output([NODE, "-e", js]))
    ref = pkg.synth.terrain_columns(X, Y)
    assert np.array_equal(np.asarray(d["rows"]), ref["wall_rows"]) and np.array_equal(np.asarray(d["sea"]), ref["sea"])
    assert np.allclose(d["veg"], ref["veg_noise"], rtol=0, atol=1e-6)  # V8 vs libm sin(): last-ulp differences x 43758 x 150
    for k, r in (("snow", "snow"), ("T", "T_air"), ("tot", "total_water"), ("cloud", "cloud_water")):
        assert np.array_equal(np.asarray(d[k], np.float32), ref[r]), k


@needs_node
@pytest.mark.gpu
def test_node_new_simulation_and_streaming(pkg, tmp_path):
    """WeatherSim.newSimulation (setupColumns on the device) + streamFrame through JS/N-API == the Python host."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    X, Y, n = 256, 96, 30
    out_file = str(tmp_path / "new.weathersandbox")
    js = ("const H=require('%s'), fs=require('fs');"
          "const sim=H.WeatherSim.newSimulation(%d,%d,{sunFixed:true,settings:{enablePrecipitation:false}});"
          "sim.frame(%d); const tp=sim.tunePlacement(2,3); sim.setOption(2,0);"  # (placement search + an option through N-API: the state must not change)
          "const fr=sim.streamFrame(8,4,64,32); sim.frame(5); const v=fr.wait();"
          "let s=0; for (const x of v.WATER_CUR) s+=x; let t=0; for (const x of v.BASE_DISP) t+=x; let e=0; for (const x of v.EMITTED) e+=x;"
          "const eh=new Uint16Array(4*3*2); sim.addon.readRect(sim.h, H.FIELD.EMITTED, 8, 80, 3, 2, eh);"
          "const ef=new Float32Array(4*3*2); sim.addon.readRect(sim.h, H.FIELD.EMITTED, 8, 80, 3, 2, ef);"
          "console.log(JSON.stringify({iter:sim.iterNum(), water:s, base:t, wall0:v.WALL_DISP[0], n:v.CURL.length, emitted:e, eh:Array.from(eh), ef:Array.from(ef), tp:tp}));"
          "fs.writeFileSync('%s', H.encodeSave(sim.toSave())); sim.destroy();" % (os.path.join(ROOT, "host", "sim_host.js"), X, Y, n, out_file))
    info = json.loads(subprocess.check_output([NODE, "-e", js]).decode().strip().split("\n")[-1])
    assert info["iter"] == n + 5 and info["n"] == 64 * 32
    assert len(info["tp"]) == 2 and 0 < info["tp"][1] <= info["tp"][0]
    E = pkg.engine
    gui = pkg.params.merge_settings({"enablePrecipitation": False})  # droplets exist (the save format needs X*Y/25) but stay inert
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)  # the hosts' default (sim.py / sim_host.js)
    h = E.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(n)
    fr = {f: h.read_rect(f, 8, 4, 64, 32) for f in ("WATER_CUR", "BASE_DISP", "WALL_DISP")}
    assert np.isclose(info["water"], float(fr["WATER_CUR"].astype(np.float64).sum()), rtol=1e-12)
    assert np.isclose(info["base"], float(fr["BASE_DISP"].astype(np.float64).sum()), rtol=1e-12)
    assert info["wall0"] == int(fr["WALL_DISP"].reshape(-1)[0])
    # emittedLight: the streamed block holds binary16 bits (sum of the uint16 codes), readRect takes Uint16Array (bits) or Float32Array
    # (sunlight comes down one row per iteration: after 30 of them the viewport's rows 4..35 of 96 are still dark, rows 80.. are lit)
    assert info["emitted"] == int(h.read_rect("EMITTED", 8, 4, 64, 32).view(np.uint16).astype(np.int64).sum())
    h.step(5)
    e = h.read_rect("EMITTED", 8, 80, 3, 2)
    assert e.any() and info["eh"] == e.view(np.uint16).ravel().tolist() and info["ef"] == e.astype(np.float32).ravel().tolist()
    got = pkg.codec.load(out_file)
    assert np.array_equal(got.base, h.read_rect("BASE_CUR")) and np.array_equal(got.water, h.read_rect("WATER_0"))
    assert np.array_equal(got.wall, h.read_rect("WALL_CUR"))
    assert got.droplets.shape == (X * Y // 25, 5) and (got.droplets[:, 2] < 0).all()  # initRainDrops: all inactive


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 4])
def test_node_new_simulation_with_device_terrain(pkg, tmp_path, gpus):
    """WeatherSim.newSimulation({deviceTerrain: true}): the terrain noise of the setup shader evaluated on the device (wx_setup_terrain; with
    --gpus N every slab generates its own columns) -- the same save file as the host-side descriptors produce, bit for bit."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    X, Y, n = 1024, 100, 12
    outs = []
    for dev in ("true", "false"):
        out_file = str(tmp_path / f"new_{dev}.weathersandbox")
        slab = "gpus:%d,halo:12,transport:2," % gpus if gpus > 1 else ""
        js = ("const H=require('%s'), fs=require('fs');"
              "const sim=H.WeatherSim.newSimulation(%d,%d,{%ssunFixed:true,deviceTerrain:%s,seed:0.37,heightMult:0.4,settings:{enablePrecipitation:false}});"
              "sim.frame(%d); fs.writeFileSync('%s', H.encodeSave(sim.toSave())); sim.destroy();" % (os.path.join(ROOT, "host", "sim_host.js"), X, Y, slab, dev, n, out_file))
        subprocess.check_call([NODE, "-e", js])
        outs.append(pkg.codec.load(out_file))
    a, b = outs
    assert len(np.unique((a.wall[..., 1] == 0).sum(0))) >= 3  # hills (low ones on a 100-row grid)
    assert np.array_equal(a.base, b.base) and np.array_equal(a.water, b.water) and np.array_equal(a.wall, b.wall)


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 4])
def test_node_new_simulation_with_device_droplets(pkg, gpus):
    """WeatherSim.newSimulation({dropletSeed}): the droplet pool generated on the device (wx_init_droplets), identically on every slab of a
    decomposed domain -- what readParticles assembles equals the numpy restatement of the generator, before and (particles running)
    consistent after a few iterations: the same droplets are active on one handle and on four slabs in exact mode."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    X, Y, N = 1024, 128, 4096
    slab = "gpus:%d,halo:64,transport:2," % gpus if gpus > 1 else ""
    js = ("const H=require('%s');"
          "const sim=H.WeatherSim.newSimulation(%d,%d,{%ssunFixed:true,deviceTerrain:true,nDroplets:%d,dropletSeed:1234});"
          "const d0=Array.from(sim.readParticles()); sim.setOption(1,1); if (sim.slabs) sim.setOption(7,1); sim.frame(8);"
          "const d1=Array.from(sim.readParticles()); console.log(JSON.stringify({d0:d0,d1:d1})); sim.destroy();"
          % (os.path.join(ROOT, "host", "sim_host.js"), X, Y, slab, N))
    info = json.loads(subprocess.check_output([NODE, "-e", js]).decode().strip().split("\n")[-1])
    ref = pkg.synth.init_rain_drops_hashed(N, 1234)
    assert np.array_equal(np.asarray(info["d0"], np.float32).reshape(N, 5), ref)
    d1 = np.asarray(info["d1"], np.float32).reshape(N, 5)
    assert np.isfinite(d1).all()
    # the same run on one Python handle (deterministic splat order): the JS host on one handle / on four exact slabs must match it
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    h = pkg.engine.Handle(X, Y, N)
    h.setup_terrain(pkg.synth.sounding_rows(Y), sim_height=float(gui["simHeight"]))
    h.init_droplets(1234)
    h.set_option(h.OPT_SPLAT_ORDER, 1)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(8)
    assert np.array_equal(h.read_particles(), d1)


@needs_node
@pytest.mark.gpu
def test_addon_rejects_short_arrays(pkg):
    """The N-API shim checks every typed array against the handle's dimensions (X*Y*4, Y+1, X, nDroplets*5) before the
    C ABI reads through the pointer: a short array is a RangeError, a wrong type a TypeError, never an out-of-bounds read."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    js = ("const m=require('%s'); const X=64,Y=32,N=10,n=X*Y*4; const h=m.create(X,Y,N); const r=[];"
          "function t(name,f){try{f(); r.push([name,'ok'])}catch(e){r.push([name,e.constructor.name])}}"
          "const B=new Float32Array(n),W=new Float32Array(n),L=new Int8Array(n),D=new Float32Array(N*5);"
          "t('short_base',()=>m.upload(h,new Float32Array(n-1),W,L,D));"
          "t('short_wall',()=>m.upload(h,B,W,new Int8Array(n-4),D));"
          "t('short_drops',()=>m.upload(h,B,W,L,new Float32Array(N*5-1)));"
          "t('wrong_type',()=>m.upload(h,new Float64Array(n),W,L,D));"
          "t('good_upload',()=>m.upload(h,B,W,L,D));"
          "t('short_T0',()=>m.setParams(h,{},new Float32Array(Y)));"
          "t('short_snd',()=>m.setParams(h,{},new Float32Array(Y+1),new Float32Array(3)));"
          "t('good_params',()=>m.setParams(h,{},new Float32Array(Y+1).fill(300)));"
          "t('short_cols',()=>m.setupColumns(h,new Int32Array(X-1),new Uint8Array(X),new Float64Array(X),new Float32Array(X),"
          "new Float32Array(Y),new Float32Array(Y),new Float32Array(Y),null));"
          "t('short_rows',()=>m.setupColumns(h,new Int32Array(X),new Uint8Array(X),new Float64Array(X),new Float32Array(X),"
          "new Float32Array(Y-1),new Float32Array(Y),new Float32Array(Y),null));"
          "m.destroy(h); console.log(JSON.stringify(r));" % os.path.join(ROOT, "host", "wxsim_napi.node"))
    got = dict(json.loads(subprocess.check_output([NODE, "-e", js]).decode().strip().split("\n")[-1]))
    assert got == {"short_base": "RangeError", "short_wall": "RangeError", "short_drops": "RangeError", "wrong_type": "TypeError",
                   "good_upload": "ok", "short_T0": "RangeError", "short_snd": "RangeError", "good_params": "ok",
                   "short_cols": "RangeError", "short_rows": "RangeError"}, got


def _raw_sounding():
    # top of the sounding first, as the reference's scraper delivers it; one invalid sample in the middle
    alts = [13000, 9000, 6000, 3000, 1500, 800, 300, 0]
    nan = float("nan")
    return [{"alt": a, "t": 15.0 - 0.0065 * a, "td": (nan if a == 3000 else 10.0 - 0.008 * a), "vel": 10.0 + a / 500.0, "angle": 30.0 + a / 200.0}
            for a in alts]


def test_sounding_arrays_interpolation(pkg):
    """params.sounding_arrays = rawSoundingToSimSounding (app.js:149-186) + app.js:5444-5463, checked by hand."""
    Y, sim_h, lapse = 40, 12000.0, 120.0
    T, W, V = pkg.params.sounding_arrays(_raw_sounding(), Y, sim_h, lapse)
    assert T.shape == W.shape == V.shape == (Y + 1,) and T.dtype == np.float32
    # y = 0 is exactly the ground sample
    assert np.isclose(T[0], 15.0 + 273.15) and np.isclose(W[0], ((10.0 + 273.15) / 250.0) ** 17, rtol=1e-6)
    # y = 2 -> 600 m: between 300 m and 800 m (a = 0.6)
    a = 0.6
    t = (15.0 - 0.0065 * 300) * (1 - a) + (15.0 - 0.0065 * 800) * a
    assert np.isclose(T[2], t + 273.15 + (2 / Y) * lapse, rtol=1e-6)
    vel = (10.0 + 300 / 500.0) * (1 - a) + (10.0 + 800 / 500.0) * a
    ang = (30.0 + 300 / 200.0) * (1 - a) + (30.0 + 800 / 200.0) * a
    assert np.isclose(V[2], vel * np.cos(ang * 0.0174533) / 3.6 * 3600.0 / 40.0 * 0.00008, rtol=1e-6)  # (40 m: the reference's cellHeight at that point)
    # 2100 m (y = 7): the invalid 3000 m sample is skipped, interpolation runs from 1500 m to 6000 m ... with the sample
    # BELOW taken as the next list element (the invalid one) exactly like the reference does -> NaN dew point propagates
    assert np.isnan(W[7]) and np.isfinite(T[7])
    with pytest.raises(ValueError):
        pkg.params.sounding_arrays(_raw_sounding(), Y, 20000.0, lapse)  # sounding ends below the model top


@needs_node
def test_js_sounding_arrays_match_python(pkg):
    Y, sim_h, lapse = 40, 12000.0, 120.0
    raw = json.dumps(_raw_sounding()).replace("NaN", "null")
    js = ("const H=require('%s'); const raw=JSON.parse('%s').map(function(d){for (const k in d) if (d[k]===null) d[k]=NaN; return d;});"
          "const s=H.soundingArrays(raw,%d,%f,%f); console.log(JSON.stringify({T:Array.from(s.T),W:Array.from(s.W).map(function(x){return isNaN(x)?null:x;}),V:Array.from(s.Vel)}))"
          % (os.path.join(ROOT, "host", "sim_host.js"), raw, Y, sim_h, lapse))
    d = json.loads(subprocess.check_output([NODE, "-e", js]))
    T, W, V = pkg.params.sounding_arrays(_raw_sounding(), Y, sim_h, lapse)
    assert np.array_equal(np.asarray(d["T"], np.float32), T) and np.array_equal(np.asarray(d["V"], np.float32), V)
    Wj = np.asarray([np.nan if x is None else x for x in d["W"]], np.float32)
    assert np.array_equal(np.isnan(Wj), np.isnan(W)) and np.array_equal(Wj[~np.isnan(W)], W[~np.isnan(W)])
