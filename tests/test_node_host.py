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


@needs_node
@pytest.mark.skipif(not HAVE_HEADERS, reason="node headers missing")
def test_addon_builds_loads_and_fails_loudly_without_gpu(pkg):
    from weather_sandbox_amd import engine
    engine.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    js = ("const m=require('%s'); console.log(m.abiVersion());"
          "try{m.create(64,32,0); console.log('created')}catch(e){console.log('ERR '+e.message)}" % os.path.join(ROOT, "host", "wxsim_napi.node"))
    out = subprocess.check_output([NODE, "-e", js]).decode().split("\n")
    assert out[0] == "2"
    import torch
    if not torch.cuda.is_available():
        assert out[1].startswith("ERR") and "no CPU fallback" in out[1]


@needs_node
@pytest.mark.gpu
def test_node_host_end_to_end_equals_python_host(pkg, golden, tmp_path):
    """save in -> N iterations through JS/N-API/C-ABI/HIP -> save out == the same run through the Python host."""
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
    out = subprocess.check_output([NODE, os.path.join(ROOT, "host", "sim_host.js"), src, str(n), dst, "--sun-fixed"])
    info = json.loads(out.decode().strip().split("\n")[-1])
    assert info["iterNum"] == n
    got = pkg.codec.load(dst)
    sim = pkg.WeatherSim.from_save(sf, sun_angle_deg=settings["sunAngle"])
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
          "sim.frame(%d); const fr=sim.streamFrame(8,4,64,32); sim.frame(5); const v=fr.wait();"
          "let s=0; for (const x of v.WATER_CUR) s+=x; let t=0; for (const x of v.BASE_DISP) t+=x;"
          "console.log(JSON.stringify({iter:sim.iterNum(), water:s, base:t, wall0:v.WALL_DISP[0], n:v.CURL.length}));"
          "fs.writeFileSync('%s', H.encodeSave(sim.toSave())); sim.destroy();" % (os.path.join(ROOT, "host", "sim_host.js"), X, Y, n, out_file))
    info = json.loads(subprocess.check_output([NODE, "-e", js]).decode().strip().split("\n")[-1])
    assert info["iter"] == n + 5 and info["n"] == 64 * 32
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
    h.step(5)
    got = pkg.codec.load(out_file)
    assert np.array_equal(got.base, h.read_rect("BASE_CUR")) and np.array_equal(got.water, h.read_rect("WATER_0"))
    assert np.array_equal(got.wall, h.read_rect("WALL_CUR"))
    assert got.droplets.shape == (X * Y // 25, 5) and (got.droplets[:, 2] < 0).all()  # initRainDrops: all inactive
