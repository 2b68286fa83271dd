""".weathersandbox codec + settings merge + derived parameters (SURVEY.md Appendix B numbers)."""
import json

import numpy as np
import pytest


def _golden_save(pkg, golden):
    """Rebuild a SaveFile from the committed fixture data (the reference's save is not on the GPU box)."""
    g, u = golden("save100qa_precip")
    sf = pkg.codec.SaveFile(int(g["X"]), int(g["Y"]), g["in_base"], g["in_water"], g["in_wall"], g["in_drops"], [],
                            {"vorticity": 0.007, "dragMultiplier": 0.01, "globalEffectsHeight": 5000})
    return sf


def test_roundtrip(pkg, golden):
    sf = _golden_save(pkg, golden)
    sf.stations = [(10, 20), (55, 3)]
    blob = pkg.codec.encode(sf)
    assert int.from_bytes(blob[:4], "little") == 263574036
    sf2 = pkg.codec.decode(blob)
    assert (sf2.X, sf2.Y) == (100, 100)
    assert np.array_equal(sf2.base, sf.base) and np.array_equal(sf2.water, sf.water) and np.array_equal(sf2.wall, sf.wall)
    assert np.array_equal(sf2.droplets, sf.droplets) and sf2.droplets.shape == (400, 5)
    assert sf2.stations == sf.stations and sf2.settings == sf.settings


def test_js_number_formatting(pkg):
    """codec.js_number == JavaScript's Number::toString (checked against `node -e` when this list was written)."""
    f = pkg.codec.js_number
    cases = {0.00001: "0.00001", 1e-7: "1e-7", 1.5e-7: "1.5e-7", 0.000001: "0.000001", 5.0: "5", -0.0: "0", 0.1: "0.1", 100: "100", 1e21: "1e+21",
             1e20: "100000000000000000000", 123456789.125: "123456789.125", -2.5: "-2.5", 67.45275198770811: "67.45275198770811",
             0.007: "0.007", 1.2e-10: "1.2e-10", 12000.0: "12000", 0.25: "0.25", True: "true", 3: "3", float("nan"): "null"}
    for x, want in cases.items():
        assert f(x) == want, (x, f(x), want)
    assert pkg.codec.js_json({"a": [1.0, 0.00001, "x"], "b": None, "c": {"d": False, "e": 1e-7}}) == '{"a":[1,0.00001,"x"],"b":null,"c":{"d":false,"e":1e-7}}'


REF_SAVE = "/root/reference/saves/100 X 100 Test.weathersandbox"


@pytest.mark.skipif(not __import__("os").path.exists(REF_SAVE), reason="the reference tree is only present in the build container")
def test_reencoding_the_reference_save_is_byte_identical(pkg):
    """decode -> encode of the reference's own save: the deflated PAYLOAD (dimensions, three textures, droplets, stations and the
    guiControls JSON written by JSON.stringify, app.js:6610-6621) comes out byte for byte; only the deflate stream differs."""
    import zlib
    blob = open(REF_SAVE, "rb").read()
    sf = pkg.codec.decode(blob)
    again = pkg.codec.encode(sf)
    assert again[:4] == blob[:4]
    assert zlib.decompress(again[4:]) == zlib.decompress(blob[4:])


def test_rejects_wrong_version(pkg):
    with pytest.raises(ValueError):
        pkg.codec.decode(b"\x01\x02\x03\x04" + b"garbage")
    with pytest.raises(ValueError):
        pkg.codec.decode(b"")


def test_settings_merge_rule(pkg):
    P = pkg.params
    old = {"vorticity": 0.007, "evapHeat": 1.9, "globalEffectsHeight": 5000, "waterWeight": -1, "sound": True}
    m = P.merge_settings(old)
    assert m["vorticity"] == 0.007 and m["evapHeat"] == 1.9
    assert m["condensationRate"] == 0.005  # missing numeric -> default
    assert m["waterWeight"] == 0.25  # a saved -1 is reset too (dat.GUI patch side effect)
    assert m["dynamicWaterTemperature"] is False  # missing boolean -> false
    assert m["globalEffectsEndAlt"] == 10000 and m["globalEffectsStartAlt"] == 0
    # new simulation: the defaults with simHeight / globalEffectsEndAlt = the start dialog's height (app.js:3380-3381; executed in
    # tests/test_host_golden.py)
    assert P.merge_settings(None) == dict(P.GUI_DEFAULTS, simHeight=12000, globalEffectsEndAlt=12000)
    assert P.merge_settings(None, sim_height=8000)["globalEffectsEndAlt"] == 8000


def test_derived_uniforms(pkg):
    P = pkg.params
    gui = P.merge_settings({"sunAngle": 67.45275198770811, "sunIntensity": 1})
    u = P.uniforms_from_gui(gui, 100)
    assert u["dryLapse"] == 120.0
    assert abs(u["sunAngle"] - (-0.39352388)) < 1e-7
    assert abs(u["sunIntensity"] - 1289.7039) < 1e-3
    T0 = u["initial_T"]
    assert T0.shape == (101,) and T0.dtype == np.float32
    assert T0[0] == np.float32(288.15) and abs(T0[100] - 333.15) < 1e-4
    assert abs(u["globalEffectsEndAlt"] - 10000 / 12000) < 1e-12
    assert u["waterTemperature"] == 298.15 and u["userInputType"] == -1


def test_golden_uniforms_are_the_merged_save_settings(pkg, golden):
    """The uniform values stored with the goldens are what params.py derives (guards params drift)."""
    g, u = golden("save100qa")
    assert abs(u["vorticity"] - 0.007) < 1e-12 and u["dynamicWaterTemperature"] == 0.0
    assert abs(u["evapHeat"] - 1.9) < 1e-12 and abs(u["wind"] + 1e-4) < 1e-12
