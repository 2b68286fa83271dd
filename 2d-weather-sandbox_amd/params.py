"""Host-side parameter plumbing: guiControls -> uniform struct.

Mirrors the reference's parameter path for the simulation step:
  * defaults                      app.js:347-407  (``guiControls_default``)
  * save-file settings merge      app.js:3378-3399 + libraries/dat.gui.min.js:136-150
  * derived constants             app.js:5436-5476 (``dryLapse``, ``initial_T``)
  * per-GUI uniform push          app.js:3401-3443 (``setGuiUniforms``)
  * sun geometry                  app.js:6510-6561 (``updateSunlight``)

The resulting ``WxParams`` ctypes structure is the C-ABI ``wx_params`` of include/wxsim.h.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, Optional

import numpy as np

# app.js:347-407
GUI_DEFAULTS: Dict[str, Any] = {
    "vorticity": 0.005,
    "dragMultiplier": 0.001,
    "wind": 0.0,
    "globalEffectsStartAlt": 0,
    "globalEffectsEndAlt": 10000,
    "globalDrying": 0.0,
    "globalHeating": 0.0,
    "soundingForcing": 0.0,
    "sunIntensity": 1.0,
    "waterTemperature": 25.0,
    "dynamicWaterTemperature": True,
    "landEvaporation": 0.00005,
    "waterEvaporation": 0.0001,
    "evapHeat": 2.90,
    "meltingHeat": 0.43,
    "condensationRate": 0.0050,
    "waterWeight": 0.25,
    "inactiveDroplets": 0,
    "aboveZeroThreshold": 1.0,
    "subZeroThreshold": 0.005,
    "spawnChance": 0.00005,
    "snowDensity": 0.2,
    "fallSpeed": 0.0003,
    "growthRate0C": 0.0001,
    "growthRate_30C": 0.001,
    "freezingRate": 0.01,
    "meltingRate": 0.01,
    "evapRate": 0.0008,
    "displayMode": "DISP_REAL",
    "wrapHorizontally": True,
    "SmoothCam": True,
    "camSpeed": 0.01,
    "exposure": 1.0,
    "timeOfDay": 9.9,
    "latitude": 45.0,
    "month": 6.65,
    "sunAngle": 9.9,
    "dayNightCycle": True,
    "greenhouseGases": 0.001,
    "waterGreenHouseEffect": 0.0015,
    "IR_rate": 1.0,
    "tool": "TOOL_NONE",
    "brushSize": 20,
    "wholeWidth": False,
    "intensity": 0.01,
    "showGraph": False,
    "realDewPoint": False,
    "enablePrecipitation": True,
    "showDrops": False,
    "paused": False,
    "IterPerFrame": 10,
    "auto_IterPerFrame": True,
    "sound": True,
    "dryLapseRate": 10.0,
    "simHeight": 12000,
    "twelveHourClock": False,
    "lengthUnit": "LENGTH_UNIT_METRIC",
    "tempUnit": "TEMP_UNIT_C",
    "windUnit": "SPEED_UNIT_KMH",
}

DEG2RAD = 0.0174533  # app.js:340 (the reference's own rounded constant)
RAD2DEG = 57.2957795


def merge_settings(saved: Optional[Dict[str, Any]], sim_height: float = 12000) -> Dict[str, Any]:
    """Settings as the reference sees them after loading a save.

    Rule (libraries/dat.gui.min.js:136-150, app.js:3394-3398): a missing numeric control is created as
    -1 and then replaced by its default - so a legitimately saved -1 is reset too; a missing boolean
    becomes False; a missing selector gets its first option (irrelevant to the simulation).
    With ``saved is None`` (new simulation, app.js:3378-3391) the defaults are used, with ``simHeight`` AND
    ``globalEffectsEndAlt`` set to the height chosen in the start dialog (``sim_height``, app.js:437, 3380-3381) -- so a new
    simulation's drying / heating window ends at the top of the domain, not at the default slider value of 10 000 m
    (found by executing those lines: tests/golden/host_uniforms.json).
    """
    if saved is None:
        out = dict(GUI_DEFAULTS)
        out["simHeight"] = sim_height
        out["globalEffectsEndAlt"] = sim_height
        return out
    out = dict(saved)
    for key, dflt in GUI_DEFAULTS.items():
        if isinstance(dflt, bool):
            if key not in out:
                out[key] = False
        elif isinstance(dflt, (int, float)):
            if key not in out or out[key] == -1:
                out[key] = dflt
        else:
            out.setdefault(key, dflt)
    return out


def initial_temperature_profile(Y: int, sim_height: float, dry_lapse: float) -> np.ndarray:
    """``initial_T[y]`` for y in 0..Y (app.js:5467-5474), float64 math then fp32 as gl.uniform4fv does."""
    y = np.arange(Y + 1, dtype=np.float64)
    altitude = y / (Y + 1) * sim_height
    real_temp = np.maximum(15.0 + (altitude - 0.0) * (-70.0 - 15.0) / (12000.0 - 0.0), -60.0)
    pot = (real_temp + 273.15) + (y / Y) * dry_lapse  # realToPotentialT, app.js:708
    return pot.astype(np.float32)


REFERENCE_CELL_HEIGHT_AT_LOAD = 12000.0 / 300.0  # app.js:435


def sounding_arrays(raw_sounding, Y: int, sim_height: float, dry_lapse: float, cell_height: float = REFERENCE_CELL_HEIGHT_AT_LOAD):
    """``realWorldSounding_T / _W / _Vel`` (Y + 1 entries each) from a raw sounding, as app.js builds them for the
    sounding-forcing term of the advection pass: ``rawSoundingToSimSounding`` (app.js:149-186) + app.js:5444-5463.

    ``raw_sounding``: samples ordered from the TOP of the sounding to the ground (the scraper's order: the reference
    walks the list from its last element upward), each with ``alt`` [m], ``t`` and ``td`` [deg C], ``vel`` [km/h] and
    ``angle`` [deg]; samples with a NaN in t / td / vel are skipped. Returns float32 arrays for ``wx_set_params``:
    potential temperature [K], total water of the dew point (maxWater), horizontal velocity in cells / iteration.

    ``cell_height``: what ``msToRawVelocity`` divides by. In the reference that is the GLOBAL ``cellHeight``, which still holds its
    page-load value 12000 / 300 = 40 m when mainScript builds these arrays (app.js:5444-5463) -- it is assigned
    ``simHeight / sim_res_y`` only afterwards (app.js:5476). The velocity profile is therefore scaled for 40 m cells whatever the
    grid (found by executing those lines, tests/golden/host_uniforms.json); the default reproduces the reference, pass
    ``sim_height / Y`` for the physically consistent value."""
    s = list(raw_sounding)
    time_per_iteration = 0.00008                      # hours, app.js:449
    invalid = lambda d: any(math.isnan(float(d[k])) for k in ("t", "td", "vel"))  # sampleIsInvalid, app.js:147
    T = np.zeros(Y + 1, np.float32)
    W = np.zeros(Y + 1, np.float32)
    V = np.zeros(Y + 1, np.float32)
    idx = len(s) - 1  # start from the lowest data point
    for y in range(Y + 1):
        alt = y * (sim_height / Y)
        while s[idx]["alt"] < alt or invalid(s[idx]):  # go up in the sounding until the altitude matches or exceeds
            idx -= 1
            if idx < 0:
                raise ValueError(f"sounding ends below the simulated altitude {alt} m")
        above = s[idx]
        below = s[min(idx + 1, len(s) - 1)]
        smp = {k: float(above[k]) for k in ("t", "td", "vel", "angle")}
        if above["alt"] != alt and alt >= s[-1]["alt"]:
            a = (alt - below["alt"]) / (above["alt"] - below["alt"])
            smp = {k: float(below[k]) * (1 - a) + float(above[k]) * a for k in ("t", "td", "vel", "angle")}  # mixGeneric, app.js:33-40
        vel_ms = smp["vel"] * math.cos(smp["angle"] * DEG2RAD) / 3.6             # km/h along the 2-D plane -> m/s
        raw_vel = vel_ms * 3600.0 / cell_height * time_per_iteration             # msToRawVelocity, app.js:698-704
        T[y] = (smp["t"] + 273.15) + (y / Y) * dry_lapse                         # realToPotentialT(CtoK(t), y)
        W[y] = ((smp["td"] + 273.15) / 250.0) ** 17                              # maxWater(CtoK(td)), app.js:557-561
        V[y] = raw_vel
    return T, W, V


def sun_from_angle(sun_angle_deg: float, sun_intensity_gui: float):
    """(solarZenithAngle [rad], sunIntensity [W/m2]) from guiControls.sunAngle (app.js:6538-6561)."""
    zenith = (sun_angle_deg - 90.0) * DEG2RAD
    inten = sun_intensity_gui * math.pow(max(math.sin((180.0 - sun_angle_deg) * DEG2RAD), 0.0), 0.1) * 1300.0
    return zenith, inten


def sun_angle_from_time(time_of_day: float, month: float, latitude: float) -> float:
    """guiControls.sunAngle [deg] from clock, month and latitude (app.js:6522-6536)."""
    tod = (time_of_day / 24.0) * 2.0 * math.pi - math.pi / 2.0
    tilt_deg = math.sin(month * 0.5236 - 1.92) * 23.5
    t = tilt_deg * DEG2RAD
    l = latitude * DEG2RAD
    ang = math.asin(math.sin(t) * math.sin(l) + math.cos(t) * math.cos(l) * math.sin(tod)) * RAD2DEG
    if latitude - tilt_deg < 0.0:
        ang = 180.0 - ang
    return ang


class WxParams(C.Structure):
    """C-ABI ``wx_params`` (include/wxsim.h). Field order is part of the ABI."""

    _fields_ = [
        ("dragMultiplier", C.c_float),
        ("wind", C.c_float),
        ("vorticity", C.c_float),
        ("landEvaporation", C.c_float),
        ("waterEvaporation", C.c_float),
        ("dynamicWaterTemperature", C.c_float),
        ("evapHeat", C.c_float),
        ("waterWeight", C.c_float),
        ("sunAngle", C.c_float),
        ("dryLapse", C.c_float),
        ("meltingHeat", C.c_float),
        ("condensationRate", C.c_float),
        ("globalDrying", C.c_float),
        ("globalHeating", C.c_float),
        ("soundingForcing", C.c_float),
        ("globalEffectsStartAlt", C.c_float),
        ("globalEffectsEndAlt", C.c_float),
        ("waterTemperature", C.c_float),
        ("sunIntensity", C.c_float),
        ("greenhouseGases", C.c_float),
        ("waterGreenHouseEffect", C.c_float),
        ("IR_rate", C.c_float),
        ("aboveZeroThreshold", C.c_float),
        ("subZeroThreshold", C.c_float),
        ("spawnChanceMult", C.c_float),
        ("snowDensity", C.c_float),
        ("fallSpeed", C.c_float),
        ("growthRate0C", C.c_float),
        ("growthRate_30C", C.c_float),
        ("freezingRate", C.c_float),
        ("meltingRate", C.c_float),
        ("evapRate", C.c_float),
        ("inactiveDroplets", C.c_float),
        ("userInputValues", C.c_float * 4),
        ("userInputMove", C.c_float * 2),
        ("userInputType", C.c_int32),
        ("wrapHorizontally", C.c_int32),
        ("airplaneValues", C.c_float * 4),
        ("enablePrecipitation", C.c_int32),
        ("quad_scale", C.c_int32),
        ("pass_mask", C.c_uint32),
    ]


PASS_VELOCITY = 1
PASS_VORTICITY = 2
PASS_BOUNDARY = 4
PASS_ADVECTION = 8
PASS_PRESSURE = 16
PASS_LIGHTING = 32
PASS_PRECIPITATION = 64
PASS_ALL = 0x7F
PASS_DRY = PASS_VELOCITY | PASS_ADVECTION | PASS_PRESSURE  # BASELINE config 2


def uniforms_from_gui(gui: Dict[str, Any], Y: int, *, sun_angle_deg: Optional[float] = None,
                      quad_scale: int = 1, pass_mask: int = PASS_ALL) -> Dict[str, Any]:
    """All uniform values of the simulation programs, as plain Python numbers.

    ``sun_angle_deg`` overrides guiControls.sunAngle ('MANUAL_ANGLE' path of updateSunlight).
    Keys equal the uniform names in the shaders / the fields of ``wx_params``.
    """
    sim_h = float(gui["simHeight"])
    dry_lapse = sim_h * float(gui["dryLapseRate"]) / 1000.0  # app.js:5439
    ang = float(gui["sunAngle"]) if sun_angle_deg is None else float(sun_angle_deg)
    zenith, sun_int = sun_from_angle(ang, float(gui["sunIntensity"]))
    u = {
        "dragMultiplier": gui["dragMultiplier"],
        "wind": gui["wind"],
        "vorticity": gui["vorticity"],
        "landEvaporation": gui["landEvaporation"],
        "waterEvaporation": gui["waterEvaporation"],
        "dynamicWaterTemperature": 1.0 if gui["dynamicWaterTemperature"] else 0.0,
        "evapHeat": gui["evapHeat"],
        "waterWeight": gui["waterWeight"],
        "sunAngle": zenith,
        "dryLapse": dry_lapse,
        "meltingHeat": gui["meltingHeat"],
        "condensationRate": gui["condensationRate"],
        "globalDrying": gui["globalDrying"],
        "globalHeating": gui["globalHeating"],
        "soundingForcing": gui["soundingForcing"],
        "globalEffectsStartAlt": gui["globalEffectsStartAlt"] / sim_h,  # app.js:3425-3426
        "globalEffectsEndAlt": gui["globalEffectsEndAlt"] / sim_h,
        "waterTemperature": gui["waterTemperature"] + 273.15,  # app.js:3427
        "sunIntensity": sun_int,
        "greenhouseGases": gui["greenhouseGases"],
        "waterGreenHouseEffect": gui["waterGreenHouseEffect"],
        "IR_rate": gui["IR_rate"],
        "aboveZeroThreshold": gui["aboveZeroThreshold"],
        "subZeroThreshold": gui["subZeroThreshold"],
        "spawnChanceMult": gui["spawnChance"],
        "snowDensity": gui["snowDensity"],
        "fallSpeed": gui["fallSpeed"],
        "growthRate0C": gui["growthRate0C"],
        "growthRate_30C": gui["growthRate_30C"],
        "freezingRate": gui["freezingRate"],
        "meltingRate": gui["meltingRate"],
        "evapRate": gui["evapRate"],
        # the inactiveDroplets uniform is never pushed by setGuiUniforms: GL default 0 until the loop's
        # first 600-iteration count (app.js:5957-5966)
        "inactiveDroplets": 0.0,
        "userInputValues": (0.0, 0.0, 0.0, 0.0),
        "userInputMove": (0.0, 0.0),
        "userInputType": -1,  # app.js:5749
        "wrapHorizontally": 1 if gui.get("wrapHorizontally", True) else 0,
        "airplaneValues": (0.0, 0.0, 0.0, 0.0),
        "enablePrecipitation": 1 if gui.get("enablePrecipitation", True) else 0,
        "quad_scale": int(quad_scale),
        "pass_mask": int(pass_mask),
    }
    u["initial_T"] = initial_temperature_profile(Y, sim_h, dry_lapse)
    return u


def fill_struct(struct: C.Structure, u: Dict[str, Any]) -> C.Structure:
    """Copy matching keys of ``u`` into a ctypes structure (arrays element-wise)."""
    names = {f[0]: f[1] for f in struct._fields_}
    for k, v in u.items():
        if k not in names:
            continue
        if isinstance(v, (tuple, list, np.ndarray)):
            arr = getattr(struct, k)
            for i, e in enumerate(v):
                arr[i] = e
        else:
            setattr(struct, k, v)
    return struct
