"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see wx_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Any, Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwxoracle.so")


class OracleParams(C.Structure):
    _fields_ = [
        ("X", C.c_int32), ("Y", C.c_int32), ("X_global", C.c_int32), ("x_off", C.c_int32),
        ("quad_scale", C.c_int32),
        ("dragMultiplier", C.c_float), ("wind", C.c_float),
        ("vorticity", C.c_float), ("landEvaporation", C.c_float), ("waterEvaporation", C.c_float),
        ("dynamicWaterTemperature", C.c_float), ("evapHeat", C.c_float), ("waterWeight", C.c_float),
        ("sunAngle", C.c_float), ("dryLapse", C.c_float),
        ("meltingHeat", C.c_float), ("condensationRate", C.c_float), ("globalDrying", C.c_float),
        ("globalHeating", C.c_float), ("soundingForcing", C.c_float),
        ("globalEffectsStartAlt", C.c_float), ("globalEffectsEndAlt", C.c_float),
        ("waterTemperature", C.c_float),
        ("sunIntensity", C.c_float), ("greenhouseGases", C.c_float),
        ("waterGreenHouseEffect", C.c_float), ("IR_rate", C.c_float),
        ("aboveZeroThreshold", C.c_float), ("subZeroThreshold", C.c_float),
        ("spawnChanceMult", C.c_float), ("snowDensity", C.c_float), ("fallSpeed", C.c_float),
        ("growthRate0C", C.c_float), ("growthRate_30C", C.c_float), ("freezingRate", C.c_float),
        ("meltingRate", C.c_float), ("evapRate", C.c_float), ("inactiveDroplets", C.c_float),
        ("userInputValues", C.c_float * 4), ("userInputMove", C.c_float * 2),
        ("userInputType", C.c_int32), ("wrapHorizontally", C.c_int32),
        ("airplaneValues", C.c_float * 4),
        ("enablePrecipitation", C.c_int32),
        ("varyings", C.c_void_p),
        ("subpixel_bits", C.c_int32),
        ("splat_order", C.c_int32),
    ]


FIELDS = {
    "BASE_CUR": (0, np.float32, 4), "BASE_DISP": (1, np.float32, 4),
    "WATER_0": (2, np.float32, 4), "WATER_CUR": (3, np.float32, 4),
    "WALL_CUR": (4, np.int8, 4), "WALL_DISP": (5, np.int8, 4),
    "LIGHT_0": (6, np.float32, 4), "LIGHT_1": (7, np.float32, 4),
    "CURL": (8, np.float32, 1), "VORT": (9, np.float32, 2),
    "PRECIP_FB": (10, np.float32, 4), "PRECIP_DEP": (11, np.float32, 2),
    "EMITTED": (14, np.float32, 4),  # unrounded; the reference's attachment is RGBA16F: compare .astype(np.float16)
}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "wx_oracle.c")
    hdr = os.path.join(_HERE, "wx_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libwxoracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
        PP = C.POINTER(OracleParams)
        L.wxo_velocity.argtypes = [PP, fp, i8p, fp, i8p]
        L.wxo_curl.argtypes = [PP, fp, fp]
        L.wxo_vorticity.argtypes = [PP, fp, fp]
        L.wxo_boundary.argtypes = [PP, fp, C.c_float, fp, fp, fp, i8p, fp, fp, fp, fp, fp, i8p]
        L.wxo_advection.argtypes = [PP, fp, C.c_void_p, C.c_void_p, C.c_void_p, fp, fp, i8p, fp, fp, i8p]
        L.wxo_pressure.argtypes = [PP, fp, i8p, fp, i8p]
        L.wxo_lighting.argtypes = [PP, fp, fp, i8p, fp, fp]
        L.wxo_lighting_mrt.argtypes = [PP, fp, fp, i8p, fp, fp, C.c_void_p]
        L.wxo_precipitation.argtypes = [PP, C.c_float, C.c_int, fp, fp, fp, fp, fp, fp, fp]
        L.wxo_lightning_location.argtypes = [PP, C.c_float, fp, fp]
        L.wxo_hash.argtypes = [C.c_uint32]
        L.wxo_hash.restype = C.c_uint32
        L.wxo_random2d.argtypes = [C.c_float, C.c_float]
        L.wxo_random2d.restype = C.c_float
        L.wxo_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.wxo_create.restype = C.c_void_p
        L.wxo_destroy.argtypes = [C.c_void_p]
        L.wxo_upload.argtypes = [C.c_void_p, fp, fp, i8p, C.c_void_p]
        L.wxo_set_params.argtypes = [C.c_void_p, PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wxo_step.argtypes = [C.c_void_p, C.c_int]
        L.wxo_step_ex.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        L.wxo_get_iter.argtypes = [C.c_void_p]
        L.wxo_get_iter.restype = C.c_int64
        L.wxo_set_iter.argtypes = [C.c_void_p, C.c_int64]
        L.wxo_field.argtypes = [C.c_void_p, C.c_int]
        L.wxo_field.restype = C.c_void_p
        _lib = L
    return _lib


def make_params(u: Dict[str, Any], X: int, Y: int, X_global: Optional[int] = None, x_off: int = 0) -> OracleParams:
    p = OracleParams()
    names = {f[0] for f in OracleParams._fields_}
    for k, v in u.items():
        if k not in names or k == "varyings":
            continue
        if isinstance(v, (tuple, list, np.ndarray)):
            arr = getattr(p, k)
            for i, e in enumerate(v):
                arr[i] = e
        else:
            setattr(p, k, v)
    p.X, p.Y = X, Y
    p.X_global = X if X_global is None else X_global
    p.x_off = x_off
    vary = u.get("varyings")
    if vary is not None:
        vary = np.ascontiguousarray(vary, np.float32)
        assert vary.size == X * Y * 4
        p._keep = vary  # keep alive
        p.varyings = vary.ctypes.data
    return p


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleSim:
    """Whole-simulation oracle object (texture set + ping-pong of app.js:5830-6005)."""

    def __init__(self, X: int, Y: int, n_drops: int = 0, X_global: Optional[int] = None, x_off: int = 0):
        self.X, self.Y, self.n_drops = X, Y, n_drops
        self.X_global, self.x_off = (X if X_global is None else X_global), x_off
        self._h = lib().wxo_create(X, Y, n_drops)
        self.pass_mask = 0x7F

    def close(self):
        if self._h:
            lib().wxo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, base, water, wall, drops=None):
        base = np.ascontiguousarray(base, np.float32).reshape(-1)
        water = np.ascontiguousarray(water, np.float32).reshape(-1)
        wall = np.ascontiguousarray(wall, np.int8).reshape(-1)
        d = None
        if drops is not None and self.n_drops > 0:
            d = np.ascontiguousarray(drops, np.float32).reshape(-1)
            assert d.size == self.n_drops * 5
        lib().wxo_upload(self._h, base, water, wall, _vp(d))

    def set_params(self, u: Dict[str, Any]):
        p = make_params(u, self.X, self.Y, self.X_global, self.x_off)
        self._vary_keep = getattr(p, "_keep", None)
        self.pass_mask = int(u.get("pass_mask", 0x7F))
        T0 = np.ascontiguousarray(u["initial_T"], np.float32)
        assert T0.size >= self.Y + 1
        snd = [u.get(k) for k in ("sounding_T", "sounding_W", "sounding_Vel")]
        snd = [None if a is None else np.ascontiguousarray(a, np.float32) for a in snd]
        lib().wxo_set_params(self._h, C.byref(p), _vp(T0), _vp(snd[0]), _vp(snd[1]), _vp(snd[2]))

    def step(self, n: int = 1):
        lib().wxo_step_ex(self._h, n, self.pass_mask)

    @property
    def iter(self) -> int:
        return lib().wxo_get_iter(self._h)

    @iter.setter
    def iter(self, v: int):
        lib().wxo_set_iter(self._h, v)

    def view(self, name: str) -> np.ndarray:
        """Writable no-copy view of a grid field (used by the slab halo exchange test engine)."""
        fid, dt, ch = FIELDS[name]
        ptr = lib().wxo_field(self._h, fid)
        ct = C.c_float if dt == np.float32 else C.c_int8
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), (self.Y, self.X, ch))

    def field(self, name: str) -> np.ndarray:
        if name == "LIGHTNING":
            ptr = lib().wxo_field(self._h, 12)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (4,)).copy()
        if name == "DROPS":
            ptr = lib().wxo_field(self._h, 13)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (self.n_drops, 5)).copy()
        fid, dt, ch = FIELDS[name]
        ptr = lib().wxo_field(self._h, fid)
        ct = C.c_float if dt == np.float32 else C.c_int8
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), (self.Y, self.X, ch))
        return a.copy()
