"""``.weathersandbox`` save-file codec.

Format (reference app.js:1261-1344 ``loadData`` and app.js:6575-6628 ``prepareDownload``):

    u32 LE versionID (263574036, legacy 1939327491)  ||  zlib-deflate of:
        u16 X, u16 Y,
        f32 base[4*X*Y], f32 water[4*X*Y], i8 wall[4*X*Y],
        f32 droplets[5 * floor(X*Y/25)],
        u16 nStations, i16 stationXY[2*n],            (current version only)
        UTF-8 JSON guiControls (to EOF)               (current version only)

The reference uses pako 1.0.3 (``libraries/pako.min.js``) = RFC 1950 zlib; Python's zlib is the same codec.
"""
from __future__ import annotations

import json
import struct
import zlib
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

SAVE_FILE_VERSION_ID = 263574036  # app.js:345
LEGACY_VERSION_ID = 1939327491  # app.js:1265
NUM_DROPLETS_DIVIDER = 25  # app.js:452


@dataclass
class SaveFile:
    X: int
    Y: int
    base: np.ndarray  # (Y, X, 4) float32, row 0 = bottom
    water: np.ndarray  # (Y, X, 4) float32
    wall: np.ndarray  # (Y, X, 4) int8
    droplets: np.ndarray  # (N, 5) float32: pos.xy, mass.xy (water, ice), density
    stations: List[Tuple[int, int]] = field(default_factory=list)
    settings: Optional[Dict[str, Any]] = None  # raw guiControls JSON; None for legacy files
    version: int = SAVE_FILE_VERSION_ID


def num_droplets(X: int, Y: int) -> int:
    """NUM_DROPLETS = X*Y/25 (app.js:1282); drawArrays truncates a fractional count."""
    return (X * Y) // NUM_DROPLETS_DIVIDER


def decode(data: bytes) -> SaveFile:
    if len(data) < 4:
        raise ValueError("not a .weathersandbox file: too short")
    (version,) = struct.unpack_from("<I", data, 0)
    if version not in (SAVE_FILE_VERSION_ID, LEGACY_VERSION_ID):
        raise ValueError(f"Incompatible file! version id {version}")  # app.js:1352
    raw = zlib.decompress(data[4:])
    X, Y = struct.unpack_from("<HH", raw, 0)
    off = 4
    n = X * Y * 4
    base = np.frombuffer(raw, np.float32, n, off).reshape(Y, X, 4).copy()
    off += n * 4
    water = np.frombuffer(raw, np.float32, n, off).reshape(Y, X, 4).copy()
    off += n * 4
    wall = np.frombuffer(raw, np.int8, n, off).reshape(Y, X, 4).copy()
    off += n
    # the byte length uses the (possibly fractional) JS NUM_DROPLETS; Blob.slice truncates
    nd_bytes = int((X * Y) / NUM_DROPLETS_DIVIDER * 4 * 5)
    nd = nd_bytes // 20
    droplets = np.frombuffer(raw, np.float32, nd * 5, off).reshape(nd, 5).copy()
    off += nd_bytes
    stations: List[Tuple[int, int]] = []
    settings = None
    if version == SAVE_FILE_VERSION_ID:
        (ns,) = struct.unpack_from("<h", raw, off)
        off += 2
        st = np.frombuffer(raw, np.int16, ns * 2, off)
        off += ns * 4
        stations = [(int(st[2 * i]), int(st[2 * i + 1])) for i in range(ns)]
        txt = raw[off:].decode("utf-8")
        settings = json.loads(txt) if txt.strip() else None
    return SaveFile(X, Y, base, water, wall, droplets, stations, settings, version)


def load(path: str) -> SaveFile:
    with open(path, "rb") as f:
        return decode(f.read())


def js_number(x) -> str:
    """A number as JavaScript's Number::toString prints it -- what JSON.stringify(guiControls) writes into a save (app.js:6610):
    shortest round-trip digits, no ".0" on integers, decimal notation for 1e-7 < |x| < 1e21 (Python: 1e-05, JS: 0.00001),
    exponents as e-7 / e+21. Non-finite values become null, like JSON.stringify does."""
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, int) and abs(x) < 10 ** 21:
        return str(x)
    x = float(x)  # (integers from 1e21 on print in exponent form like any other Number: 1e+21)
    if x != x or x in (float("inf"), float("-inf")):
        return "null"
    if x == 0:
        return "0"
    sign = "-" if x < 0 else ""
    r = repr(abs(x))  # shortest round-trip digits, as ECMAScript requires
    mant, _, exp = r.partition("e")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    n = len(ip.lstrip("0")) if ip.strip("0") else -(len(fp) - len(fp.lstrip("0")))  # value = 0.digits * 10^n
    n += int(exp) if exp else 0
    digits = digits.rstrip("0") or "0"
    k = len(digits)
    if k <= n <= 21:
        return sign + digits + "0" * (n - k)
    if 0 < n <= 21:
        return sign + digits[:n] + "." + digits[n:]
    if -6 < n <= 0:
        return sign + "0." + "0" * (-n) + digits
    e = n - 1
    return sign + digits[0] + ("." + digits[1:] if k > 1 else "") + "e" + ("+" if e > 0 else "-") + str(abs(e))


def js_json(obj) -> str:
    """JSON.stringify(obj) (no indentation): key order as inserted, numbers by js_number, non-ASCII characters unescaped."""
    if obj is None:
        return "null"
    if isinstance(obj, (bool, int, float)):
        return js_number(obj)
    if isinstance(obj, str):
        return json.dumps(obj, ensure_ascii=False)
    if isinstance(obj, (list, tuple)):
        return "[" + ",".join(js_json(v) for v in obj) + "]"
    if isinstance(obj, dict):
        return "{" + ",".join(json.dumps(str(k), ensure_ascii=False) + ":" + js_json(v) for k, v in obj.items()) + "}"
    if isinstance(obj, np.generic):
        return js_json(obj.item())
    raise TypeError(f"not JSON serialisable: {type(obj)}")


def encode(sf: SaveFile, level: int = 6) -> bytes:
    """Byte layout of prepareDownload() (app.js:6610-6621); the settings block is formatted like JSON.stringify does, so that
    decode -> encode reproduces the payload of a file the reference wrote byte for byte (the deflate stream itself differs:
    pako and zlib choose different matches)."""
    parts = [
        struct.pack("<H", sf.X),
        struct.pack("<H", sf.Y),
        np.ascontiguousarray(sf.base, np.float32).tobytes(),
        np.ascontiguousarray(sf.water, np.float32).tobytes(),
        np.ascontiguousarray(sf.wall, np.int8).tobytes(),
        np.ascontiguousarray(sf.droplets, np.float32).tobytes(),
        struct.pack("<H", len(sf.stations)),
        np.asarray([c for xy in sf.stations for c in xy], np.int16).tobytes(),
        js_json(sf.settings if sf.settings is not None else {}).encode("utf-8"),
    ]
    return struct.pack("<I", SAVE_FILE_VERSION_ID) + zlib.compress(b"".join(parts), level)


def save(path: str, sf: SaveFile) -> None:
    with open(path, "wb") as f:
        f.write(encode(sf))
