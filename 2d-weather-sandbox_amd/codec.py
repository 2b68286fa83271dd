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


def encode(sf: SaveFile, level: int = 6) -> bytes:
    """Byte layout of prepareDownload() (app.js:6610-6621)."""
    parts = [
        struct.pack("<H", sf.X),
        struct.pack("<H", sf.Y),
        np.ascontiguousarray(sf.base, np.float32).tobytes(),
        np.ascontiguousarray(sf.water, np.float32).tobytes(),
        np.ascontiguousarray(sf.wall, np.int8).tobytes(),
        np.ascontiguousarray(sf.droplets, np.float32).tobytes(),
        struct.pack("<H", len(sf.stations)),
        np.asarray([c for xy in sf.stations for c in xy], np.int16).tobytes(),
        json.dumps(sf.settings if sf.settings is not None else {}, separators=(",", ":")).encode("utf-8"),
    ]
    return struct.pack("<I", SAVE_FILE_VERSION_ID) + zlib.compress(b"".join(parts), level)


def save(path: str, sf: SaveFile) -> None:
    with open(path, "wb") as f:
        f.write(encode(sf))
