"""Device-side helpers for benchmarks and self-verification (PyTorch is plumbing here: it wraps the engine's own device buffers).

* ``field_tensor``  -- a live torch view of a field's current device storage (``wx_device_ptr``), no copy;
* ``seed_flow``     -- adds a seeded velocity field (smooth divergence-free eddies, or white noise) to the air cells on the device,
  a pure function of the GLOBAL cell coordinates (slabs of a decomposed domain and the undecomposed handle get bit-identical
  values): turns the still start state of a synthetic grid into a moving fluid without a 1.2 GB host upload;
* ``flow_stats``    -- rms / max |v| and the share of cells whose back-trace leaves the lane's own cell;
* ``checksum``      -- position-weighted integer checksum of the owned columns of a field, again keyed by the global cell index:
  N slab handles and the undecomposed handle produce the same numbers iff the fields are bit-identical.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .engine import FIELD_CHANNELS, Handle

_INT8_FIELDS = ("WALL_CUR", "WALL_DISP")


class _DevArray:
    """The CUDA array interface is all torch needs to wrap a foreign device pointer."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def field_tensor(h: Handle, field: str) -> torch.Tensor:
    """(Y, X_local, channels) view of the field's CURRENT storage. The engine rotates its buffers every iteration: take a fresh
    view after each step, and order your work after the handle's stream (``h.sync()`` or the same torch stream)."""
    ptr = h.device_ptr(field)
    if not ptr:
        raise RuntimeError(f"wx_device_ptr({field}) returned NULL")
    ch = FIELD_CHANNELS.get(field, 4)
    typestr = "|i1" if field in _INT8_FIELDS else "<f4"
    return torch.as_tensor(_DevArray(ptr, (h.Y, h.X, ch), typestr), device=torch.device("cuda", torch.cuda.current_device()))


def _hash32(x: torch.Tensor) -> torch.Tensor:
    """Integer mixing on int64 lanes restricted to 32 bits (deterministic on every device)."""
    m = 0xFFFFFFFF
    x = x & m
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & m
    x = ((x ^ (x >> 15)) * 0x846CA68B) & m
    return (x ^ (x >> 16)) & m


def _global_index(h: Handle, Xg: int, x0: int, halo: int, cols: slice, device) -> torch.Tensor:
    """int64 (Y, n) global cell indices y * Xg + gx of the local columns in ``cols``."""
    lc = torch.arange(h.X, device=device, dtype=torch.int64)[cols]
    gx = (lc + (x0 - halo)) % Xg
    gy = torch.arange(h.Y, device=device, dtype=torch.int64)[:, None]
    return gy * Xg + gx[None, :]


def seed_flow(h: Handle, sigma: float, seed: int = 1, Xg: Optional[int] = None, x0: int = 0, kind: str = "eddies") -> None:
    """Adds a velocity field with standard deviation ``sigma`` cells / iteration to base_0.xy of every air cell above row 0, clipped
    to |v| <= 0.9 (the range the slab decomposition's 6-column cone assumes). In place on the device; the handle must be idle.

    ``kind="eddies"`` (default): a developed flow -- the discrete curl of a stream function made of a dozen seeded Fourier modes
    (eddies of 50 .. 400 cells, zero at the floor and the top), i.e. smooth and divergence-free ON THE STAGGERED GRID
    (vx = psi(x, y) - psi(x, y-1) on the right face, vy = -(psi(x, y) - psi(x-1, y)) on the top face, pressureShader.frag:16-43's
    divergence of it is zero), so the pressure pass does not radiate it away: it keeps moving through a benchmark run.
    ``kind="noise"``: independent U(-a, a) per cell and component -- every lane's back-trace footprint differs from its neighbour's:
    the worst case for the kernels' ring reads (and a field the pressure pass kills within a few hundred iterations)."""
    Xg = h.X if Xg is None else Xg
    h.sync()
    dev = torch.device("cuda", torch.cuda.current_device())
    base = field_tensor(h, "BASE_CUR")
    air = field_tensor(h, "WALL_CUR")[..., 1] != 0
    air[0, :] = False
    if kind == "noise":
        idx = _global_index(h, Xg, x0, h.halo, slice(None), dev)
        a = float(sigma) * 3.0 ** 0.5
        for c in (0, 1):
            u = (_hash32(_hash32(idx * 2 + c) + seed * 0x9E3779B1) >> 8).to(torch.float32) * (1.0 / (1 << 24))  # [0, 1), exact in fp32
            v = ((u - 0.5) * (2.0 * a)).clamp_(-0.9, 0.9)
            base[..., c] += torch.where(air, v, torch.zeros_like(v))
    else:
        rng = np.random.Generator(np.random.Philox(seed))
        Y = h.Y
        gx = ((torch.arange(h.X, device=dev, dtype=torch.int64) + (x0 - h.halo)) % Xg).to(torch.float64)  # global column
        gy = torch.arange(Y, device=dev, dtype=torch.float64)

        def psi(xs, ys):  # stream function at the cell corners (xs + 1/2, ys + 1/2); float64, a pure function of the global coordinates
            out = torch.zeros((len(ys), len(xs)), device=dev, dtype=torch.float64)
            r = np.random.Generator(np.random.Philox(seed))
            for _ in range(12):
                lam = float(r.uniform(50.0, 400.0))
                m = max(1, round(Xg / lam))  # whole periods around the periodic domain
                n = max(1, round(2.0 * Y / float(r.uniform(50.0, 400.0))))  # half periods between floor and top
                ph = float(r.uniform(0.0, 2.0 * np.pi))
                amp = 1.0 / np.hypot(2.0 * np.pi * m / Xg, np.pi * n / Y)  # every mode contributes the same velocity amplitude
                out += amp * torch.sin(2.0 * np.pi * m * (xs + 0.5) / Xg + ph)[None, :] * torch.sin(np.pi * n * (ys + 0.5) / Y)[:, None]
            return out

        p11, p10, p01 = psi(gx, gy), psi(gx, gy - 1.0), psi(gx - 1.0, gy)
        vx, vy = p11 - p10, -(p11 - p01)
        del rng
        scale = float(sigma) / (12 ** 0.5 * 0.5)  # 12 modes of velocity amplitude ~1 per component and rms 1/2 each
        for c, v in ((0, vx), (1, vy)):
            v = (v * scale).clamp_(-0.9, 0.9).to(torch.float32)
            base[..., c] += torch.where(air, v, torch.zeros_like(v))
    torch.cuda.synchronize()


def seed_vortices(h: Handle, n: int, peak: float = 1.3, radius: float = 6.0, seed: int = 5) -> None:
    """Adds ``n`` compact vortices (synth.add_vortices: curls of Gaussian stream-function blobs, divergence-free on the staggered grid,
    tangential speed ``peak`` cells / iteration at r = ``radius``) at seeded places of a WHOLE-DOMAIN handle, on the device, unclamped --
    the reference has no velocity clamp (advectionShader.frag:85-99): cells with |v| >= 0.9 are what the kernels' exact paths are for."""
    if h.halo:
        raise ValueError("seed_vortices: whole-domain handles only")
    h.sync()
    dev = torch.device("cuda", torch.cuda.current_device())
    base = field_tensor(h, "BASE_CUR")
    wall = field_tensor(h, "WALL_CUR")
    rng = np.random.Generator(np.random.Philox(seed))
    A = float(peak) * float(radius) * float(np.exp(0.5))
    R = int(np.ceil(6 * radius)) + 2
    for k in range(n):
        cx = float(rng.uniform(R + 1, h.X - R - 2))
        cy = float(rng.uniform(R + 2, h.Y - R - 2))
        sg = 1.0 if k % 2 == 0 else -1.0
        y0, x0 = int(cy) - R, int(cx) - R
        ys = torch.arange(y0, y0 + 2 * R + 1, device=dev, dtype=torch.float64)
        xs = torch.arange(x0, x0 + 2 * R + 1, device=dev, dtype=torch.float64)

        def psi(xx, yy):
            r2 = (xx[None, :] + 0.5 - cx) ** 2 + (yy[:, None] + 0.5 - cy) ** 2
            return sg * A * torch.exp(-r2 / (2.0 * radius * radius))

        p11, p10, p01 = psi(xs, ys), psi(xs, ys - 1.0), psi(xs - 1.0, ys)
        win = (slice(y0, y0 + 2 * R + 1), slice(x0, x0 + 2 * R + 1))
        air = wall[win][..., 1] != 0
        for c, v in ((0, p11 - p10), (1, -(p11 - p01))):
            v = v.to(torch.float32)
            base[win][..., c] += torch.where(air, v, torch.zeros_like(v))
    torch.cuda.synchronize()


def flow_stats(h: Handle) -> Dict[str, float]:
    """|v| statistics of the owned columns' air cells (post-pressure state)."""
    h.sync()
    cols = slice(h.halo, h.X - h.halo)
    b = field_tensor(h, "BASE_CUR")[:, cols]
    air = field_tensor(h, "WALL_CUR")[:, cols, 1] != 0
    v = torch.sqrt(b[..., 0] ** 2 + b[..., 1] ** 2)[air]
    if v.numel() == 0:
        return {"rms_v": 0.0, "max_v": 0.0, "frac_v_gt_0.05": 0.0, "cells_component_ge_0.9": 0}
    big = (torch.maximum(b[..., 0].abs(), b[..., 1].abs()) >= 0.9) & air  # cells whose back-trace may leave the 3x3 neighbourhood (exact path)
    return {"rms_v": float(torch.sqrt((v.double() ** 2).mean())), "max_v": float(v.max()), "frac_v_gt_0.05": float((v > 0.05).double().mean()),
            "cells_component_ge_0.9": int(big.sum())}


def checksum(h: Handle, fields: Sequence[str], Xg: Optional[int] = None, x0: int = 0, cols: Optional[slice] = None) -> np.ndarray:
    """int64 [len(fields), 2]: for every field the sums over (cell, channel) of the low / high 16 bits of the raw 32-bit (8-bit)
    patterns, each weighted by 1 + (hash of the global cell index and channel) % 251 -- small enough that int64 never overflows up to
    2^29 values. ``cols``: local columns to cover (default: the owned ones)."""
    Xg = h.X if Xg is None else Xg
    h.sync()
    dev = torch.device("cuda", torch.cuda.current_device())
    cols = slice(h.halo, h.X - h.halo) if cols is None else cols
    idx_all = _global_index(h, Xg, x0, h.halo, cols, dev)
    out = np.zeros((len(fields), 2), np.int64)
    for k, f in enumerate(fields):
        full = field_tensor(h, f)[:, cols]
        for y0 in range(0, h.Y, 512):  # row chunks: the int64 temporaries stay small next to a 33 GB state
            t, idx = full[y0:y0 + 512], idx_all[y0:y0 + 512]
            if t.dtype == torch.float32:
                bits = t.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
            else:
                bits = t.to(torch.int64) & 0xFF
            ch = torch.arange(bits.shape[-1], device=dev, dtype=torch.int64)
            w = 1 + _hash32(idx[..., None] * 4 + ch) % 251
            out[k, 0] += int(((bits & 0xFFFF) * w).sum())
            out[k, 1] += int(((bits >> 16) * w).sum())
            del bits, w, t
    return out
