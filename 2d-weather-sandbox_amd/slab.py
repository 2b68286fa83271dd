"""Column-slab decomposition of the periodic domain across GPUs (one process per GPU).

The reference is single-GPU (one GL context, app.js:3368); this is new work with no reference counterpart
(SURVEY.md 8e). The domain is periodic in x (REPEAT wrap), so rank r owns columns [r*Xo, (r+1)*Xo) and keeps
``halo`` ghost columns on each side. One iteration's dependency cone is <= 6 columns per side
(pressure 1 + advection 2 (|v| < 1) + boundary/vortForce 1 + vorticity 1 + curl/velocity 1), so with
``halo`` ghost columns the owned columns stay exact for ``halo // 6`` iterations; then the ``halo`` outermost
owned columns of the carried state (base_0, wall_0, water_1, both light textures: 68 B/cell) are sent to the
ring neighbours with point-to-point send/recv (RCCL over xGMI on GPUs, gloo in the CPU tests) and unpacked
into their ghost columns. There is no collective on the data path.

``SlabSim`` is written against a small engine interface so that the identical exchange logic runs on the HIP
engine (``HipSlabEngine``, device tensors + NCCL) and on a checker engine in the CPU tests (gloo).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import params
from .engine import Handle

CONE_PER_ITERATION = 6  # columns per side, see module docstring
DEFAULT_HALO = 48       # eight iterations per exchange: amortises the ~0.1 ms P2P round trip, 4.7 % redundant columns at 2048/GPU


def slab_columns(X: int, rank: int, world: int):
    if X % world:
        raise ValueError(f"X={X} is not divisible by the number of slabs {world}")
    xo = X // world
    return rank * xo, xo


class HipSlabEngine:
    """One slab on one GPU through the C ABI (wx_create_slab / wx_halo_pack / wx_halo_unpack)."""

    def __init__(self, X_global: int, Y: int, x0: int, X_owned: int, halo: int, device: torch.device):
        self.device = device
        torch.cuda.set_device(device)
        self.h = Handle(X_owned, Y, 0, X_global=X_global, x0=x0, halo=halo)
        # run the kernels on torch's current stream so that they are ordered with the NCCL send/recv ops
        self.h.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.nbytes = self.h.halo_bytes()

    def upload(self, base, water, wall):
        self.h.upload(base, water, wall)

    def set_params(self, u: Dict[str, Any]):
        self.h.set_params(params.fill_struct(params.WxParams(), u), u["initial_T"], u.get("sounding_T"), u.get("sounding_W"),
                          u.get("sounding_Vel"))

    def new_buffer(self) -> torch.Tensor:
        return torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)

    def pack(self, side: int, buf: torch.Tensor):
        self.h.halo_pack(side, buf.data_ptr())

    def unpack(self, side: int, buf: torch.Tensor):
        self.h.halo_unpack(side, buf.data_ptr())

    def step(self, n: int):
        self.h.step(n)

    def sync(self):
        self.h.sync()


class SlabSim:
    """Drives one slab: ``step(n)`` = n iterations with a ring halo exchange every ``halo // 6`` iterations."""

    def __init__(self, engine, rank: int, world: int, halo: int):
        if world > 1 and halo < CONE_PER_ITERATION:
            raise ValueError(f"halo must be >= {CONE_PER_ITERATION}")
        self.engine, self.rank, self.world, self.halo = engine, rank, world, halo
        self.iters_per_exchange = max(1, halo // CONE_PER_ITERATION)
        self.left, self.right = (rank - 1) % world, (rank + 1) % world
        self._since_exchange = 0
        if world > 1:
            self.send = [engine.new_buffer(), engine.new_buffer()]  # [to left, to right]
            self.recv = [engine.new_buffer(), engine.new_buffer()]  # [from left, from right]

    @property
    def handle(self):
        return getattr(self.engine, "h", self.engine)

    def exchange(self):
        """Ring exchange: my left edge -> left neighbour's right ghosts, my right edge -> right neighbour's left ghosts."""
        if self.world == 1:
            return
        e = self.engine
        e.pack(0, self.send[0])
        e.pack(1, self.send[1])
        if self.world == 2:
            # both neighbours are the same rank: order the two messages identically on both sides
            ops = [dist.P2POp(dist.isend, self.send[0], self.left), dist.P2POp(dist.isend, self.send[1], self.right),
                   dist.P2POp(dist.irecv, self.recv[1], self.right), dist.P2POp(dist.irecv, self.recv[0], self.left)]
        else:
            ops = [dist.P2POp(dist.isend, self.send[0], self.left), dist.P2POp(dist.irecv, self.recv[1], self.right),
                   dist.P2POp(dist.isend, self.send[1], self.right), dist.P2POp(dist.irecv, self.recv[0], self.left)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        e.unpack(0, self.recv[0])  # left ghosts  <- left neighbour's right edge
        e.unpack(1, self.recv[1])  # right ghosts <- right neighbour's left edge
        self._since_exchange = 0

    def step(self, n: int):
        done = 0
        while done < n:
            k = min(self.iters_per_exchange - self._since_exchange, n - done)
            self.engine.step(k)
            done += k
            self._since_exchange += k
            if self._since_exchange >= self.iters_per_exchange:
                self.exchange()

    def sync(self):
        self.engine.sync()

    # ---- construction on the HIP engine ----
    @classmethod
    def from_generator(cls, pkg, X: int, Y: int, u: Dict[str, Any], rank: int, world: int, device: torch.device,
                       halo: int = DEFAULT_HALO) -> "SlabSim":
        """Each rank generates only its own slab (plus ghost columns) of the synthetic terrain grid."""
        x0, xo = slab_columns(X, rank, world)
        eng = HipSlabEngine(X, Y, x0, xo, halo, device)
        base, water, wall = pkg.synth.terrain_grid(X, Y, cols=(x0 - halo, xo + 2 * halo))
        eng.upload(base, water, wall)
        eng.set_params(u)
        return cls(eng, rank, world, halo)

    @classmethod
    def from_arrays(cls, X: int, Y: int, base, water, wall, u: Dict[str, Any], rank: int, world: int, device: torch.device,
                    halo: int = DEFAULT_HALO) -> "SlabSim":
        """Cut this rank's slab out of whole-domain arrays (Y, X, 4)."""
        x0, xo = slab_columns(X, rank, world)
        idx = (x0 - halo + np.arange(xo + 2 * halo)) % X
        eng = HipSlabEngine(X, Y, x0, xo, halo, device)
        eng.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
        eng.set_params(u)
        return cls(eng, rank, world, halo)

    def owned(self, field: str) -> np.ndarray:
        """This rank's owned columns of a field (host array)."""
        h = self.handle
        return h.read_rect(field, self.halo, 0, h.X - 2 * self.halo, h.Y)
