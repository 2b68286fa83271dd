"""Column-slab decomposition of the periodic domain across GPUs (one process per GPU).

The reference is single-GPU (one GL context, app.js:3368); this is new work with no reference counterpart
(SURVEY.md 8e). The domain is periodic in x (REPEAT wrap), so rank r owns columns [r*Xo, (r+1)*Xo) and keeps
``halo`` ghost columns on each side. One iteration's dependency cone is <= 6 columns per side
(pressure 1 + advection 2 (|v| < 1) + boundary/vortForce 1 + vorticity 1 + curl/velocity 1), so with
``halo`` ghost columns the owned columns stay exact for ``halo // 6`` iterations; then the ``halo`` outermost
owned columns of the carried state (base_0, wall_0, water_1, both light textures: 68 B/cell) are sent to the
ring neighbours with point-to-point send/recv (RCCL over xGMI on GPUs, gloo in the CPU tests) and unpacked
into their ghost columns. There is no collective on the data path.

Overlap (``HipSlabEngine`` without particles): pack, send / recv and unpack run on a side stream (``wx_set_comm_stream``). The
last iteration before an exchange launches the edge strips first (``wx_step_overlap(..., EDGES_FIRST)``): the exchange starts as
soon as the columns the neighbours need are final and proceeds while the interior strips compute; the first iteration after it
launches the interior strips first and the edge strips -- the only ones that read ghost columns -- once the unpack event has
fired (``EDGES_LAST``). The host never blocks.

Particles (``n_droplets > 0``): the droplet pool is PARTITIONED (csrc/wx_kernels.h, ``SlabP``). An active droplet is tracked by the
rank whose owned columns contain it (and as a ghost copy by the neighbour while it is within ``halo`` columns of the common edge);
inactive droplets are static records every rank holds and tests against its own columns (their spawn probe hashes to anywhere in
the domain). At every halo exchange (which then also carries the feedback / deposition textures): an all-gather of the few hundred
droplets whose active / inactive status flipped during the period (32 B each; a rank that spawned a phantom from a stale record loses
it here), then the droplets near the slab edges travel to the ring neighbours in the SAME batch of send / recv as the grid halos
(24 B each). The 4-float lightning state is reconciled with two tiny all-reduces. No message grows with the pool size.

``SlabSim`` is written against a small engine interface so that the identical exchange logic runs on the HIP
engine (``HipSlabEngine``, device tensors + NCCL) and on a checker engine in the CPU tests (gloo).
"""
from __future__ import annotations

import contextlib
import os
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import params
from .engine import Handle

CONE_PER_ITERATION = 6  # columns per side, see module docstring
DEFAULT_HALO = 42       # seven iterations per exchange: amortises the ~0.1 ms P2P round trip, 4.1 % redundant columns at 2048/GPU.
                        # Not 48: a 4096 + 96 = 4192-column slab has a row pitch of 131 x 512 B, and the dry stencil runs 14 % slower on it
                        # than on 4180 or 4144 columns (profiles/r03_slab_shapes.txt) -- rows a power-of-two multiple apart share channels


def particle_period(halo: int) -> int:
    """Iterations per exchange of a slab with particles (WX_SLAB_PERIOD_PARTICLES)."""
    return 0 if halo < 12 else min(15, 1 + (halo - 12) // 9)


def slab_columns(X: int, rank: int, world: int):
    if X % world:
        raise ValueError(f"X={X} is not divisible by the number of slabs {world}")
    xo = X // world
    return rank * xo, xo


class HipSlabEngine:
    """One slab on one GPU through the C ABI (wx_create_slab / wx_halo_pack / wx_halo_unpack)."""

    def __init__(self, X_global: int, Y: int, x0: int, X_owned: int, halo: int, device: torch.device, n_droplets: int = 0,
                 rank: int = 0):
        self.device = device
        torch.cuda.set_device(device)
        self.n_droplets = n_droplets
        self.h = Handle(X_owned, Y, n_droplets, X_global=X_global, x0=x0, halo=halo)
        self.h.slab_set_rank(rank)
        # compute on torch's current stream; the halo exchange (pack -> send / recv -> unpack) on a side stream that the library
        # fences against compute with events. With particles the exchange stays in order on the compute stream: the pool
        # reconciliation needs the finished iteration anyway.
        self.h.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.comm = None
        if n_droplets == 0 and os.environ.get("WX_SLAB_OVERLAP", "1") != "0":
            self.comm = torch.cuda.Stream(device, priority=-1)  # (its small pack / unpack kernels run next to a marching kernel that holds every wave slot)
            self.h.set_comm_stream(self.comm.cuda_stream)
        self.nbytes = self.h.halo_bytes()
        self._light = None
        self._light_gen = -1

    @staticmethod
    def comm_unique_id(rank: int) -> bytes:
        """Rank 0 draws the 128-byte id of the library's RCCL communicator, torch.distributed only carries it to the others."""
        box = [Handle.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def comm_init_native(self, rank: int, world: int, uid: bytes):
        """RCCL inside the library (wx_comm_init)."""
        self.h.comm_init(uid, rank, world)

    @property
    def supports_overlap(self) -> bool:
        return self.comm is not None

    def comm_context(self):
        """Stream context for the host's send / recv calls: the stream pack and unpack run on."""
        return torch.cuda.stream(self.comm) if self.comm is not None else contextlib.nullcontext()

    def lightning_tensor(self) -> torch.Tensor:
        """Live device view of the handle's 4-float lightning state (x, y, start iteration, intensity): no host round trip."""
        if self._light is None or self._light_gen != self.h.generation:  # (wx_tune_placement moves the planes: cached views dangle)
            self._light_gen = self.h.generation

            class _Dev:  # the CUDA array interface is all torch needs to wrap a foreign device pointer
                def __init__(self, ptr):
                    self.__cuda_array_interface__ = {"shape": (4,), "typestr": "<f4", "data": (ptr, False), "version": 2}
            self._light = torch.as_tensor(_Dev(self.h.device_ptr("LIGHTNING")), device=self.device)
        return self._light

    def upload(self, base, water, wall, drops=None):
        self.h.upload(base, water, wall, drops)

    def water_free(self) -> bool:
        return self.h.water_free()

    def assert_water_free(self, agreed: bool):
        self.h.slab_assert_water_free(agreed)

    # partitioned droplet pool (device tensors in, device tensors out)
    def new_pool_buffers(self, world: int):
        """(my event buffer, the gathered event buffers of all ranks, [edge buffers to left / right], [from left / right])."""
        eb, gb = self.h.pool_event_bytes(), self.h.pool_edge_bytes()
        z = lambda n: torch.zeros(n, dtype=torch.uint8, device=self.device)
        return z(eb), z(eb * world), [z(gb), z(gb)], [z(gb), z(gb)]

    def pool_events_pack(self, buf: torch.Tensor):
        self.h.pool_events_pack(buf.data_ptr())

    def pool_events_apply(self, gathered: torch.Tensor, world: int, stride: int = 0):
        self.h.pool_events_apply(gathered.data_ptr(), world, stride)

    def pool_edges_pack(self, left: torch.Tensor, right: torch.Tensor, refresh_inactive: bool):
        self.h.pool_edges_pack(left.data_ptr(), right.data_ptr(), refresh_inactive)

    def pool_edges_apply(self, buf: torch.Tensor):
        self.h.pool_edges_apply(buf.data_ptr())

    def set_pool_exact(self, on: bool):
        self.h.set_option(Handle.OPT_POOL_EXACT, 1 if on else 0)

    def lightning(self):
        return self.h.lightning()

    def set_lightning(self, v):
        self.h.set_lightning(v)

    def period_begin(self):
        self.h.slab_period_begin()

    def set_params(self, u: Dict[str, Any]):
        self.h.set_params(params.fill_struct(params.WxParams(), u), u["initial_T"], u.get("sounding_T"), u.get("sounding_W"),
                          u.get("sounding_Vel"))

    def new_buffer(self) -> torch.Tensor:
        return torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)

    def message_bytes(self) -> int:
        """Bytes of a halo buffer that a message of the current period occupies (wx_halo_message_bytes: the base texture alone between
        slabs of the agreed water-free dry stencil); equal on every rank."""
        return self.h.halo_message_bytes()

    def pack(self, side: int, buf: torch.Tensor):
        self.h.halo_pack(side, buf.data_ptr())

    def unpack(self, side: int, buf: torch.Tensor):
        self.h.halo_unpack(side, buf.data_ptr())

    def pack_both(self, left: torch.Tensor, right: torch.Tensor):
        """Both edges in ONE launch: a second small kernel would queue behind the marching kernel's thousands of workgroups."""
        self.h.halo_pack_both(left.data_ptr(), right.data_ptr())

    def unpack_both(self, left: torch.Tensor, right: torch.Tensor):
        self.h.halo_unpack_both(left.data_ptr(), right.data_ptr())

    def step(self, n: int, overlap: int = 0):
        self.h.step(n, overlap)

    def sync(self):
        self.h.sync()

    # slabs exact at any speed (include/wxsim.h): the |vx| this slab measured / the bound all slabs agreed on
    def vx_take(self) -> float:
        return self.h.slab_vx_take()

    def set_vx_bound(self, v: float) -> int:
        """Sizes the coming exchange period by the maximum over all slabs; returns the iterations per exchange."""
        self.h.slab_set_vx_bound(v)
        return self.h.slab_period


EXACT_EVENTS_PER_ITERATION = 16384  # status flips a rank can report per iteration in exact mode (the all-gather has a fixed stride: no
                                     # host round trip for the counts); more is an error reported by the next blocking call


class SlabSim:
    """Drives one slab: ``step(n)`` = n iterations with a ring halo exchange every ``halo // 6`` iterations.

    ``exact`` (slabs with particles, WX_OPT_POOL_EXACT): status flips, lightning requests and the 600-iteration inactive count are
    all-gathered after EVERY iteration, so that the decomposed run equals the undecomposed one exactly (bit for bit with the
    deterministic splat order); the grid halos and edge droplets still travel once per period."""

    def __init__(self, engine, rank: int, world: int, halo: int, exact: bool = False):
        if world > 1 and halo < CONE_PER_ITERATION:
            raise ValueError(f"halo must be >= {CONE_PER_ITERATION}")
        self.engine, self.rank, self.world, self.halo = engine, rank, world, halo
        self.particles = getattr(engine, "n_droplets", 0) > 0
        # with particles the owned columns need a sprite radius (6 px) of valid ghost columns in the last iteration too
        self.iters_per_exchange = max(1, halo // CONE_PER_ITERATION)
        if self.particles:  # WX_SLAB_PERIOD_PARTICLES (include/wxsim.h): 6 columns for the first iteration, 9 for every further one, a sprite
            self.iters_per_exchange = particle_period(halo)  # radius left in the last; at most 15 (16-bit flip history)
        self.left, self.right = (rank - 1) % world, (rank + 1) % world
        self._since_exchange = 0
        self._iters = 0
        self._exchanged = False  # ghost columns of the current period came from an exchange (not from the upload)
        self._overlap = world > 1 and bool(getattr(engine, "supports_overlap", False))
        self.exact = bool(exact or os.environ.get("WX_SLAB_EXACT", "0") == "1") and self.particles and world > 1 and hasattr(engine, "set_pool_exact")
        if self.exact:
            engine.set_pool_exact(True)
        if world > 1:
            self.send = [engine.new_buffer(), engine.new_buffer()]  # [to left, to right]
            self.recv = [engine.new_buffer(), engine.new_buffer()]  # [from left, from right]
            if self.particles:
                self.ev, self.ev_all, self.psend, self.precv = engine.new_pool_buffers(world)
            # gloo has no device-tensor send/recv (it is the CPU-test / single-GPU plumbing transport): stage through the host
            self._stage = dist.get_backend() == "gloo" and self.send[0].is_cuda
            if self._stage:
                self._hsend = [torch.empty_like(b, device="cpu") for b in self.send]
                self._hrecv = [torch.empty_like(b, device="cpu") for b in self.recv]
            self.agree_water_free()
        # On GPUs the exchange runs INSIDE the library (wx_comm_init / wx_slab_step: pack -> ncclSend / ncclRecv -> unpack on the handle's
        # comm stream; with particles also the droplet-pool protocol: ncclAllGather of the status flips with a stride derived from the previous
        # period's counts, edge droplets in the halos' batch, all of it behind the interior strips of the next iteration); torch.distributed is then only the launcher and the carrier of the communicator's id. The host-driven
        # path below stays for the gloo transport of the CPU tests (and as the fallback).
        self._native = False
        self.transport = "host-driven (torch.distributed send / recv)"
        if (world > 1 and hasattr(engine, "comm_init_native") and dist.get_backend() == "nccl"
                and os.environ.get("WX_SLAB_NATIVE", "1") != "0"):
            # First contact insurance: communicator set-up and one real exchange (harmless right after an upload: the ghosts receive the
            # values they already hold) under a watchdog -- a transport that cannot be initialised falls back to the host-driven
            # exchange, one that hangs ends the process with a readable message instead of a silent timeout.
            import threading
            done, err = threading.Event(), []
            uid = engine.comm_unique_id(rank)  # (a torch collective: on the main thread, whose current device is this rank's)

            def first_contact():  # (library calls only: they make the handle's own device current themselves)
                try:
                    engine.comm_init_native(rank, world, uid)
                    engine.h.exchange()
                    engine.h.sync()
                except Exception as e:  # noqa: BLE001
                    err.append(e)
                done.set()
            threading.Thread(target=first_contact, daemon=True).start()
            if not done.wait(float(os.environ.get("WX_DIST_TIMEOUT_S", "180"))):
                print(f'{{"error": "rank {rank}: the in-library RCCL transport (wx_comm_init / wx_exchange) did not complete; set WX_SLAB_NATIVE=0 for the host-driven exchange"}}', flush=True)
                os._exit(3)
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self.send[0].device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # (all ranks or none)
            if int(ok.item()) == 1:
                self._native = True
                self.transport = "in-library RCCL (wx_comm_init / wx_slab_step: ncclSend / ncclRecv on the comm stream)"
            else:
                if err:
                    print(f"[slab] rank {rank}: in-library transport unavailable ({err[0]}); using the host-driven exchange", flush=True)
                if self.particles:  # wx_comm_init gave the handle a side stream for the pool exchange; the host-driven one runs in order
                    engine.h.set_comm_stream(0)

    def agree_water_free(self):
        """The water-free dry iteration (36 B/cell) is only valid on a slab if NO slab of the domain carries water (ghost columns
        flow in from the neighbours): all-reduce(MIN) of what each rank's upload found, once per upload -- so that no step ever
        has to look at the ghost columns on the host (the device validates them at every unpack, wx_slab_assert_water_free)."""
        e = self.engine
        if self.world == 1 or not hasattr(e, "water_free"):
            return
        dev = self.send[0].device if (self.send[0].is_cuda and not self._stage) else torch.device("cpu")
        t = torch.tensor([1 if e.water_free() else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        e.assert_water_free(bool(int(t.item())))

    def agree_vx_bound(self):
        """Slabs exact at any speed: one iteration invalidates 6 + floor|vx| ghost columns, so at the start of every exchange period the
        ranks all-reduce (MAX) the largest |vx| each of them measured and size the period by it (wx_slab_vx_take / wx_slab_set_vx_bound;
        a velocity that outruns the bound inside the period is reported by the next blocking call). The host-driven exchange pays a
        synchronisation per period for it; the in-library transport (wx_slab_step) carries the maxima with the exchange instead."""
        e = self.engine
        if self.world == 1 or not hasattr(e, "vx_take"):
            return
        dev = self.send[0].device if (self.send[0].is_cuda and not self._stage) else torch.device("cpu")
        t = torch.tensor([e.vx_take()], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.iters_per_exchange = max(1, e.set_vx_bound(float(t.item())))

    @property
    def handle(self):
        return getattr(self.engine, "h", self.engine)

    def _staged(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if self._stage else t

    def exchange(self):
        """Ring exchange: my left edge -> left neighbour's right ghosts, my right edge -> right neighbour's left ghosts. Everything
        is enqueued (on the engine's comm stream when it has one); nothing here waits on the host except the gloo staging path.
        With particles the droplet-pool exchange rides along: status-flip events (all-gather), then the edge droplets in the same
        batch of send / recv as the grid halos."""
        if self.world == 1:
            return
        e = self.engine
        ctx = e.comm_context() if hasattr(e, "comm_context") else contextlib.nullcontext()
        with ctx:
            if hasattr(e, "pack_both"):
                e.pack_both(self.send[0], self.send[1])
            else:
                e.pack(0, self.send[0])
                e.pack(1, self.send[1])
            if self.particles and self.exact:  # flips, lightning and the inactive count are current already (exact_events after every iteration)
                e.pool_edges_pack(self.psend[0], self.psend[1], False)
            elif self.particles:
                # 1. who flipped between active and inactive this period, and what did it become (a few hundred droplets)
                e.pool_events_pack(self.ev)
                # The buffer has room for every droplet (the start-up burst of an all-inactive pool), a normal period fills a few KB of it:
                # the counts travel first (16 B per rank; the one host synchronisation of an exchange), then only the filled part
                hdr = self.ev[:16].cpu() if self.ev.is_cuda else self.ev[:16].clone()
                hdrs = [torch.empty_like(hdr) for _ in range(self.world)]
                if dist.get_backend() == "nccl":
                    dh = [t.to(self.ev.device) for t in hdrs]
                    dist.all_gather(dh, hdr.to(self.ev.device))
                    hdrs = [t.cpu() for t in dh]
                else:
                    dist.all_gather(hdrs, hdr)
                most = max(int(t.view(torch.int32)[0]) for t in hdrs)
                stride = min(len(self.ev), (16 + most * 32 + 4095) // 4096 * 4096)
                mine, every = self.ev[:stride], self.ev_all[:stride * self.world]
                if self._stage:
                    parts = [torch.empty(stride, dtype=torch.uint8) for _ in range(self.world)]
                    dist.all_gather(parts, mine.cpu())
                    every.copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(every, mine)
                e.pool_events_apply(every, self.world, stride)
                # 2. ownership by position; the droplets near my edges become the neighbours' ghost copies
                refresh = (self._iters // 600) != ((self._iters - self._since_exchange) // 600)  # app.js:5957-5966: every 600 iterations
                e.pool_edges_pack(self.psend[0], self.psend[1], refresh)
            mb = e.message_bytes() if hasattr(e, "message_bytes") else len(self.send[0])  # (what the packed message occupies of its buffer)
            out = [self.send[0][:mb], self.send[1][:mb]] + (self.psend if self.particles else [])
            inn = [self.recv[0][:mb], self.recv[1][:mb]] + (self.precv if self.particles else [])
            if self._stage:
                e.sync()
                out = [t.cpu() for t in out]
                hin = [torch.empty_like(t, device="cpu") for t in inn]
            else:
                hin = inn
            ops = []
            for k in range(0, len(out), 2):  # pairs (to / from left, to / from right)
                if self.world == 2:
                    # both neighbours are the same rank: order the two messages identically on both sides
                    ops += [dist.P2POp(dist.isend, out[k], self.left), dist.P2POp(dist.isend, out[k + 1], self.right),
                            dist.P2POp(dist.irecv, hin[k + 1], self.right), dist.P2POp(dist.irecv, hin[k], self.left)]
                else:
                    ops += [dist.P2POp(dist.isend, out[k], self.left), dist.P2POp(dist.irecv, hin[k + 1], self.right),
                            dist.P2POp(dist.isend, out[k + 1], self.right), dist.P2POp(dist.irecv, hin[k], self.left)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()  # NCCL / RCCL: orders the current (comm) stream behind the transfer, does not block the host
            if self._stage:
                for hb, b in zip(hin, inn):
                    b.copy_(hb)
            if hasattr(e, "unpack_both"):  # left ghosts <- left neighbour's right edge, right ghosts <- right neighbour's left edge
                e.unpack_both(self.recv[0], self.recv[1])
            else:
                e.unpack(0, self.recv[0])
                e.unpack(1, self.recv[1])
            if self.particles:
                e.pool_edges_apply(self.precv[0])
                e.pool_edges_apply(self.precv[1])
        if self.particles:
            if not self.exact:
                self.reconcile_lightning()
            e.period_begin()
        self._since_exchange = 0
        self._exchanged = True

    def exact_events(self):
        """Exact mode, after every iteration: all-gather of every rank's status flips + iteration record (lightning request, deposit at
        texel (0,0)) with a FIXED stride -- nothing waits on the host -- and their application (wx_pool_events_pack / _apply)."""
        e = self.engine
        e.pool_events_pack(self.ev)
        stride = min(len(self.ev), (16 + 32 * (1 + EXACT_EVENTS_PER_ITERATION) + 4095) // 4096 * 4096)
        mine, every = self.ev[:stride], self.ev_all[:stride * self.world]
        if self._stage:
            parts = [torch.empty(stride, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu())
            every.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(every, mine)
        e.pool_events_apply(every, self.world, stride)

    def reconcile_lightning(self):
        """Every rank's lightning state becomes the latest strike any rank registered: MAX over a strike key, SUM of the winner's four
        floats. Device tensors only: no host round trip.

        Strike key = start iteration << 10 | (1023 - rank) in an int64 tensor (the iteration is never wrapped): the latest strike wins;
        two ranks that registered different strikes in the same iteration are told apart by the rank, so exactly one of them
        contributes (no averaging of two bolts)."""
        e = self.engine
        live = hasattr(e, "lightning_tensor")
        dev = self.send[0].device if (self.send[0].is_cuda and not self._stage) else torch.device("cpu")
        light = e.lightning_tensor() if live else torch.from_numpy(np.asarray(e.lightning(), np.float32).copy())
        lw = light.to(dev)
        it = lw[2].to(torch.int64)
        my_key = torch.where(lw[2] > 0, it * 1024 + (1023 - self.rank), torch.zeros_like(it)).reshape(1)
        win_key = my_key.clone()
        dist.all_reduce(win_key, op=dist.ReduceOp.MAX)
        mine = ((my_key > 0) & (my_key == win_key)).to(torch.float32)[0]
        tail = torch.cat([mine.reshape(1), lw * mine])
        dist.all_reduce(tail, op=dist.ReduceOp.SUM)
        new_light = torch.where(tail[0] > 0, tail[1:], lw)  # exactly one rank contributed the winning strike
        if live:
            light.copy_(new_light.to(light.device))
        else:
            e.set_lightning(new_light.cpu().numpy())

    def step(self, n: int):
        if self._native:  # periods, launch order around the exchange, the |vx| bound and the exchange itself: wx_slab_step
            self.engine.h.slab_step(n)
            self._iters += n
            return
        done = 0
        while done < n:
            if self._since_exchange == 0:
                self.agree_vx_bound()
            k = min(self.iters_per_exchange - self._since_exchange, n - done)
            more = done + k < n and hasattr(self.engine, "h")  # WX_OVERLAP_MORE_TO_COME: only the last piece of this step stores the display-side fields
            if self.exact:  # one iteration per call, each followed by the events of all ranks
                for j in range(k):
                    if hasattr(self.engine, "h") and done + j + 1 < n:
                        self.engine.step(1, 4)  # (WX_OVERLAP_MORE_TO_COME)
                    else:
                        self.engine.step(1)
                    self.exact_events()
            elif self._overlap:
                # first iteration after an exchange: interior strips first, edge strips once the ghosts have arrived;
                # last iteration before one: edge strips first, so that the exchange starts while the interior computes
                flags = (2 if (self._since_exchange == 0 and self._exchanged) else 0) | (1 if self._since_exchange + k >= self.iters_per_exchange else 0)
                self.engine.step(k, flags | (4 if more else 0))
            elif more:
                self.engine.step(k, 4)
            else:
                self.engine.step(k)
            done += k
            self._since_exchange += k
            self._iters += k
            if self._since_exchange >= self.iters_per_exchange:
                self.exchange()

    def sync(self):
        self.engine.sync()

    def upload(self, base, water, wall, drops=None):
        """Re-upload this rank's slab (local arrays incl. ghost columns): the ghost columns are fresh again, so the exchange
        period starts over and the next step does not run the interior-first split against an exchange that never happened."""
        self.engine.upload(base, water, wall, drops)
        self.agree_water_free()
        self._since_exchange = 0
        self._exchanged = False
        if self.particles and hasattr(self.engine, "period_begin"):
            self.engine.period_begin()

    # ---- construction on the HIP engine ----
    @classmethod
    def from_generator(cls, pkg, X: int, Y: int, u: Dict[str, Any], rank: int, world: int, device: torch.device,
                       halo: int = DEFAULT_HALO, drops=None, cloud_deck: bool = False, exact: bool = False) -> "SlabSim":
        """Each rank fills only its own slab (plus ghost columns) of the synthetic terrain grid, on the device
        (wx_setup_columns); ``drops`` is the WHOLE droplet pool (identical on every rank)."""
        x0, xo = slab_columns(X, rank, world)
        eng = HipSlabEngine(X, Y, x0, xo, halo, device, 0 if drops is None else len(drops), rank)
        eng.h.setup_columns(pkg.synth.terrain_columns(X, Y, cols=(x0 - halo, xo + 2 * halo), cloud_deck=cloud_deck), drops)
        eng.set_params(u)
        return cls(eng, rank, world, halo, exact=exact)

    @classmethod
    def from_dry_generator(cls, pkg, X: int, Y: int, u: Dict[str, Any], rank: int, world: int, device: torch.device,
                           halo: int = DEFAULT_HALO, flow_sigma: float = 0.0) -> "SlabSim":
        """BASELINE configs[1]'s dry-air state (synth.dry_grid: every value a function of the GLOBAL column), this rank's slab only."""
        x0, xo = slab_columns(X, rank, world)
        eng = HipSlabEngine(X, Y, x0, xo, halo, device, 0, rank)
        eng.upload(*pkg.synth.dry_grid(X, Y, cols=(x0 - halo, xo + 2 * halo), flow_sigma=flow_sigma))
        eng.set_params(u)
        return cls(eng, rank, world, halo)

    @classmethod
    def from_arrays(cls, X: int, Y: int, base, water, wall, u: Dict[str, Any], rank: int, world: int, device: torch.device,
                    halo: int = DEFAULT_HALO, drops=None, exact: bool = False) -> "SlabSim":
        """Cut this rank's slab out of whole-domain arrays (Y, X, 4); ``drops`` is the whole droplet pool."""
        x0, xo = slab_columns(X, rank, world)
        idx = (x0 - halo + np.arange(xo + 2 * halo)) % X
        eng = HipSlabEngine(X, Y, x0, xo, halo, device, 0 if drops is None else len(drops), rank)
        eng.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]), drops)
        eng.set_params(u)
        return cls(eng, rank, world, halo, exact=exact)

    def gather_particles(self) -> Optional[np.ndarray]:
        """The whole droplet pool (n_droplets x 5) assembled on rank 0 (None elsewhere): an active droplet's record comes from the rank
        that has it in its owned columns, an inactive one's is the same on every rank (wx_pool_flags). Host-side, for readback / saves."""
        h = self.handle
        if self.world > 1 and self._native:
            h.exchange()  # settle ownership: right after an exchange every active droplet has exactly one owner
        elif self.world > 1 and self._since_exchange != 0:
            # inside a period a droplet can be "owned" by two ranks at once (a phantom spawn from a stale record, resolved by the next
            # exchange): the sum below would then add two records
            raise RuntimeError("gather_particles: call it right after an exchange (step a whole number of exchange periods)")
        d, f = h.read_particles(), h.pool_flags()
        if self.world == 1:
            return d
        # (RCCL reduces device tensors only; gloo -- the CPU tests -- host tensors)
        dev = self.send[0].device if (self.send[0].is_cuda and not self._stage) else torch.device("cpu")
        mine = torch.from_numpy(np.where((f == 2)[:, None], d, 0).astype(np.float32)).to(dev)
        cnt = torch.from_numpy((f == 2).astype(np.int32)).to(dev)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)  # (exactly one owner per active droplet right after an exchange: the sum IS its record)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        cnt, mine = cnt.cpu().numpy(), mine.cpu().numpy()
        if cnt.max(initial=0) > 1:
            raise RuntimeError("gather_particles: a droplet is owned by more than one rank")
        out = np.where((cnt > 0)[:, None], mine, d)  # nobody owns it: inactive, my own record is everybody's
        return out if self.rank == 0 else None

    def owned(self, field: str) -> np.ndarray:
        """This rank's owned columns of a field (host array)."""
        h = self.handle
        return h.read_rect(field, self.halo, 0, h.X - 2 * self.halo, h.Y)
