#!/usr/bin/env python3
"""What the exchange PROTOCOL costs one rank of an 8-GPU run, measured on one GPU that the slab has to itself: ONE slab of X/8 owned
columns whose neighbours are itself (a periodic domain X/8 wide + ghost columns; wx_step_overlap with the edge-first / interior-first
flags, wx_halo_pack / wx_halo_unpack on a comm stream as slab.py issues them -- every launch an N-GPU rank makes, minus the link), against the same handle
stepped without any exchange (what tools/slab_shapes.py times) -- ONE handle for all modes, so that the placement lottery drops out. Usage: python tools/slab_protocol_cost.py [X_global Y halo reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
E = pkg.engine
from weather_sandbox_amd import devtools  # noqa: E402

XG = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
HALO = int(sys.argv[3]) if len(sys.argv) > 3 else 42
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NSLAB = int(os.environ.get("NSLAB", "8"))  # (the slab of one rank of an NSLAB-GPU run)
XO = XG // NSLAB
WORK = os.environ.get("WORKLOAD", "wet")


def uniforms():
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, **({"pass_mask": pkg.params.PASS_DRY} if WORK == "dry" else {}))
    u["enablePrecipitation"] = 0
    return u


def fill(h, Xg, cols):
    if WORK == "dry":
        h.upload(*pkg.synth.dry_grid(Xg, Y, cols=cols))
        h.slab_assert_water_free(True)  # (what the ranks agree on after an upload: the water-free marching kernel)
    else:
        h.setup_columns(pkg.synth.terrain_columns(Xg, Y, cols=cols))


u = uniforms()
p = pkg.params.fill_struct(pkg.params.WxParams(), u)
import ctypes  # noqa: E402


class SelfExchange:
    """ONE handle (one placement of the planes: handles of one process differ by +-8 % from the placement alone, section 4 of LABNOTES) run in
    four modes: no exchange at all (`plain`: what a slab's kernels cost), the overlapped protocol as one ordered launch per split iteration
    (round 5) or as two launch groups on two streams (rounds 2-4), and the protocol in order on the compute stream."""

    def __init__(self):
        self.h = E.Handle(XO, Y, 0, X_global=XO, x0=0, halo=HALO)
        fill(self.h, XO, (XO - HALO, XO + 2 * HALO))
        self.h.set_params(p, u["initial_T"])
        devtools.seed_flow(self.h, 0.2, Xg=XO, x0=0)
        self.comm = torch.cuda.Stream(priority=-1)
        nb = self.h.halo_bytes()
        self.buf = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
        # (this tool drives wx_step_overlap / wx_halo_* itself: it tells the handle once that the seeded flow stays below 1.5 cells / iteration
        # -- 7 ghost columns per iteration -- and sizes its periods by that)
        self.h.slab_set_vx_bound(1.0)
        self.ipe = max(1, self.h.slab_period)
        self.mode = None

    def set_mode(self, mode):  # "plain" | "one" | "two" | "inorder"
        self.h.sync()
        torch.cuda.synchronize()
        self.h.set_comm_stream(self.comm.cuda_stream if mode in ("one", "two", "two_a") else 0)
        if mode in ("one", "two", "two_a"):
            self.h.set_option(E.Handle.OPT_SPLIT_LAUNCH, 1 if mode == "one" else 0)
        self.mode, self.exchanged, self.since = mode, False, 0

    def step(self, n):
        if self.mode == "plain":
            self.h.step(n)
            return
        overlap = self.mode in ("one", "two", "two_a")
        b_split = 0 if self.mode == "two_a" else 2  # (two_a: only the iteration BEFORE an exchange is split; the one after it waits for the unpack)
        done = 0
        while done < n:
            k = min(self.ipe - self.since, n - done)
            last = self.since + k >= self.ipe
            flags = ((b_split if (self.exchanged and self.since == 0) else 0) | (1 if last else 0)) if overlap else 0
            self.h.step(k, flags | (4 if done + k < n else 0))  # (WX_OVERLAP_MORE_TO_COME: only the frame's last piece stores the display-side fields)
            done += k
            self.since += k
            if not last:
                continue
            self.since = 0
            self.h.halo_pack_both(self.buf[0].data_ptr(), self.buf[1].data_ptr())
            self.h.halo_unpack_both(self.buf[1].data_ptr(), self.buf[0].data_ptr())  # my right edge is my left neighbour's right edge: my own left ghosts
            self.exchanged = True

    def sync(self):
        self.h.sync()
        torch.cuda.synchronize()


sx = SelfExchange()


def timed(mode, iters=210, frame=7):
    sx.set_mode(mode)
    sx.step(2 * frame)
    sx.sync()
    t0 = time.perf_counter()
    for _ in range(iters // frame):
        sx.step(frame)
    sx.sync()
    return (time.perf_counter() - t0) / iters * 1e3


MODES = {"plain": "plain (no exchange)", "one": "protocol, overlapped: ONE ordered launch (r5)", "two": "protocol, overlapped: two launch groups (r4)",
         "two_a": "  the same, only the iteration BEFORE split",
         "inorder": "protocol, in order"}
for m in MODES:
    timed(m)
if int(os.environ.get("TUNE", "8")):
    sx.set_mode("plain")
    sx.h.tune_placement(int(os.environ.get("TUNE", "8")))
res = {v: [] for v in MODES.values()}
for _ in range(REPS):
    for m, label in MODES.items():
        res[label].append(timed(m))
per = sx.ipe
if sx.h.halo_message_bytes() != sx.h.halo_bytes():
    print(f"# agreed water-free dry slab: halo message {sx.h.halo_message_bytes()} of {sx.h.halo_bytes()} bytes (base texture alone); the library runs its periods IN ORDER, "
          "iterations in pairs -- wx_step_overlap's flags are ignored: the two 'overlapped' rows are the in-order protocol on the comm stream")
print(f"{WORK} {XG}x{Y} as one of {NSLAB} slabs ({XO} + 2 x {HALO} columns, {per} iterations per exchange): ms / iteration, {REPS} interleaved repetitions")
for k, v in res.items():
    print(f"  {k:48s} " + " ".join(f"{x:.4f}" for x in v) + f"   median {np.median(v):.4f}")
