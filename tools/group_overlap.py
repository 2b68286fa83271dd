#!/usr/bin/env python3
"""What WX_OPT_EXCHANGE_OVERLAP buys on ONE GPU: BASELINE configs[4] (16384 x 2048, 1 M droplets) and configs[2] (particles off) cut
into N slabs that all live on device 0 and exchange through the library's "local" transport (wx_group_step: device-to-device copies
fenced by events) -- once with the exchange on the slabs' side streams, once in order on their compute streams, interleaved.
A one-GPU group is not a scaling measurement (the N slabs share the chip, so even the in-order variant overlaps one slab's exchange
with another slab's compute); what it shows is the cost of the exchange protocol per period and that the side-stream ordering does
not serialise anything. Usage: python tools/group_overlap.py [n_slabs] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
E = pkg.engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
X, Y = 16384, 2048


def make(ndrops, halo, overlap):
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1 if ndrops else 0
    g = E.Group(N, X, Y, halo=halo, devices=[0] * N, transport=E.TRANSPORT_LOCAL, n_droplets=ndrops)
    drops = pkg.synth.init_rain_drops(ndrops) if ndrops else None
    for i, h in enumerate(g.slabs):
        c = g.columns(i)
        h.setup_columns(pkg.synth.terrain_columns(X, Y, cols=(int(c[0]), len(c)), cloud_deck=bool(ndrops)), drops)
    g.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    g.set_option(E.Handle.OPT_EXCHANGE_OVERLAP, overlap)
    return g


def timed(g, iters, frame=10):
    g.sync()
    t0 = time.perf_counter()
    for _ in range(iters // frame):
        g.step(frame)
    g.sync()
    return (time.perf_counter() - t0) / iters * 1e3


for label, ndrops, halo in (("configs[4]: 1 M droplets", 1 << 20, 64), ("configs[2]: particles off", 0, 48)):
    gs = {ov: make(ndrops, halo, ov) for ov in (1, 0)}
    for g in gs.values():
        timed(g, 200)
    res = {1: [], 0: []}
    for _ in range(REPS):
        for ov, g in gs.items():
            res[ov].append(timed(g, 200))
    per = 1 + (halo - 12) // 9 if ndrops else halo // 6
    print(f"{label} as {N} slabs on one GPU, halo {halo} ({per} iterations / exchange): ms / iteration (all slabs), {REPS} interleaved repetitions of 200 iterations")
    for ov in (1, 0):
        print(f"  exchange {'on side streams' if ov else 'in order      '}: " + " ".join(f"{v:.4f}" for v in res[ov]) + f"   median {np.median(res[ov]):.4f}")
    for g in gs.values():
        g.close()
