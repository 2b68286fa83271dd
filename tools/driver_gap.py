#!/usr/bin/env python3
"""Why does a 20-iteration timed region right after wx_tune_placement run slower than the tune's own probe?

Times single frames of 10 iterations (host clock around a synchronised frame, and the engine's own HIP events) from an idle chip,
right after the placement search, and after a deliberate idle gap -- the curve shows how long the clocks need and whether the
placement the search kept is the sustained one.  Output: one JSON line per phase.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frames(h, n, frame=10):
    out = []
    for _ in range(n):
        h.sync()
        t0 = time.perf_counter()
        h.step(frame)
        h.sync()
        out.append(round((time.perf_counter() - t0) / frame * 1e3, 4))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--X", type=int, default=16384)
    ap.add_argument("--Y", type=int, default=2048)
    ap.add_argument("--tune", type=int, default=8)
    ap.add_argument("--idle", type=float, default=2.0)
    a = ap.parse_args()
    import torch  # noqa: F401
    import wxpkg
    pkg = wxpkg.load_package()
    from weather_sandbox_amd import devtools
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, a.Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(a.X, a.Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(a.X, a.Y, cloud_deck=False), None)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    devtools.seed_flow(h, 0.2, Xg=a.X, x0=0, kind="eddies")
    h.sync()
    time.sleep(a.idle)
    print(json.dumps({"phase": "from idle, first allocation", "ms_per_iteration_by_frame": frames(h, 40)}), flush=True)
    if a.tune > 0:
        t0 = time.perf_counter()
        ms0, ms1 = h.tune_placement(a.tune, 30)
        print(json.dumps({"phase": "tune", "first": ms0, "kept": ms1, "seconds": round(time.perf_counter() - t0, 2)}), flush=True)
        print(json.dumps({"phase": "right after tune", "ms_per_iteration_by_frame": frames(h, 40)}), flush=True)
    time.sleep(a.idle)
    print(json.dumps({"phase": f"after {a.idle} s idle", "ms_per_iteration_by_frame": frames(h, 40)}), flush=True)
    # one long frame sequence without host syncs in between (what bench.py --steps 200 does)
    h.sync()
    t0 = time.perf_counter()
    for _ in range(40):
        h.step(10)
    h.sync()
    print(json.dumps({"phase": "400 iterations back to back", "ms_per_iteration": round((time.perf_counter() - t0) / 400 * 1e3, 4)}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
