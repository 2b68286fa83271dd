#!/usr/bin/env python3
"""Benchmark of the hot path: Mcell-steps/s of the simulation iteration on MI355X.

`python bench.py --gpus N --steps K --warmup W`. A "step" is ONE simulation iteration (all grid passes + lighting; BASELINE.json
configs[2]) over the 16384 x 2048 synthetic terrain grid, issued the way the reference's frame loop issues them: frames of
IterPerFrame = 10 iterations (app.js:398), i.e. the display-only outputs (curl, post-boundary water, post-advection base) are
produced once per 10 iterations (`--frame 1` = every iteration, round 1's definition; both are in the N = 1 line). The fluid MOVES:
a seeded velocity field (sigma = 0.2 cells / iteration, `--flow`) is added on the device before the warm-up, so back-traces leave
the lane's own cell; the figure for the still start state is reported next to it (`at_rest`).
For N > 1 the same grid is cut into N column slabs (strong scaling) with a ring halo exchange over RCCL, overlapped with compute
on a side stream; `--workload dry --X 32768 --Y 4096` runs BASELINE's north-star stencil the same way. Typed without a launcher
(`python bench.py --gpus 8`: no WORLD_SIZE in the environment) the script re-executes itself under `torch.distributed.run`, one rank
per GPU. An N-rank grid run is self-validating by default (`--no-verify` switches it off): every rank checksums its owned columns and
rank 0 compares them with an undecomposed run of the same state (`"verify": "ok"`, `ranks_seen`, `transport` in the line).
Prints one JSON line on rank 0 (contract in the task description): throughput with inputs resident in HBM, plus `roofline`
(dominant kernel, HIP-event timed on the engine's stream), `cpu_baseline` (the CPU oracle timed on this box's host cores, N = 1 only),
`hbm_ceiling` (tools/ubench_hbm: read / write / copy streams with the kernels' own access pattern) and `north_star_dry` (N = 1 only):
BASELINE.json's north-star stencil -- the fused pressure + velocity + advection kernel on a 32768 x 4096 dry grid -- measured in
the same process with its own roofline, at rest and on the moving fluid.
"""
import argparse
import datetime
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured float4-copy ceiling

# Algorithmic (compulsory) bytes per cell and launch for each kernel -- DESIGN.md section 4.
# fp32 RGBA texel = 16 B, wall RGBA8I = 4 B, curl 4 B, vortForce 8 B.
ALGO_BYTES = {
    "velocity": 20 + 20, "curl": 16 + 4, "vorticity": 4 + 8, "boundary": 60 + 36, "advection": 36 + 36,
    "pressure": 20 + 20, "lighting": 52 + 16,
    # the whole iteration as one row-marching kernel: R base16 wall4 water16 light16 (source x, zw + light_0 y), W base16 wall4 water16
    # light16 -- every compulsory byte of the iteration exactly once
    "march_wet_full_iteration": 52 + 52,
    # dry config: A_dry = base 16 R + 16 W + wall 4 R; the tiled kernel also passes the wall texel through (+4 W), the
    # marching kernel only when the brush / an airplane crash could change it
    "fused_dry_vel_advect_pressure": 36,
    "march_dry_vel_advect_pressure": 36,
    # WX_OPT_DRY_PAIRS: one launch = TWO iterations = two cell-steps per cell: 2 x A_dry algorithmic bytes per launch (A_dry is per
    # cell-STEP); what the launch really moves is 36 B/cell (the second iteration's input never leaves the wavefront) -- see roofline.traffic
    "march_dry2_two_iterations_per_launch": 72,
}
# iterations one launch of a kernel advances (the pair kernel: two): ALGO_BYTES is per LAUNCH, SURVEY's figures are per cell-STEP
ITERS_PER_LAUNCH = {"march_dry2_two_iterations_per_launch": 2}
PAIR_NOTE = ("one launch = TWO iterations (WX_OPT_DRY_PAIRS, csrc/wx_march2.h): `achieved` / `frac` count SURVEY 8d's A_dry = 36 B per cell-STEP, i.e. 72 B/cell per "
             "launch, as the contract defines them -- temporal blocking moves only 36 B/cell per launch (the second iteration's input never leaves the wavefront), "
             "so `frac` is no longer bounded by 1 (at rest it exceeds it); `frac_moved` = the bytes a launch MUST move (36 B/cell) / time / peak, `traffic` = what it "
             "did move: the kernel is now bound by the vector ALU (`valu.busy_frac` with 4 cycles per wave64 instruction reads > 1: the real issue rate is ~3.6)")
VERIFY_FIELDS = ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--X", type=int, default=16384)
    ap.add_argument("--Y", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the short rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--workload", choices=["wet", "dry"], default="wet",
                    help="wet: BASELINE configs[2] (default, the metric's config); dry: configs[1] pressure+velocity+advection only")
    ap.add_argument("--particles", type=int, default=0, help="also run the particle pass with N droplets (BASELINE configs[4])")
    ap.add_argument("--splat-order", type=int, default=0, choices=[0, 1], help="with --particles: 1 = deterministic splat order (WX_OPT_SPLAT_ORDER: a sort per iteration)")
    ap.add_argument("--frame", type=int, default=10, help="iterations per wx_step call (the reference's IterPerFrame, app.js:398)")
    ap.add_argument("--flow", type=float, default=0.2, help="std of the seeded velocity field, cells / iteration (0: fluid at rest)")
    ap.add_argument("--flow-kind", choices=["eddies", "noise"], default="eddies",
                    help="eddies: smooth divergence-free eddies of 50..400 cells (a developed flow; default); noise: white noise per cell (worst case for the ring reads)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 32768 x 4096 dry north-star measurement (N=1)")
    ap.add_argument("--no-arith-fast", action="store_true", help="skip the side measurement of the opt-in tolerance build (csrc/libwxsim_fast.so) reported as the EXTRA key `arith_fast` (N=1)")
    ap.add_argument("--no-extras", action="store_true", help="skip at_rest / frame-1 / hbm_ceiling side measurements (N=1)")
    ap.add_argument("--tune", type=int, default=20, help="wx_tune_placement: further device allocations to try for the handle's planes (0: keep the first)")
    ap.add_argument("--dry-pairs", type=int, default=1, choices=[0, 1], help="dry stencil: WX_OPT_DRY_PAIRS (two iterations per launch; round 5 prototype)")
    ap.add_argument("--verify", dest="verify", action="store_true", default=True,
                    help="N > 1, grid only (default ON): checksum every rank's owned columns against an undecomposed run of the same state on rank 0")
    ap.add_argument("--no-verify", dest="verify", action="store_false", help="N > 1: skip the self-validation of the decomposed run")
    a = ap.parse_args()
    a.frame = max(1, a.frame)  # (0 or a negative value would never advance run_frames)
    return a


def reference_cpu_path():
    """the reference itself (its shaders under HeadlessChrome + SwiftShader, the `kaleido` pip package, driven by
    oracle/golden/harness.js) is the preferred CPU baseline IF it can run on this box. It needs three things, none of which is part of
    this repository's snapshot: the kaleido package, the reference checkout (its shader files are loaded at run time and may not
    travel) and the harness next to it. Probe and say what is there; the timing itself is `oracle/golden/gen_golden.py save100raw`
    (2.0-2.2 Mcell-steps/s at 100 x 100 on 8 vCPU in the build container)."""
    import importlib.util
    have_kaleido = importlib.util.find_spec("kaleido") is not None
    have_ref = os.path.isdir("/root/reference/shaders")
    have_harness = os.path.exists(os.path.join(ROOT, "oracle", "golden", "harness.js"))
    if have_kaleido and have_ref and have_harness:
        return "present (kaleido + /root/reference + harness): time it with oracle/golden/gen_golden.py; not run inside the bench"
    missing = [n for n, ok in (("kaleido package", have_kaleido), ("/root/reference checkout", have_ref), ("oracle/golden/harness.js", have_harness)) if not ok]
    return "absent on this box (missing: " + ", ".join(missing) + ")"


def cpu_baseline(pkg, budget_s=12.0):
    """Time the CPU oracle (OpenMP, all host cores) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wx_oracle
    wx_oracle.build()
    X, Y = 2048, 512
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y)
    u["enablePrecipitation"] = 0
    o = wx_oracle.OracleSim(X, Y, 0)
    o.upload(base, water, wall)
    o.set_params(u)
    o.step(2)  # warm-up (page faults, OpenMP team start)
    n, chunk = 0, 4
    t0 = time.perf_counter()
    while True:  # bounded sample: whole chunks until the time budget is used
        o.step(chunk)
        n += chunk
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 2000:
            break
    return {"value": X * Y * n / dt / 1e6, "unit": "Mcell-steps/s", "cores": os.cpu_count(), "kind": "port", "reference_cpu_path": reference_cpu_path(),
            "sample": f"CPU oracle (C/OpenMP restatement, not the reference itself), {X}x{Y} terrain grid, {n} iterations, {dt:.1f} s"}


def hbm_ceiling(X, Y):
    """tools/ubench_hbm (built by __graft_entry__.build()): float4 read / write / copy streams and the wet kernel's stream mix with the
    kernels' own access pattern -- the practical ceiling the roofline fractions are to be read against."""
    exe = os.path.join(ROOT, "tools", "ubench_hbm")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, str(X), str(Y), "64", "15"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120, check=True).stdout.decode()
        return json.loads(out.strip().splitlines()[-1])
    except Exception:
        return None


KERNEL_SYMBOL = {  # profile name -> substring of the kernel symbol rocprofv3 reports
    "march_wet_full_iteration": "k_march_wet",
    "fused_dry_vel_advect_pressure": "k_fused_dry", "march_dry_vel_advect_pressure": "k_march_dry", "march_dry2_two_iterations_per_launch": "k_march_dry2", "advection": "k_advection", "boundary": "k_boundary", "lighting": "k_lighting",
    "velocity": "k_velocity", "pressure": "k_pressure", "curl": "k_curl", "vorticity": "k_vorticity", "precipitation": "k_precipitation",
}


def pmc_traffic(a, kernel, X=None, Y=None, workload=None):
    """HBM bytes per launch of `kernel` from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
    in SEPARATE rocprofv3 --pmc passes (kernel-trace only), units of KiB, FETCH_SIZE doubled (on gfx950 it reports half of a
    wide coalesced read stream). Re-runs this script for a few steps under rocprofv3; returns None on any problem."""
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3")
    sym = KERNEL_SYMBOL.get(kernel)
    if not exe or not sym:
        return None, None, None
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT"):
            d = tempfile.mkdtemp(prefix="wxpmc_", dir="/tmp")
            # one whole frame of warm-up, one measured: the averaged launches have the timed region's mix of plain iterations and the
            # one per frame that also writes the display fields
            nfr = max(1, int(a.frame))
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--steps", str(nfr), "--warmup", str(nfr),
                   "--X", str(X or a.X), "--Y", str(Y or a.Y), "--workload", workload or a.workload, "--no-cpu-baseline", "--no-pmc",
                   "--no-north-star", "--no-extras", "--no-arith-fast", "--frame", str(a.frame), "--flow", str(a.flow), "--flow-kind", a.flow_kind, "--tune", "0", "--dry-pairs", str(a.dry_pairs)]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            c = sqlite3.connect(dbs[0])
            rows = c.execute("select k.kernel_name, d.event_id, d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k "
                             "on d.kernel_id = k.id order by d.start").fetchall()
            ev = [e for n, e, _ in rows if sym in n and (sym != "k_march_dry" or "k_march_dry2" not in n)][nfr:]  # skip the warm-up launches
            per = []
            for e in ev:
                v = c.execute("select sum(value) from rocpd_pmc_event where event_id = ?", (e,)).fetchone()[0]
                if v is not None:
                    per.append(v)
            vals[counter] = sum(per) / len(per)
            shutil.rmtree(d, ignore_errors=True)
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, vals["SQ_INSTS_VALU"], vals["SQ_LDS_BANK_CONFLICT"]
    except Exception:
        return None, None, None


def arith_fast_entry(a):
    """EXTRA key (round-5 verdict, item 3): the same headline workload and the north-star stencil on the opt-in TOLERANCE build
    (csrc/libwxsim_fast.so: FMA contraction, 1-ulp reciprocal / sqrt -- SURVEY.md Appendix A; gated by tests/test_fast_arith.py against the
    reference's own outputs and the rounding envelope, NOT bit-identical to the oracle). A process holds one libwxsim, so this re-runs the
    script with WXSIM_LIB pointing at that build; the line's `value` is and stays the exact build."""
    fast = os.path.join(ROOT, "2d-weather-sandbox_amd", "csrc", "libwxsim_fast.so")
    if not os.path.exists(fast) or os.environ.get("WXSIM_LIB"):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(a.steps), "--warmup", str(a.warmup), "--X", str(a.X), "--Y", str(a.Y), "--workload", a.workload,
           "--frame", str(a.frame), "--flow", str(a.flow), "--flow-kind", a.flow_kind, "--tune", str(min(a.tune, 6)), "--no-cpu-baseline", "--no-pmc", "--no-extras", "--no-arith-fast"]
    if a.no_north_star:
        cmd.append("--no-north-star")
    try:
        out = subprocess.run(cmd, env=dict(os.environ, WXSIM_LIB=fast), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True).stdout.decode()
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        res = {"arith": d.get("arith"), "library": "csrc/libwxsim_fast.so (-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt)", "value": d["value"], "unit": d["unit"],
               "ms_per_step": d["ms_per_step"], "roofline_frac": d["roofline"]["frac"] if d.get("roofline") else None, "placement": d.get("placement"),
               "gate": "tests/test_fast_arith.py: reference-output tests + drift inside 8 x the 1-ulp envelope at configs[1] / [2] size; masks bit-exact; NOT bit-identical to the oracle"}
        ns = d.get("north_star_dry")
        if ns:
            res["north_star_dry"] = {"value": ns["value"], "ms_per_step": ns["ms_per_step"], "roofline_frac": ns["roofline"]["frac"]}
        return res
    except Exception as e:
        return {"error": repr(e)}


def run_frames(step, n, frame):
    """n iterations as frames of `frame` iterations, like the reference's draw() loop."""
    done = 0
    while done < n:
        k = min(frame, n - done)
        step(k)
        done += k


CONDITION_S = 0.25  # see condition_clocks()


def condition_clocks(step, sync, frame, seconds=CONDITION_S, agree=None):
    """Untimed iterations for at least `seconds` of wall time, issued right before a timed region, whatever --warmup says.

    The chip needs ~30-40 ms of continuous load to reach its sustained clocks after ANY idle phase (allocation, the placement search's
    frees, a host-side pause): from idle the first frames of 10 iterations run at 0.92 / 0.83 / 0.77 / 0.745 ms per iteration before the
    kernel settles at 0.737 (profiles/r04_driver_gap.txt). A region of `--warmup 5 --steps 20` (4 + 16 ms) would sit entirely on that
    ramp and report the power management, not the kernel. Returns the number of iterations run. `agree` (N ranks): a callable that
    turns this rank's "keep going" into the job's (an all-reduce), so that every rank runs the same number of iterations."""
    n = 0
    sync()
    t0 = time.perf_counter()
    more = True
    while more:
        step(frame)
        n += frame
        sync()
        more = time.perf_counter() - t0 < seconds
        if agree is not None:
            more = agree(more)
    return n


def timed_run(h, steps, frame, profile=True):
    """(seconds, profile dict) of `steps` iterations on a single handle, synchronised on both sides."""
    import torch
    if profile:
        h.profile(True)
    h.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_frames(h.step, steps, frame)
    h.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = h.profile_read() if profile else {}
    if profile:
        h.profile(False)
    return dt, prof


def north_star_dry(a, pkg, X=32768, Y=4096, steps=200, warmup=20):
    """BASELINE.json north_star: the fused pressure + velocity + advection stencil on a 32768 x 4096 dry grid (configs[1]'s
    passes at configs[3]'s size) on one GPU, same process, own roofline. A_dry = 36 B/cell-step. Measured on the still
    start state of SURVEY's C2 (`at_rest`) and -- the claim -- on a moving fluid (seeded sigma = --flow velocities)."""
    from weather_sandbox_amd import devtools
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    base, water, wall = pkg.synth.dry_grid(X, Y)
    h.upload(base, water, wall)
    del base, water, wall
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.set_option(h.OPT_DRY_PAIRS, a.dry_pairs)
    n = (steps // a.frame) * a.frame
    placement = None
    if a.tune > 0:
        ms0, ms1 = h.tune_placement(a.tune, 30)
        placement = {"tries": a.tune, "ms_per_iteration_first_allocation": ms0, "ms_per_iteration_kept": ms1}

    def measure(label):
        run_frames(h.step, warmup, a.frame)
        condition_clocks(h.step, h.sync, a.frame)
        runs = [timed_run(h, n, a.frame) for _ in range(3)]  # three timed runs: the slowest one is the claim
        dt, prof = max(runs, key=lambda r: r[0])
        name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_ms = ms / cnt
        achieved = ALGO_BYTES[name] * X * Y / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "algo_bytes_per_cell": ALGO_BYTES[name], "avg_launch_ms": avg_ms, "launches": cnt, "traffic": None}
        if ITERS_PER_LAUNCH.get(name, 1) > 1:
            roof["iterations_per_launch"] = ITERS_PER_LAUNCH[name]
            roof["frac_moved"] = ALGO_BYTES[name] / ITERS_PER_LAUNCH[name] * X * Y / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["note"] = PAIR_NOTE
        return {"state": label, "value": X * Y * n / dt / 1e6, "ms_per_step": dt / n * 1e3, "runs_Mcell_steps_per_s": [X * Y * n / r[0] / 1e6 for r in runs],
                "flow": devtools.flow_stats(h), "roofline": roof}

    rest = measure("at rest (SURVEY 8d C2: v = 0, P = 0)")
    moving = noise = None
    if a.flow > 0:
        devtools.seed_flow(h, a.flow, kind=a.flow_kind)
        moving = measure(f"moving fluid: seeded {a.flow_kind}, std {a.flow} cells/iteration")
        if a.flow_kind != "noise" and not a.no_extras:  # the worst case for the ring reads: every lane's footprint differs from its neighbour's
            devtools.seed_flow(h, a.flow, kind="noise", seed=2)
            noise = measure(f"white-noise velocities on top, std {a.flow} cells/iteration (worst case)")
    fast = None
    if a.flow > 0 and a.dry_pairs and not a.no_extras:
        fast = fast_cells_entry(a, h, X, Y, n, moving)
    h.close()
    main_ = moving or rest
    res = {"workload": f"{X}x{Y} dry-air grid, pressure+velocity+advection only (BASELINE north_star / configs[1] passes), frames of {a.frame}",
           "value": main_["value"], "unit": "Mcell-steps/s", "steps": n, "ms_per_step": main_["ms_per_step"],
           "runs_Mcell_steps_per_s": main_["runs_Mcell_steps_per_s"], "claim": "slowest of three runs, " + main_["state"], "flow": main_["flow"],
           "roofline": main_["roofline"], "placement": placement}
    if moving:
        res["at_rest"] = {k: rest[k] for k in ("value", "ms_per_step", "runs_Mcell_steps_per_s", "flow")}
        res["at_rest"]["roofline_frac"] = rest["roofline"]["frac"]
    if fast:
        res["with_fast_cells"] = fast
    if noise:
        res["white_noise_worst_case"] = {k: noise[k] for k in ("value", "ms_per_step", "runs_Mcell_steps_per_s", "flow")}
        res["white_noise_worst_case"]["roofline_frac"] = noise["roofline"]["frac"]
    if not a.no_pmc:
        traffic, valu, conflicts = pmc_traffic(a, res["roofline"]["kernel"], X, Y, "dry")
        res["roofline"]["traffic"] = traffic
        res["roofline"]["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, KiB counters)"
        if valu:
            res["roofline"]["valu"] = {"wave_insts_per_launch": valu, "busy_frac": valu * 4.0 / (1024 * 2.4e9 * res["roofline"]["avg_launch_ms"] * 1e-3)}
        if conflicts is not None:
            res["roofline"]["lds_bank_conflict_cycles_per_launch"] = conflicts
    return res


def fast_cells_entry(a, h, X, Y, n, clean):
    """Round-5 verdict, item 1: the reference has no velocity clamp (advectionShader.frag:85-99), and the pair kernel's SECOND iteration has
    no exact path inside the march for back-traces of 0.9 cells and more. The same flow plus a handful of seeded vortices of 1.3 cells /
    iteration (fresh ones before every timed run: they decay), timed like the claim; `cells_recomputed` / `pairs_repeated_whole` are the
    device's own counters (wx_pair_stats). Round 5 repeated the whole grid twice per such pair (2.4 x per iteration)."""
    from weather_sandbox_amd import devtools
    runs = []
    for k in range(3):
        devtools.seed_vortices(h, 8, peak=1.3, radius=20.0, seed=11 + k)
        before = devtools.flow_stats(h)
        condition_clocks(h.step, h.sync, a.frame, seconds=0.05)
        h.pair_stats()
        dt, _ = timed_run(h, n, a.frame, profile=False)
        fixed, repeated = h.pair_stats()
        after = devtools.flow_stats(h)
        runs.append({"ms_per_step": dt / n * 1e3, "cells_recomputed": fixed, "pairs_repeated_whole": repeated,
                     "cells_ge_0.9_before": before["cells_component_ge_0.9"], "cells_ge_0.9_after": after["cells_component_ge_0.9"], "max_v_before": before["max_v"]})
    worst = max(runs, key=lambda r: r["ms_per_step"])
    return {"flow": "the moving fluid + 8 fresh vortices of 1.3 cells / iteration (radius 20) before each of three timed runs", "steps_per_run": n,
            "value": X * Y / worst["ms_per_step"] / 1e3, "unit": "Mcell-steps/s", "ms_per_step": worst["ms_per_step"],
            "vs_clean_flow": worst["ms_per_step"] / clean["ms_per_step"], "runs": runs,
            "exact_path": "k_dry2_fix: one wavefront per recorded 8 x 8 tile (fast or NaN-tainted second-iteration cells; both iterations rebuilt for the tile from the pair's inputs, csrc/wx_march2.h); bit-identical to one iteration per launch"}


def measure_handle(h, cells, steps, warmup, frame, algo_bytes=None, events_in_timed_region=True):
    """warm-up, clock conditioning, `steps` timed iterations on one handle with the per-kernel profile: the entry of one configuration.

    events_in_timed_region=False (configurations whose iteration is tens of microseconds): the timed region runs WITHOUT the per-kernel
    HIP events -- two event records per launch group are a ~3 us bubble each, 10 % of a 31 us iteration (profiles/r04_bench_timeline_event_gaps.txt,
    profiles/r05_c1_segment_sweep.txt: 30.8 us per iteration without them, 34.5 with) -- and a second, profiled run of the same length follows for
    the kernel breakdown; `roofline_frac` is then taken from the un-profiled whole step (launch gaps included: conservative), the event-timed
    figure is reported next to it."""
    run_frames(h.step, warmup, frame)
    condition_clocks(h.step, h.sync, frame)
    if events_in_timed_region:
        dt, prof = timed_run(h, steps, frame)
    else:
        dt, _ = timed_run(h, steps, frame, profile=False)
        _, prof = timed_run(h, steps, frame)
    out = {"value": cells * steps / dt / 1e6, "unit": "Mcell-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
           "kernels_ms_per_step": {k: v[0] / steps for k, v in prof.items()}}
    if prof:
        name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
        ab = (algo_bytes or {}).get(name, ALGO_BYTES.get(name))
        out["dominant_kernel"] = name
        out["avg_launch_ms"] = ms / cnt
        if ab:
            out["algo_bytes_per_cell"] = ab
            ev = ab * cells / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS
            if ITERS_PER_LAUNCH.get(name, 1) > 1:
                out["iterations_per_launch"] = ITERS_PER_LAUNCH[name]
            if events_in_timed_region:
                out["roofline_frac"] = ev
            else:
                out["roofline_frac"] = ab / ITERS_PER_LAUNCH.get(name, 1) * cells / (dt / steps) / 1e9 / HBM_PEAK_GBS  # (per cell-STEP against the time of a step)
                out["roofline_frac_event_timed"] = ev
                out["timed_region"] = "no per-kernel events (a second, profiled run gives kernels_ms_per_step / avg_launch_ms); roofline_frac from the whole un-profiled step"
    return out


def other_configs(a, pkg):
    """The other single-GPU configurations of BASELINE.json, measured in the same process so that the driver's record carries them:
    configs[0] (the reference's own 100 x 100 save, 1000 iterations; inputs from tests/golden/save100raw.npz), configs[1] (4096 x 1024
    dry, SURVEY 8d C2: 200 warm-up + 1000 timed) and configs[4] on one GPU (16384 x 2048 + 1 048 576 droplets over a cloud deck,
    200 iterations of spin-up, 200 timed; SURVEY's C5 protocol -- 2000 + 1000 -- is `bench.py --particles 1048576 --warmup 2000 --steps 1000`)."""
    import numpy as np
    from weather_sandbox_amd import devtools
    out = {}
    # configs[0]
    g = np.load(os.path.join(ROOT, "tests", "golden", "save100raw.npz"))
    u = json.loads(str(g["uniforms_json"]))
    u["initial_T"] = g["initial_T"]
    for k in ("userInputValues", "userInputMove", "airplaneValues"):
        u[k] = tuple(u[k])
    u = dict(u, quad_scale=0, enablePrecipitation=0)
    X, Y = int(g["X"]), int(g["Y"])
    h = pkg.engine.Handle(X, Y, 0)
    h.upload(g["in_base"], g["in_water"], g["in_wall"])
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    out["c0_save100x100_1000_iterations"] = dict(measure_handle(h, X * Y, 1000, 100, a.frame, events_in_timed_region=False), note="launch-bound (two launches per iteration); the reference "
                                                 "under SwiftShader on 8 vCPU: 2.0-2.2 Mcell-steps/s (BASELINE.md)")
    h.close()
    # the reference's OWN grid sizes: a new simulation is 2500 x 300 by default and the resolution sliders end at 16000 x 500
    # (/root/reference/index.html:325, 335); its frame is IterPerFrame = 10 iterations (app.js:398). Full wet iteration, setup-pass terrain.
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    for X, Y, name in ((2500, 300, "ref_default_2500x300"), (16000, 500, "ref_largest_16000x500")):
        ur = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
        ur["enablePrecipitation"] = 0
        h = pkg.engine.Handle(X, Y, 0)
        h.setup_columns(pkg.synth.terrain_columns(X, Y))
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), ur), ur["initial_T"])
        devtools.seed_flow(h, a.flow if a.flow > 0 else 0.2, kind=a.flow_kind)
        out[name] = dict(measure_handle(h, X * Y, 1000, 100, a.frame, events_in_timed_region=False), note="the reference's own grid size, all six grid passes + lighting, particles off, moving fluid")
        h.close()
    # configs[1]
    X, Y = 4096, 1024
    gui = pkg.params.merge_settings(None)
    ud = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    ud["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    h.upload(*pkg.synth.dry_grid(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), ud), ud["initial_T"])
    out["c1_dry_4096x1024"] = measure_handle(h, X * Y, 1000, 200, a.frame, events_in_timed_region=False)
    h.close()
    # configs[4] on one GPU
    X, Y, N = 16384, 2048, 1048576
    gui["sunAngle"] = 50.0
    uw = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    uw["enablePrecipitation"] = 1
    h = pkg.engine.Handle(X, Y, N)
    h.setup_columns(pkg.synth.terrain_columns(X, Y, cloud_deck=True), pkg.synth.init_rain_drops(N))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), uw), uw["initial_T"])
    # with particles the marching kernel also reads the feedback texture: 104 + 12 B/cell since the end of round 4 (three written channels;
    # the deposition texture and the post-advection temperature only near terrain). SURVEY's A_full = 128 (RGBA feedback 16 + deposition 8)
    # would flatter the fraction now that those bytes are not moved
    place = None
    if a.tune > 0:  # (the placement lottery applies to this handle like to the main one: 0.86 against 0.99 ms for the marching kernel on two boxes)
        run_frames(h.step, 20, a.frame)
        ms0, ms1 = h.tune_placement(a.tune, 30)
        place = {"tries": a.tune, "ms_per_iteration_first_allocation": ms0, "ms_per_iteration_kept": ms1}
    out["c4_particles_1M_16384x2048_one_gpu"] = dict(measure_handle(h, X * Y, 200, 200, a.frame, {"march_wet_full_iteration": 116}), placement=place)
    h.close()
    return out


def p2p_selftest(dist, torch, rank, world, device):
    """First contact with the transport: every rank sends its rank number to both ring neighbours and checks what arrives, with a
    bounded wait -- so that a broken RCCL path fails here, readably, instead of hanging in the timed region."""
    left, right = (rank - 1) % world, (rank + 1) % world
    send = torch.full((256,), float(rank), device=device)
    recv = [torch.full((256,), -1.0, device=device) for _ in range(2)]
    if world == 2:
        ops = [dist.P2POp(dist.isend, send, left), dist.P2POp(dist.isend, send.clone(), right),
               dist.P2POp(dist.irecv, recv[1], right), dist.P2POp(dist.irecv, recv[0], left)]
    else:
        ops = [dist.P2POp(dist.isend, send, left), dist.P2POp(dist.irecv, recv[1], right),
               dist.P2POp(dist.isend, send.clone(), right), dist.P2POp(dist.irecv, recv[0], left)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    if device.type == "cuda":
        torch.cuda.synchronize()
    got = (float(recv[0][0]), float(recv[1][0]))
    if got != (float(left), float(right)):
        raise RuntimeError(f"rank {rank}: P2P self-test received {got} from ranks ({left}, {right})")


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): re-execute this script under
    `torch.distributed.run`, one rank per GPU, rendezvous on 127.0.0.1 (the container's hostname may not resolve) -- the command the
    driver types for N > 1, typed for it. Returns only if there is nothing to launch."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus and not os.environ.get("WX_BENCH_SHARE_GPU"):
        print(json.dumps({"error": f"--gpus {a.gpus} asked, {have} HIP device(s) visible on this box", "n_gpus": a.gpus, "devices_visible": have}), flush=True)
        sys.exit(2)
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:  # a free port
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    a = parse()
    self_launch(a)
    import numpy as np
    import torch
    import wxpkg
    pkg = wxpkg.load_package()
    from weather_sandbox_amd import devtools

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if os.environ.get("WX_BENCH_SHARE_GPU"):  # plumbing test on a 1-GPU box: all ranks on cuda:0, gloo transport
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("WX_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=int(os.environ.get("WX_DIST_TIMEOUT_S", "180")))  # a stuck transfer ends the run with an error, not a hang
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        ranks_seen = dist.get_world_size()
        try:
            p2p_selftest(dist, torch, rank, world, device if backend == "nccl" else torch.device("cpu"))
        except Exception as e:  # readable first-contact failure
            print(json.dumps({"error": f"P2P self-test failed on rank {rank} ({backend}): {e}", "ranks_seen": ranks_seen}), flush=True)
            raise
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert not (a.workload == "dry" and a.particles), "the dry workload has no particle pass"

    X, Y = a.X, a.Y
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0  # fixed sun ('MANUAL_ANGLE'), day side
    if a.workload == "dry":
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0, pass_mask=pkg.params.PASS_DRY)
    else:
        u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1 if a.particles else 0  # (+ a cloud deck so that droplets spawn, grow and fall during the run)

    def make_whole():
        h = pkg.engine.Handle(X, Y, a.particles)
        if a.workload == "dry":
            h.upload(*pkg.synth.dry_grid(X, Y))
            h.set_option(h.OPT_DRY_PAIRS, a.dry_pairs)
        else:  # device-side initialiser: 1-D terrain / sounding descriptors instead of 1.2 GB of host arrays
            h.setup_columns(pkg.synth.terrain_columns(X, Y, cloud_deck=bool(a.particles)),
                            pkg.synth.init_rain_drops(a.particles) if a.particles else None)
        h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
        if a.particles and a.splat_order:
            h.set_option(h.OPT_SPLAT_ORDER, 1)
        return h

    x0_owned = 0
    if world == 1:
        h = make_whole()
        stepper, drv = h, None
        step, sync, barrier = h.step, (lambda: (h.sync(), torch.cuda.synchronize())), (lambda: None)
    else:
        from weather_sandbox_amd import slab
        if a.workload == "dry":
            drv = slab.SlabSim.from_dry_generator(pkg, X, Y, u, rank, world, device)
        elif a.particles:  # partitioned droplet pool (every rank is handed the whole initial pool once); halo 64: sprite clipping needs tile-aligned slab edges
            drv = slab.SlabSim.from_generator(pkg, X, Y, u, rank, world, device, halo=64,
                                              drops=pkg.synth.init_rain_drops(a.particles), cloud_deck=True)
        else:
            drv = slab.SlabSim.from_generator(pkg, X, Y, u, rank, world, device)
        stepper = drv.handle
        x0_owned = slab.slab_columns(X, rank, world)[0]
        step, sync, barrier = drv.step, (lambda: (drv.sync(), torch.cuda.synchronize())), dist.barrier

    # Where the planes lie in physical memory is worth several % (profiles/r03_alloc_probe.txt): the engine tries a few
    # allocations with its own iteration and keeps the fastest -- set-up work, like the allocation itself; the state is unchanged. Done
    # FIRST, so that every figure of this line (at_rest, the headline, frame_1) is measured on the SAME, kept placement.
    placement = None
    if a.tune > 0 and not (world > 1 and a.particles):
        ms0, ms1 = stepper.tune_placement(a.tune, 30)
        placement = {"tries": a.tune, "ms_per_iteration_first_allocation": ms0, "ms_per_iteration_kept": ms1,
                     "applies_to": "value, at_rest and frame_1 (all measured after the search, on the kept placement)"}
    elif world == 1:  # --tune 0: a whole-domain handle of 8 Mi cells and more searches by itself inside its first wx_step (WX_OPT_PLACEMENT_SEARCH, 6 tries)
        step(a.frame)
        pi = stepper.placement_info()
        if pi is not None:
            placement = {"tries": "implicit: 6, inside the handle's first wx_step (WX_OPT_PLACEMENT_SEARCH; what a host that never calls wx_tune_placement gets)",
                         "ms_per_iteration_first_allocation": pi[0], "ms_per_iteration_kept": pi[1]}
    at_rest = None
    if world == 1 and a.flow > 0 and not a.no_extras and not a.particles:  # the still start state first (round 2's definition)
        run_frames(step, a.warmup, a.frame)
        condition_clocks(step, sync, a.frame)
        dt0, _ = timed_run(stepper, a.steps, a.frame, profile=False)
        at_rest = {"value": X * Y * a.steps / dt0 / 1e6, "ms_per_step": dt0 / a.steps * 1e3, "flow": devtools.flow_stats(stepper),
                   "state": "still start state (what round 2 timed)", "placement": "the kept one (same as value)"}
    if a.flow > 0:  # a moving fluid: identical on slabs and on the undecomposed grid (function of the global cell index)
        devtools.seed_flow(stepper, a.flow, Xg=X, x0=x0_owned, kind=a.flow_kind)
    flow0 = devtools.flow_stats(stepper)

    run_frames(step, a.warmup, a.frame)
    # the --warmup iterations above are the contract's; the clocks are conditioned by TIME on top of them (condition_clocks), so that a
    # short region (the driver's --warmup 5 --steps 20 is 20 ms in all) measures the kernel at its sustained clocks
    agree = None
    if world > 1:
        def agree(more):
            t = torch.tensor([1 if more else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(t.item())
    conditioning_iters = condition_clocks(step, sync, a.frame, agree=agree)
    sync()
    barrier()
    # per-kernel HIP events (4 event records per iteration) are cheap next to a 1 ms iteration on one GPU; on N GPUs an
    # iteration is N times shorter, so there the timed region runs without them and a short profiled segment follows
    profile_inline = world == 1
    if profile_inline:
        stepper.profile(True)
    sync()
    t0 = time.perf_counter()
    run_frames(step, a.steps, a.frame)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    extra_iters = 0
    if not profile_inline:
        stepper.profile(True)
        step(16)
        extra_iters = 16
        sync()
    prof = stepper.profile_read()
    stepper.profile(False)
    prof_steps = a.steps if profile_inline else 16
    flow1 = devtools.flow_stats(stepper)
    verify = "n/a (single rank)" if world == 1 else "off (--no-verify)"
    whole, n1, per_rank = None, None, None
    # largest |velocity component| the exact path of the wet kernel saw during the run (0: nothing reached 0.9 cells / iteration). On
    # slabs the halo width assumes |v| < 1: a larger value means the decomposed run may differ from the undecomposed one near slab edges
    fastest = stepper.fastest_velocity() if a.workload == "wet" else 0.0
    if world > 1:
        tf = torch.tensor([fastest], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        fastest = float(tf.item())
        t = torch.tensor([dt], device="cuda", dtype=torch.float64) if dist.get_backend() == "nccl" else torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if a.verify and not a.particles:
            # every rank: checksums of its owned columns; rank 0: the same state undecomposed, same number of iterations
            cs = torch.from_numpy(devtools.checksum(stepper, VERIFY_FIELDS, Xg=X, x0=x0_owned))
            if dist.get_backend() == "nccl":
                cs = cs.to(device)
            gathered = [torch.zeros_like(cs) for _ in range(world)]
            dist.all_gather(gathered, cs)
            if rank == 0:
                whole = make_whole()
                if a.flow > 0:
                    devtools.seed_flow(whole, a.flow, Xg=X, x0=0, kind=a.flow_kind)
                whole.step(a.warmup + conditioning_iters + a.steps + extra_iters)
                bad = []
                xo = X // world
                for r in range(world):
                    ref = devtools.checksum(whole, VERIFY_FIELDS, Xg=X, x0=0, cols=slice(r * xo, (r + 1) * xo))
                    got = gathered[r].cpu().numpy()
                    bad += [f"rank {r} {f}" for k, f in enumerate(VERIFY_FIELDS) if not np.array_equal(ref[k], got[k])]
                verify = "ok" if not bad else "MISMATCH: " + ", ".join(bad)
        elif a.verify:
            verify = "not available with particles (a period may differ from the undecomposed run by a few re-spawn probes: tests/test_gpu_fullsize.py quantifies it)"
        # Round-5 verdict, item 7: what the curve decomposes into, measured in THIS run on rank 0's GPU while the other ranks wait in the final
        # barrier -- (a) the same grid UNDECOMPOSED on the same box (`strong_scaling_vs_n1`: SCALE's N = 1 point without a second run; the handle
        # searches its placement inside its first step), (b) this rank's slab with NO exchange at all (`per_rank.plain_slab_ms_per_step`: the
        # kernel on a narrow launch; the rest of ms_per_step is the exchange protocol + the link).
        if rank == 0 and not a.particles:
            try:
                if whole is None:
                    whole = make_whole()
                    if a.flow > 0:
                        devtools.seed_flow(whole, a.flow, Xg=X, x0=0, kind=a.flow_kind)
                    run_frames(whole.step, a.warmup, a.frame)
                condition_clocks(whole.step, whole.sync, a.frame)
                dtw, _ = timed_run(whole, a.steps, a.frame, profile=False)
                n1 = {"n1_ms_per_step": dtw / a.steps * 1e3, "n1_value": X * Y * a.steps / dtw / 1e6, "speedup": dtw / dt,
                      "how": "the undecomposed grid timed on rank 0's GPU in the same run (same state family, same frames), the other ranks idle",
                      "n1_placement": whole.placement_info()}
                whole.close()
                whole = None
                condition_clocks(stepper.step, stepper.sync, a.frame)
                dtp, _ = timed_run(stepper, a.steps, a.frame, profile=False)
                per_rank = {"local_columns": stepper.X, "owned_columns": X // world, "plain_slab_ms_per_step": dtp / a.steps * 1e3,
                            "protocol_and_link_share_of_ms_per_step": 1.0 - (dtp / a.steps) / (dt / a.steps),
                            "plain_slab_x_n_over_n1": (dtp * world) / dtw,
                            "how": "rank 0's slab stepped with no exchange at all after the timed region (ghost columns stale: timing only)"}
            except Exception as e:  # (never at the price of the line itself)
                n1 = n1 or {"error": repr(e)}
        if whole is not None:
            whole.close()

    if rank == 0:
        cells = X * Y
        value = cells * a.steps / dt / 1e6
        local_cells = stepper.X * Y
        dom = max(prof.items(), key=lambda kv: kv[1][0]) if prof else None
        roof = None
        if dom:
            name, (ms, cnt) = dom
            avg_ms = ms / cnt
            achieved = ALGO_BYTES.get(name, 0) * local_cells / (avg_ms * 1e-3) / 1e9
            traffic, valu, conflicts = (None, None, None) if (a.no_pmc or world > 1) else pmc_traffic(a, name)
            roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, KiB counters)",
                    "algorithmic_bytes_per_launch": ALGO_BYTES.get(name, 0) * local_cells, "avg_launch_ms": avg_ms, "launches": cnt,
                    "algo_bytes_per_cell": ALGO_BYTES.get(name, 0),
                    "kernels_ms_per_step": {k: v[0] / prof_steps for k, v in prof.items()}}
            if ITERS_PER_LAUNCH.get(name, 1) > 1:
                roof["iterations_per_launch"] = ITERS_PER_LAUNCH[name]
                roof["frac_moved"] = roof["frac"] / ITERS_PER_LAUNCH[name]
                roof["note"] = PAIR_NOTE
            if world == 1 and not a.no_extras:
                # the runtime's own device-to-device copy of 1 GiB, read + write bytes counted (kept for comparison with round 2; the
                # ceiling that matters is hbm_ceiling below: the kernels' own access pattern)
                src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
                dst = torch.empty_like(src)
                dst.copy_(src)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    dst.copy_(src)
                e1.record()
                torch.cuda.synchronize()
                roof["measured_copy_GBps"] = 2.0 * (1 << 30) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                del src, dst
            if valu:
                # the second limiter: SQ_ACTIVE_INST_VALU charges a wave64 VALU instruction 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
                roof["valu"] = {"wave_insts_per_launch": valu, "busy_frac": valu * 4.0 / (1024 * 2.4e9 * avg_ms * 1e-3),
                                "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time); an all-v_mul_f32 micro-benchmark "
                                        "issues one per 3.05 cycles (profiles/r02_ubench_valu.txt)"}
            if conflicts is not None:
                roof["lds_bank_conflict_cycles_per_launch"] = conflicts
        if a.workload == "dry":
            wl = f"{X}x{Y} dry-air grid, pressure+velocity+advection only (BASELINE configs[1] passes" + (", north-star size)" if (X, Y) == (32768, 4096) else ")")
        else:
            wl = f"{X}x{Y} terrain grid, all six grid passes + lighting, "
            if a.particles:
                wl += f"{a.particles} precipitation particles (BASELINE configs[4]" + (" on one GPU)" if world == 1 else ")")
            else:
                wl += "particles off (BASELINE configs[3] grid)" if (X, Y) == (32768, 4096) else "particles off (BASELINE configs[2])"
        wl += f"; frames of {a.frame} iterations; " + (f"moving fluid (seeded {a.flow_kind}, std {a.flow} cells/iteration)" if a.flow > 0 else "fluid at rest")
        if world == 1:
            deco = "none"
        elif a.particles:
            deco = f"{world} x-slabs, 64 ghost columns, partitioned droplet pool; every {drv.iters_per_exchange} iterations: ring halo exchange + edge droplets (one batch of send/recv), status-flip events (all-gather of a few KB)"
        else:
            deco = (f"{world} x-slabs, {drv.halo} ghost columns, ring halo exchange every {drv.iters_per_exchange} iterations "
                    "(RCCL send/recv on a side stream, overlapped with compute)")
        A = 36 if a.workload == "dry" else 72
        out = {
            "metric": "Mcell-steps/s", "value": value, "unit": "Mcell-steps/s", "n_gpus": world, "steps": a.steps,
            "arith": "fast (tolerance build, NOT the parity build)" if pkg.engine.lib().wx_arith() == 1 else "exact",
            "warmup": a.warmup, "clock_conditioning": {"untimed_iterations_after_warmup": conditioning_iters, "min_seconds": CONDITION_S,
                                                       "why": "sustained clocks need ~40 ms of load after any idle phase (profiles/r04_driver_gap.txt)"},
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "grid": [X, Y], "decomposition": deco, "iteration_algorithmic_bytes_per_cell": A,
                       "frame": a.frame, "flow_std": a.flow, "flow_kind": a.flow_kind},
            "ranks_seen": ranks_seen, "verify": verify, "placement": placement, "transport": getattr(drv, "transport", None) if drv is not None else None,
            "flow": {"start_of_warmup": flow0, "end_of_timed_region": flow1, "fastest_velocity_on_the_exact_path": fastest},
            "iteration_roofline_frac_A_wet": cells * a.steps * A / dt / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": roof,
        }
        if n1:
            out["strong_scaling_vs_n1"] = n1
        if per_rank:
            out["per_rank"] = per_rank
        if at_rest:
            out["at_rest"] = at_rest
        if world == 1 and not a.no_extras and a.frame != 1 and not a.particles:
            condition_clocks(lambda k: run_frames(stepper.step, k, 1), sync, a.frame)
            dt1, _ = timed_run(stepper, a.steps, 1, profile=False)  # round 1's definition: the display fields after EVERY iteration
            out["frame_1"] = {"value": cells * a.steps / dt1 / 1e6, "ms_per_step": dt1 / a.steps * 1e3, "placement": "the kept one (same as value)"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg)
        if world == 1 and not a.no_extras:
            stepper.close()
            ceil = hbm_ceiling(16384, 2048)
            if ceil:
                out["hbm_ceiling"] = ceil
                if roof and roof.get("traffic"):
                    roof["frac_of_copy_ceiling_real_traffic"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / ceil["copy_GBps"]
        if world == 1 and not a.no_north_star and a.workload == "wet" and not a.particles:
            stepper.close()
            out["north_star_dry"] = north_star_dry(a, pkg)
        if world == 1 and not a.no_extras and not a.no_north_star and a.workload == "wet" and not a.particles and (X, Y) == (16384, 2048):
            out["configs"] = other_configs(a, pkg)
        if world == 1 and not a.no_arith_fast and not a.particles:
            fa = arith_fast_entry(a)
            if fa:
                if "value" in fa:
                    fa["vs_exact"] = fa["value"] / out["value"]
                    if "north_star_dry" in fa and "north_star_dry" in out:
                        fa["north_star_dry"]["vs_exact"] = fa["north_star_dry"]["value"] / out["north_star_dry"]["value"]
                out["arith_fast"] = fa
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
