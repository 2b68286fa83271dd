"""GPU box: one bench line per BASELINE.json configuration that fits one GPU, written to gpurun_out/configs.md."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rows = []

def bench(label, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pmc", "--no-north-star"] + args, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().split("\n")[-1])
    except ValueError:
        sys.exit(f"bench.py {' '.join(args)} printed no JSON line (rc {r.returncode}):\n{r.stderr[-3000:]}")
    rows.append((label, " ".join(args), d["ms_per_step"], d["value"], d["roofline"]["kernels_ms_per_step"]))

# C1: the reference's own, unmodified 100 x 100 test save (inputs of the save100raw golden), 1000 iterations, Python host
import wxpkg
pkg = wxpkg.load_package()
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
h.step(100); h.sync()
t0 = time.perf_counter(); h.step(1000); h.sync(); dt = time.perf_counter() - t0
rows.append(("C1 100x100 test save, all passes, 1000 iterations", "(tools/run_configs.py)", dt, X * Y * 1000 / dt / 1e6, {}))
rows[-1] = (rows[-1][0], rows[-1][1], dt / 1000 * 1e3, rows[-1][3], {})
bench("C2 4096x1024 dry (pressure+velocity+advection)", ["--workload", "dry", "--X", "4096", "--Y", "1024", "--steps", "1000", "--warmup", "200"])
bench("C3 16384x2048 wet, particles off", ["--steps", "1000", "--warmup", "100"])
bench("C4 grid on ONE GPU: 32768x4096 wet", ["--X", "32768", "--Y", "4096", "--steps", "100", "--warmup", "10"])
bench("C4-dry 32768x4096 dry (north-star kernel size)", ["--workload", "dry", "--X", "32768", "--Y", "4096", "--steps", "200", "--warmup", "20"])
bench("C5 16384x2048 wet + 1 048 576 droplets", ["--particles", "1048576", "--steps", "1000", "--warmup", "500"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "configs.md"), "w") as f:
    f.write("| configuration | bench.py arguments | ms / iteration | Mcell-steps/s | kernels (ms / iteration) |\n|---|---|---|---|---|\n")
    for label, args, ms, val, ker in rows:
        f.write(f"| {label} | `{args}` | {ms:.4f} | {val:.0f} | {', '.join(f'{k} {v:.3f}' for k, v in ker.items())} |\n")
print(open(os.path.join(ROOT, "gpurun_out", "configs.md")).read())
