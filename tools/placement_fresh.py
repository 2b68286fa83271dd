#!/usr/bin/env python3
"""The allocation lottery without PyTorch in the process: one handle of the metric grid created as the process's FIRST device
allocations (engine over ctypes only), its iteration timed at rest. PRE=n first makes n odd-sized device allocations (3 .. 87 MB, every
third one freed again) through the HIP runtime, the way a host that already uses the GPU would have. Run it in fresh processes:
  for i in 1 2 3 4 5 6; do python tools/placement_fresh.py; PRE=40 python tools/placement_fresh.py; done"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
X, Y = 16384, 2048
pre = int(os.environ.get("PRE", "0"))
big = int(os.environ.get("PRE_BIG_MB", "0"))
if pre or big:
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    held = []
    if big:  # one large allocation first (what a framework's caching allocator holds)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), big << 20) == 0
        held.append(p)
    for k in range(pre):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), (3 + 14 * (k % 7)) << 20) == 0
        if k % 3 == 2:
            hip.hipFree(p)
        else:
            held.append(p)
gui = pkg.params.merge_settings(None)
gui["sunAngle"] = 50.0
u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
u["enablePrecipitation"] = 0
h = pkg.engine.Handle(X, Y, 0)
h.setup_columns(pkg.synth.terrain_columns(X, Y))
h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
h.step(300)
h.sync()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        h.step(10)
    h.sync()
    res.append((time.perf_counter() - t0) / 200 * 1e3)
tuned = ""
if os.environ.get("TUNE"):
    tuned = f"  tune {h.tune_placement(int(os.environ['TUNE']))}"
print(f"PRE={pre} BIG={big}  ms/iteration at rest " + " ".join(f"{r:.4f}" for r in res) + tuned, flush=True)
