#!/usr/bin/env python3
"""Does the kernel time depend on WHERE the planes were allocated? Several handles of the same grid alive at once (different device
addresses), each timed alone, interleaved; prints kernel time and the device addresses of a few planes."""
# (tuning environment switches exist only in the -DWX_DEBUG build of the library: make -C 2d-weather-sandbox_amd/csrc debug)
import os as _os
_dbg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "2d-weather-sandbox_amd", "csrc", "variants", "libwxsim_debug.so")
if "WXSIM_LIB" not in _os.environ and _os.path.exists(_dbg):
    _os.environ["WXSIM_LIB"] = _dbg
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import wxpkg  # noqa: E402

pkg = wxpkg.load_package()
X, Y = 16384, 2048
# one handle per argument "arena:skew" (arena 0 = one hipMalloc per plane), e.g.  0:0 0:0 1:0 1:0 1:256 1:4352
SPECS = [a.split(":") for a in sys.argv[1:]] or [["0", "0"]] * 6


def make(spec):
    os.environ["WX_ARENA"], os.environ["WX_ARENA_SKEW"] = spec[0], spec[1]
    os.environ["WX_ARENA_CONTIG"] = spec[2] if len(spec) > 2 else "0"
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 50.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 0
    h = pkg.engine.Handle(X, Y, 0)
    h.setup_columns(pkg.synth.terrain_columns(X, Y))
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(20)
    h.sync()
    return h


def timed(h, steps=100):
    h.profile(True)
    h.sync()
    for _ in range(steps // 10):
        h.step(10)
    h.sync()
    p = h.profile_read()
    h.profile(False)
    ms, n = p["march_wet_full_iteration"]
    return ms / n


hs = [make(sp) for sp in SPECS]
res = [[] for _ in hs]
for rep in range(3):
    for i, h in enumerate(hs):
        res[i].append(timed(h))
for (h, r), sp in zip(zip(hs, res), SPECS):
    ptrs = [h.device_ptr(f) for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "CURL", "PRECIP_FB")]
    print(f"arena {sp[0]} skew {sp[1]:>8s}:", " ".join(f"{t:.4f}" for t in r), " ".join(f"{p:#x}" for p in ptrs))

# placement tuning on the first handle: candidates tried, then the sustained time of the winner
if os.environ.get("WX_PROBE_TUNE"):
    print("wx_tune_placement(tries=8, iters=30) on handle 0: first candidate / winner ms:", hs[0].tune_placement(8, 30), " sustained:", " ".join(f"{timed(hs[0]):.4f}" for _ in range(3)))
