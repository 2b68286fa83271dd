#!/usr/bin/env python3
"""Randomised soak (GPU box): waterTexture_0 made on demand (WX_OPT_WATER0_ON_DEMAND 1, the default) against the form the iterations store
themselves (0) -- 24 random sizes / flows / sun angles, random iteration counts, steps cut into WX_OVERLAP_MORE_TO_COME pieces, light textures
read in between; every field compared bit for bit. Usage: python tools/soak_water0.py"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wxpkg
pkg = wxpkg.load_package(); E = pkg.engine
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
bad = 0
for seed in range(24):
    rng = np.random.default_rng(seed)
    X = int(rng.choice([130, 256, 700, 1100, 2200])); Y = int(rng.choice([50, 96, 160, 260]))
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.25, (Y, X)), 0).astype(np.float32)
    base[..., 1] += np.where(air, rng.normal(0, 0.12, (Y, X)), 0).astype(np.float32)
    gui = pkg.params.merge_settings(None); gui["sunAngle"] = float(rng.uniform(-20, 80))
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0); u["enablePrecipitation"] = 0
    p = pkg.params.fill_struct(pkg.params.WxParams(), u)
    hs = []
    for opt in (1, 0):
        h = E.Handle(X, Y, 0); h.set_option(h.OPT_WATER0_ON_DEMAND, opt); h.upload(base, water, wall); h.set_params(p, u["initial_T"]); hs.append(h)
    for rep in range(6):
        k = int(rng.integers(1, 9)); pieces = rng.integers(0, 2)
        for h in hs:
            if pieces and k > 1:
                h.step(k - 1, 4); h.step(1)
            else:
                h.step(k)
        if rng.integers(0, 2): hs[0].read_rect("LIGHT_0", 0, 0, X, int(rng.integers(1, Y)))
        for f in ("WATER_0", "BASE_DISP", "CURL", "BASE_CUR", "WATER_CUR", "LIGHT_0", "LIGHT_1", "WALL_CUR"):
            a, b = hs[0].read_rect(f), hs[1].read_rect(f)
            if not np.array_equal(a, b):
                bad += 1; print("MISMATCH", seed, rep, f, X, Y, np.count_nonzero(a != b))
    for h in hs: h.close()
print("soak done, mismatches:", bad)
