"""GPU box: long run of the default (row-marching) kernel set against the per-pass cross-check with particles on -- not bit-comparable
(the splat atomics add in a different order), so global statistics are compared: finite fields, conserved-ish totals, droplet activity."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wxpkg

def run(fused, X, Y, n_drops, iters):
    os.environ["WX_FUSED"] = str(fused)
    pkg = wxpkg.load_package()
    E = pkg.engine
    h = E.Handle(X, Y, n_drops)
    h.setup_columns(pkg.synth.terrain_columns(X, Y, cloud_deck=True), pkg.synth.init_rain_drops(n_drops))
    gui = pkg.params.merge_settings(None)
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    done = 0
    while done < iters:
        h.step(10)
        done += 10
    b, w = h.read_rect("BASE_CUR"), h.read_rect("WATER_CUR")
    d = h.read_particles()
    fbk = h.read_rect("PRECIP_FB")
    out = {"finite": bool(np.isfinite(b).all() and np.isfinite(w).all() and np.isfinite(d).all() and np.isfinite(fbk).all()),
           "mean_T": float(b[..., 3].astype(np.float64).mean()), "vmax": float(np.abs(b[..., :2]).max()),
           "vapour": float(w[..., 0].astype(np.float64).sum()), "cloud": float(w[..., 1].astype(np.float64).sum()),
           "active": int((d[:, 2] >= 0).sum()), "lightning": h.read_rect("LIGHTNING").tolist(), "iter": h.iter}
    h.close()
    return out

if __name__ == "__main__":
    X, Y, n, it = 4096, 1024, 200000, int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    a, b = run(2, X, Y, n, it), run(0, X, Y, n, it)
    print(json.dumps({"march": a, "perpass": b}, indent=1))
    assert a["finite"] and b["finite"]
    for k in ("mean_T", "vapour", "cloud"):
        assert abs(a[k] - b[k]) <= 2e-3 * abs(b[k]) + 1e-6, k
    assert abs(a["active"] - b["active"]) <= 0.02 * max(b["active"], 1) + 50
    print("soak ok")
