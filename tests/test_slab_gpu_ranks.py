"""Two processes on ONE GPU (gloo transport): the product's multi-rank driver slab.SlabSim on the HIP engine, with
particles, end to end -- ring halo exchange (send/recv) + the partitioned droplet pool (status-flip events all-gathered, edge
droplets in the halos' batch of send / recv) -- against the undecomposed handle. The RCCL run on N GPUs uses exactly this code
with backend "nccl"."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X, Y, HALO, N_ITER, N = 512, 128, 64, 21, 4000


def _problem(pkg):
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    pkg.synth.add_cloud_deck(water, wall)
    rng = np.random.default_rng(9)
    air = wall[..., 1] != 0
    # |v| < 1 cell/iteration (the shaders' documented range) is what the 6-column dependency cone of slab.py assumes
    base[..., 0] += np.where(air, np.clip(rng.normal(0, 0.15, (Y, X)), -0.9, 0.9), 0).astype(np.float32)
    drops = pkg.synth.init_rain_drops(N)
    na = 1500
    drops[:na, 0] = rng.uniform(-1, 1, na).astype(np.float32)
    drops[:300, 0] = (rng.uniform(-8, 8, 300) / X * 2).astype(np.float32)  # around the slab edge in the middle of the domain
    drops[300:500, 0] = np.where(rng.random(200) < 0.5, -1 + rng.uniform(0, 7, 200) * 2 / X, 1 - rng.uniform(0, 7, 200) * 2 / X)
    drops[:na, 1] = rng.uniform(-0.6, 0.2, na).astype(np.float32)
    drops[:na, 2] = rng.uniform(0.1, 1.0, na).astype(np.float32)
    drops[:na, 3] = 0.0
    drops[:na, 4] = 1.0
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=0)
    u["enablePrecipitation"] = 1
    u["inactiveDroplets"] = float(N - na)
    return base, water, wall, drops, u


def _worker(rank, world, port, out_dir, particles):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import wxpkg
    pkg = wxpkg.load_package()
    from weather_sandbox_amd import slab
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, water, wall, drops, u = _problem(pkg)
    if not particles:
        drops, u["enablePrecipitation"] = None, 0
    exact = particles == "exact"
    drv = slab.SlabSim.from_arrays(X, Y, base, water, wall, u, rank, world, torch.device("cuda", 0), halo=HALO, drops=drops, exact=exact)
    assert drv.iters_per_exchange == (6 if particles else 10) and drv.exact == exact
    if exact:
        drv.handle.set_option(drv.handle.OPT_SPLAT_ORDER, 1)
    drv.step(N_ITER)
    drv.exchange()  # settle ownership of the droplets before reading the pool
    drv.sync()
    out = {f: drv.owned(f) for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "PRECIP_FB", "PRECIP_DEP")}
    if particles:
        out["flags"] = drv.handle.pool_flags()
        pool = drv.gather_particles()  # the whole pool, assembled on rank 0
        if rank == 0:
            out["drops"] = pool
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("particles", [False, True, "exact"], ids=["grid", "particles", "particles-exact"])
def test_slab_sim_two_ranks(pkg, tmp_path, particles):
    """slab.SlabSim in two processes on the GPU (gloo transport): grid only (bit-identical), the partitioned droplet pool (atomics: to
    summation order), and WX_OPT_POOL_EXACT with the deterministic splat order: the per-iteration all-gather of slab.py's exact_events
    makes pool, feedback and every field BIT-IDENTICAL to the single handle."""
    import torch
    import torch.multiprocessing as mp
    E = pkg.engine
    E.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), particles), nprocs=2, join=True)
    base, water, wall, drops, u = _problem(pkg)
    if not particles:
        drops, u["enablePrecipitation"] = None, 0
    whole = E.Handle(X, Y, N if particles else 0)
    whole.upload(base, water, wall, drops)
    whole.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    exact = particles == "exact"
    if exact:
        whole.set_option(whole.OPT_SPLAT_ORDER, 1)
    whole.step(N_ITER)
    assert np.abs(whole.read_rect("BASE_CUR")[..., :2]).max() < 1.0  # precondition of the 6-column cone
    xo = X // 2
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        if particles and r == 0:
            d_ref = whole.read_particles()
            assert (d_ref[:, 2] >= 0).sum() > 300
            assert np.array_equal(got["drops"][:, 2] >= 0, d_ref[:, 2] >= 0)
            assert np.array_equal(got["drops"], d_ref) if exact else np.abs(got["drops"] - d_ref).max() <= 1e-6
            flags = np.stack([np.load(os.path.join(str(tmp_path), f"rank{q}.npz"))["flags"] for q in range(2)])
            assert np.array_equal((flags == 2).sum(0), (d_ref[:, 2] >= 0).astype(int))  # one owner per active droplet
            assert (flags[:, d_ref[:, 2] < 0] == 1).all()  # inactive records: on every rank
        assert np.array_equal(got["WALL_CUR"], whole.read_rect("WALL_CUR")[:, r * xo:(r + 1) * xo])
        for f in ("BASE_CUR", "WATER_CUR", "PRECIP_FB", "PRECIP_DEP"):
            a, b = got[f], whole.read_rect(f)[:, r * xo:(r + 1) * xo]
            if f == "PRECIP_FB" and r == 0:
                a, b = a.copy(), b.copy()
                a[0, :2], b[0, :2] = 0, 0  # the reference's mailbox texels are not kept on slabs
            d = np.abs(a - b).max(-1)
            assert d.max() <= (1e-6 * max(1.0, np.abs(b).max()) if particles and not exact else 0.0), (f, r, d.max(), np.nonzero((d > 1e-6).any(0))[0][:20])


@pytest.mark.parametrize("workload,grid", [("wet", (2048, 256)), ("dry", (4096, 512))])
def test_bench_verify_two_ranks(workload, grid):
    """bench.py --gpus 2 --verify end to end (two ranks on the one GPU of the box, gloo transport): P2P self-test, slabs of the wet
    grid / of the north-star dry stencil with the overlapped exchange, then every rank's owned-column checksums against the
    undecomposed run on rank 0 -- the line a multi-GPU node will print says "verify": "ok" only if the decomposed result is bit
    for bit the single-GPU one."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WX_BENCH_SHARE_GPU="1", WX_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8", "--X", str(grid[0]),
           "--Y", str(grid[1]), "--workload", workload, "--verify", "--no-pmc"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["verify"] == "ok", d["verify"]
    assert d["ranks_seen"] == 2 and d["n_gpus"] == 2 and d["value"] > 0
    assert d["flow"]["end_of_timed_region"]["rms_v"] > 0.01  # the fluid moved
    # (round 6) what a scaling curve decomposes into, in the same line: the undecomposed grid on the same box, the slab without any exchange
    assert d["strong_scaling_vs_n1"]["n1_value"] > 0 and d["strong_scaling_vs_n1"]["speedup"] > 0, d.get("strong_scaling_vs_n1")
    assert d["per_rank"]["plain_slab_ms_per_step"] > 0 and d["per_rank"]["local_columns"] > d["per_rank"]["owned_columns"], d.get("per_rank")
    if workload == "dry":
        # the water-free marching kernel ran on the slabs -- in pairs: agreed water-free slabs run their periods in order (ABI 10)
        assert d["roofline"]["kernel"] == "march_dry2_two_iterations_per_launch"
