"""Multi-process (world_size 2 and 3, gloo, CPU) test of the slab decomposition + ring halo exchange.

The driver under test is the product's ``slab.SlabSim`` (exchange pattern, iterations-per-exchange logic, slab
geometry); the compute engine plugged into it here is the CPU oracle (tests may use the oracle as checker), so
the test proves: N slabs + halo exchange == the undecomposed domain, BIT FOR BIT, for all grid fields.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HALO_FIELDS = ("BASE_CUR", "WATER_CUR", "LIGHT_0", "LIGHT_1", "WALL_CUR")


class OracleSlabEngine:
    """slab-engine interface on top of the CPU oracle (x_off / X_global geometry, numpy halo pack)."""

    def __init__(self, wx_oracle, X_global, Y, x0, X_owned, halo):
        self.halo, self.Y = halo, Y
        self.X = X_owned + 2 * halo
        self.sim = wx_oracle.OracleSim(self.X, Y, 0, X_global=X_global, x_off=x0 - halo)

    def upload(self, base, water, wall):
        self.sim.upload(base, water, wall)

    def set_params(self, u):
        self.sim.set_params(u)

    def new_buffer(self):
        n = self.halo * self.Y * (4 * 16 + 4)
        return torch.zeros(n, dtype=torch.uint8)

    def _cols(self, side, pack):
        h, X = self.halo, self.X
        if pack:
            return slice(h, 2 * h) if side == 0 else slice(X - 2 * h, X - h)
        return slice(0, h) if side == 0 else slice(X - h, X)

    def pack(self, side, buf):
        c = self._cols(side, True)
        parts = [np.ascontiguousarray(self.sim.view(f)[:, c]).view(np.uint8).reshape(-1) for f in HALO_FIELDS]
        buf.copy_(torch.from_numpy(np.concatenate(parts)))

    def unpack(self, side, buf):
        c = self._cols(side, False)
        raw = buf.numpy()
        off = 0
        for f in HALO_FIELDS:
            v = self.sim.view(f)
            n = self.Y * self.halo * v.shape[2] * v.dtype.itemsize
            v[:, c] = raw[off:off + n].view(v.dtype).reshape(self.Y, self.halo, v.shape[2])
            off += n

    def step(self, n):
        self.sim.step(n)

    def sync(self):
        pass


def _worker(rank, world, port, X, Y, halo, n_iter, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wx_oracle
    import wxpkg
    pkg = wxpkg.load_package()
    from weather_sandbox_amd import slab
    base, water, wall, u = _problem(pkg, X, Y)
    x0, xo = slab.slab_columns(X, rank, world)
    idx = (x0 - halo + np.arange(xo + 2 * halo)) % X
    eng = OracleSlabEngine(wx_oracle, X, Y, x0, xo, halo)
    eng.upload(np.ascontiguousarray(base[:, idx]), np.ascontiguousarray(water[:, idx]), np.ascontiguousarray(wall[:, idx]))
    eng.set_params(u)
    drv = slab.SlabSim(eng, rank, world, halo)
    assert drv.iters_per_exchange == halo // 6
    drv.step(n_iter)
    drv.step(3)  # a second call that does not line up with the exchange period
    owned = {f: eng.sim.field(f)[:, halo:halo + xo] for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1", "WATER_0")}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **owned)
    dist.barrier()
    dist.destroy_process_group()


def _problem(pkg, X, Y):
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    rng = np.random.default_rng(11)
    air = wall[..., 1] != 0
    base[..., 0] += np.where(air, rng.normal(0, 0.2, (Y, X)), 0).astype(np.float32)  # |v| up to ~0.8: widest cone
    base[..., 1] += np.where(air, rng.normal(0, 0.1, (Y, X)), 0).astype(np.float32)
    water[..., 0] *= np.where(air, 1.0 + 0.4 * rng.random((Y, X)), 1.0).astype(np.float32)
    gui = pkg.params.merge_settings(None)
    gui["sunAngle"] = 35.0
    u = pkg.params.uniforms_from_gui(gui, Y, quad_scale=1)  # x-dependent fragCoord: exercises x_off / X_global
    u["enablePrecipitation"] = 0
    return base, water, wall, u


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,halo", [(2, 12), (3, 6), (2, 7)])
def test_slabs_equal_whole_domain(pkg, oracle, tmp_path, world, halo):
    X, Y, n_iter = 96, 48, 8
    mp.spawn(_worker, args=(world, _free_port(), X, Y, halo, n_iter, str(tmp_path)), nprocs=world, join=True)
    base, water, wall, u = _problem(pkg, X, Y)
    ref = oracle.OracleSim(X, Y, 0)
    ref.upload(base, water, wall)
    ref.set_params(u)
    ref.step(n_iter + 3)
    xo = X // world
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        for f in got.files:
            assert np.array_equal(got[f], ref.field(f)[:, r * xo:(r + 1) * xo]), f"rank {r} field {f}"


def test_slab_geometry(pkg):
    from weather_sandbox_amd import slab
    assert slab.slab_columns(16384, 3, 8) == (6144, 2048)
    with pytest.raises(ValueError):
        slab.slab_columns(100, 0, 3)
    with pytest.raises(ValueError):
        slab.SlabSim(object(), 0, 2, 4)
