"""The launch shape of the row-marching wet kernel (csrc/wx_wet.h: wet_launch_shape) as pure host logic: a small hipcc-built
harness prints the segment tables for a list of grids; every row of every band must be covered exactly once by non-empty segments,
and short segments come last. No GPU needed (the capacity falls back to 256 CUs x 12 waves)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

GRIDS = [(16384, 2048), (32768, 4096), (4192, 4096), (2144, 2048), (8240, 2048), (4096, 1024), (10781, 523), (11000, 800), (2150, 1030),
         (1000, 600), (2500, 300), (7990, 301), (8000, 500), (10000, 480), (256, 96), (100, 100), (64, 8), (2, 4), (130, 50), (4100, 20), (512, 512), (57, 511), (56, 513)]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("shape") / "shape_harness")
    # -DWX_DEBUG: the tuning environment switches exist only in debug builds (the shipped library reads none)
    subprocess.check_call([HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", "-DWX_DEBUG", "-Wno-unused-value", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "shape_harness.hip")])
    return exe


def _shapes(exe, env=None):
    args = [str(v) for g in GRIDS for v in g]
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("WX_WET_"):
            del e[k]
    e.update(env or {})
    return json.loads(subprocess.check_output([exe] + args, env=e))


@pytest.mark.parametrize("env", [None, {"WX_WET_BANDS": "0"}, {"WX_WET_NOTAIL": "1"}, {"WX_WET_BANDS": "2"}, {"WX_WET_ROUNDS": "1"},
                                 {"WX_WET_SPEC": "5x1,2x0.5,1x0.25"}, {"WX_WET_ALPHA": "2"}])
def test_segment_tables_cover_every_row_once(harness, env):
    for sh in _shapes(harness, env):
        X, Y, st, n = sh["X"], sh["Y"], sh["start"], sh["n_seg"]
        assert sh["n_strips"] == (X + 55) // 56
        assert len(st) == n + 1 and st[0] == 0 and n >= 1
        assert all(b > a for a, b in zip(st, st[1:])), (X, Y, st)  # no empty segment, ascending
        height = (Y + 7) // 8 if sh["bands"] else Y  # the table of a band is clipped to the band's own height by the kernel
        assert st[-1] == height, (X, Y, st)
        if sh["bands"]:
            assert Y >= 16
            for k in range(8):  # every band [k*Y/8, (k+1)*Y/8) is covered by the clipped table
                lo, hi = k * Y // 8, (k + 1) * Y // 8
                assert 0 < hi - lo <= height


def test_default_shape_has_a_short_tail_on_the_metric_grid(harness):
    sh = {(s["X"], s["Y"]): s for s in _shapes(harness)}
    s = sh[(16384, 2048)]
    assert s["bands"] == 1
    lens = [b - a for a, b in zip(s["start"], s["start"][1:])]
    assert lens[-1] < lens[-2] < lens[0] and lens[0] >= 48  # full segments first, the shortest last
    assert sh[(256, 96)]["bands"] == 0  # low grids keep the column blocks
    # grids below 512 rows: bands while they are at least 128 rows high and fewer than 146 strips wide (profiles/r06_ref_sizes_minrows.txt)
    assert sh[(2500, 300)]["bands"] == 1 and sh[(7990, 301)]["bands"] == 1 and sh[(8000, 500)]["bands"] == 1 and sh[(10000, 480)]["bands"] == 0


def test_halved_shape_covers_the_same_rows(harness):
    """wet_shape_halved (the edge strips of a slab, launched next to its interior strips): every border of the shape it was made
    from is still a border, every new border lies strictly inside a segment, nothing is empty."""
    for env in (None, {"WX_WET_BANDS": "0"}, {"WX_WET_SPEC": "5x1,2x0.5,1x0.25"}):
        for sh in _shapes(harness, env):
            st, hv = sh["start"], sh["halved_start"]
            assert len(hv) == sh["halved_n_seg"] + 1 and hv[0] == st[0] and hv[-1] == st[-1]
            assert all(b > a for a, b in zip(hv, hv[1:])), (sh["X"], sh["Y"], hv)
            assert set(st) <= set(hv)
            assert len(hv) <= 129
            for a, b in zip(st, st[1:]):
                inner = [v for v in hv if a < v < b]
                assert len(inner) == (1 if b - a >= 12 else 0) or len(hv) == len(st), (a, b, inner)
