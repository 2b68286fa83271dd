"""Bad arguments through the C ABI on a LIVE handle: every call answers with an error code (and a message), none takes the process or
the device down -- run in a child process so that a fault would fail this test instead of ending the session. (The argument checks
that need no device are in tests/test_abi_cpu.py.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, sys
    import numpy as np
    sys.path.insert(0, %r)
    import wxpkg
    pkg = wxpkg.load_package()
    E = pkg.engine
    L = E.lib()
    X, Y, N = 200, 64, 300
    base, water, wall = pkg.synth.terrain_grid(X, Y)
    drops = pkg.synth.init_rain_drops(N)
    u = pkg.params.uniforms_from_gui(pkg.params.merge_settings(None), Y, quad_scale=0)
    h = E.Handle(X, Y, N)
    h.upload(base, water, wall, drops)
    h.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h.step(3)                       # (the marching kernel's display iteration ran: BASE_DISP / LIGHT_* / WATER_0 are made on demand from here on)
    hp = h._h
    buf = (C.c_float * (X * Y * 4 + 64))()
    big = 2**31 - 1
    bad = []
    def must_fail(what, rc):
        if rc == 0:
            bad.append(what)
    names = ["BASE_CUR", "BASE_DISP", "WATER_0", "WATER_CUR", "WALL_CUR", "LIGHT_0", "LIGHT_1", "CURL", "PRECIP_FB", "PRECIP_DEP", "EMITTED"]
    rects = [(0, 10**6, 4, 4), (0, -5, 4, 4), (-1, 0, 4, 4), (0, 0, 0, 4), (0, 0, 4, 0), (X - 1, 0, 2, 1), (0, Y - 1, 1, 2), (1, 0, big, 1), (0, 1, 1, big),
             (big, big, big, big), (0, Y, 1, 1), (X, 0, 1, 1), (0, 0, X + 1, Y), (0, 0, -3, -3)]
    for name in names:
        for r in rects:
            try:
                h.read_rect(name, *r)
                bad.append(("read_rect", name, r))
            except E.WxError:
                pass
            except (ValueError, MemoryError, OverflowError):  # (the Python wrapper's own allocation of an absurd rectangle)
                pass
    for r in rects:
        must_fail(("stream_frame", r), L.wx_stream_frame(hp, r[0], r[1], r[2], r[3], buf))
    must_fail("read_rect unknown field", L.wx_read_rect(hp, 999, 0, 0, 1, 1, buf, 0))
    must_fail("read_rect NULL dst", L.wx_read_rect(hp, 0, 0, 0, 1, 1, None, 0))
    must_fail("read_particles range", L.wx_read_particles(hp, N - 1, 5, buf))
    must_fail("read_particles overflow", L.wx_read_particles(hp, 1, big, buf))
    must_fail("read_particles negative", L.wx_read_particles(hp, -1, 1, buf))
    must_fail("step negative", L.wx_step(hp, -1))
    must_fail("set_option unknown", L.wx_set_option(hp, 12345, 1))
    must_fail("set_option negative cap", L.wx_set_option(hp, 6, -1))
    must_fail("upload NULL", L.wx_upload(hp, None, None, None, None))
    must_fail("set_params NULL", L.wx_set_params(hp, None, None, None, None, None))
    must_fail("create zero", L.wx_create(0, 0, 0, C.byref(C.c_void_p())))
    must_fail("create negative drops", L.wx_create(64, 64, -1, C.byref(C.c_void_p())))
    must_fail("slab step on a whole-domain handle", L.wx_slab_step(hp, 1))
    must_fail("exchange on a whole-domain handle", L.wx_exchange(hp))
    must_fail("tune_placement bad tries", L.wx_tune_placement(hp, -1, 0, None, None))
    # ... and the handle still works, bit for bit like one that was never misused
    h2 = E.Handle(X, Y, N)
    h2.upload(base, water, wall, drops)
    h2.set_params(pkg.params.fill_struct(pkg.params.WxParams(), u), u["initial_T"])
    h2.step(3)
    h.step(4); h2.step(4)
    for f in ("BASE_CUR", "WATER_CUR", "WALL_CUR", "LIGHT_1", "BASE_DISP"):
        assert np.array_equal(h.read_rect(f), h2.read_rect(f)), f
    assert np.array_equal(h.read_particles(), h2.read_particles())
    print("ACCEPTED:", bad)
    print("MISUSE-OK" if not bad else "MISUSE-ACCEPTED")
''') % ROOT


def test_bad_arguments_are_refused_not_fatal():
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "MISUSE-OK" in r.stdout, r.stdout[-3000:]
