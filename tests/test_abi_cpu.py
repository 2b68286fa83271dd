"""CPU-side checks of the C ABI: the library loads, exports every symbol include/wxsim.h declares, and refuses
to run without a GPU (no CPU fallback). No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine(pkg):
    from weather_sandbox_amd import engine as E
    E.build()
    return E


def test_header_symbols_all_exported(engine):
    hdr = open(os.path.join(ROOT, "include", "wxsim.h")).read()
    declared = set(re.findall(r"\b(wx_[a-z_]+)\s*\(", hdr))
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    L = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} not exported by libwxsim.so"
    assert engine.lib().wx_abi_version() == 11


def test_params_struct_matches_header(engine, pkg):
    """Field order of the ctypes struct == field order of `struct wx_params` in the header."""
    hdr = open(os.path.join(ROOT, "include", "wxsim.h")).read()
    body = hdr[hdr.index("typedef struct wx_params {"):hdr.index("} wx_params;")]
    body = re.sub(r"/\*.*?\*/", "", body.split("{", 1)[1], flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        m = re.match(r"(?:float|int32_t|uint32_t)\s+(.*)", decl, flags=re.S)
        if m:
            names += [re.sub(r"\[.*\]", "", n).strip() for n in m.group(1).split(",")]
    assert names == [f[0] for f in pkg.params.WxParams._fields_]
    assert C.sizeof(pkg.params.WxParams) == 4 * (33 + 4 + 2 + 2 + 4 + 3)


def test_kernel_names(engine):
    L = engine.lib()
    names = [L.wx_kernel_name(k).decode() for k in range(L.wx_kernel_count())]
    assert "advection" in names and "boundary" in names and len(set(names)) == len(names)


def test_no_gpu_means_loud_failure(engine):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.WxError) as ei:
        engine.Handle(64, 32, 0)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_create_rejects_bad_geometry(engine):
    L = engine.lib()
    h = C.c_void_p()
    assert L.wx_create(0, 10, 0, C.byref(h)) == -1
    assert L.wx_create(100, 100, -5, C.byref(h)) == -1
    assert L.wx_create_slab(100, 100, 0, 50, 0, 0, C.byref(h)) == -1  # slab without halo
    assert b"halo" in L.wx_last_error(None)
    assert L.wx_step(None, 1) == -1 and L.wx_sync(None) == -1


def test_graft_entry_build_runs():
    """__graft_entry__.build() is the driver's 'does it build' check: it must succeed on a CPU-only box."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
