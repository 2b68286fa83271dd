import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import wxpkg  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = wxpkg.load_package()
    try:  # (the implicit placement search of big whole-domain handles -- a quarter of a second and more per handle -- only where a test asks for it)
        p.engine.lib().wx_set_option(None, p.engine.Handle.OPT_PLACEMENT_SEARCH, 0)
    except Exception:
        pass
    return p


@pytest.fixture(scope="session")
def oracle():
    import wx_oracle
    wx_oracle.build()
    return wx_oracle


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    u = json.loads(str(g["uniforms_json"]))
    u["initial_T"] = g["initial_T"]
    for k in ("sounding_T", "sounding_W", "sounding_Vel"):  # realWorldSounding_* arrays (only the sounding fixture has them)
        if k in g.files:
            u[k] = g[k]
    for k in ("userInputValues", "userInputMove", "airplaneValues"):
        u[k] = tuple(u[k])
    return g, u


@pytest.fixture(scope="session")
def golden():
    return load_golden
