"""Import helper: registers the package directory ``2d-weather-sandbox_amd`` as ``weather_sandbox_amd``."""
import importlib.util
import os
import sys

_NAME = "weather_sandbox_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "2d-weather-sandbox_amd")


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
