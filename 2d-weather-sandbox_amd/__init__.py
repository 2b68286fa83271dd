"""MI355X-native engine for the 2D-Weather-Sandbox simulation iteration (hot path only).

The directory name (``2d-weather-sandbox_amd``) is not a Python identifier; import it through
``wxpkg.load_package()`` (tests/conftest.py, __graft_entry__.py, bench.py do), which registers it as
``weather_sandbox_amd``.
"""
from . import codec, engine, params, sim, synth  # noqa: F401
from .sim import WeatherSim  # noqa: F401

__all__ = ["codec", "engine", "params", "sim", "synth", "WeatherSim"]
