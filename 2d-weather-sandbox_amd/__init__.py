"""MI355X-native engine for the 2D-Weather-Sandbox simulation iteration (hot path only).

The directory name (``2d-weather-sandbox_amd``) is not a Python identifier; import it through
``__graft_entry__.load_package()`` (or ``tests/conftest.py``), which registers it as
``weather_sandbox_amd``.
"""
from . import codec, params  # noqa: F401

__all__ = ["codec", "params"]
