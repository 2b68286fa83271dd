"""Host-side mirror of the reference's simulation seam.

The reference has no class for this: ``mainScript(base, water, wall, drops)`` (app.js:1495) creates
module-scope GL objects, ``draw()`` (app.js:5686) runs ``guiControls.IterPerFrame`` iterations per frame
(app.js:5830-6005), consumers call ``gl.readPixels`` and ``prepareDownload()`` writes a save
(app.js:6575-6628). ``WeatherSim`` restates exactly that surface on top of the C ABI (engine.py).
"""
from __future__ import annotations

import datetime as _dt
import sys
from typing import Any, Dict, Optional

import numpy as np

from . import codec, params
from .engine import Handle

TIME_PER_ITERATION = 0.00008  # hours of simulated time per iteration (app.js:449)
_EPOCH = _dt.datetime(1970, 1, 1)


def _js_date(year: int, month_index: int, day: int, hour: int = 0, minute: int = 0, second: int = 0) -> _dt.datetime:
    """``new Date(year, monthIndex, day, h, m, s)`` with JavaScript's roll-over rules (month index 12 = January of the
    next year, day 0 = last day of the previous month); the arguments are already truncated to integers."""
    year += month_index // 12
    return _dt.datetime(year, month_index % 12 + 1, 1) + _dt.timedelta(days=day - 1, hours=hour, minutes=minute, seconds=second)


def initial_sim_datetime(month: float, time_of_day: float, day_night_cycle: bool) -> _dt.datetime:
    """startSimulation()'s clock (app.js:3902-3910): ``new Date(2000, floor(month) - 1, (month % 1) * 30.417)`` -- the
    Date constructor truncates the fractional day -- and, with the day/night cycle on, ``onUpdateTimeOfDaySlider`` /
    ``onUpdateMonthSlider`` (app.js:6494-6507): ``setHours(timeOfDay, (timeOfDay % 1) * 60)`` then
    ``setMonth(month - 0.96, ((month - 0.96) % 1) * 30)``, every argument truncated toward zero."""
    import math
    t = _js_date(2000, math.floor(month) - 1, int((month % 1) * 30.417))
    if day_night_cycle:
        t = t.replace(hour=0, minute=0) + _dt.timedelta(hours=int(time_of_day), minutes=int((time_of_day % 1) * 60))
        m = month - 0.96
        t = _js_date(t.year, int(m), int((m % 1) * 30), t.hour, t.minute, t.second)
    return t


def advance_sim_datetime(t: _dt.datetime, delta_hours: float):
    """The clock part of ``updateSunlight(deltaT_hours)`` (app.js:6513-6516): ``new Date(getTime() + deltaT_hours * 3600 * 1000)`` -- a
    Date holds whole milliseconds, the sum is truncated -- and the two sliders recomputed from it. Returns (t, timeOfDay, month)."""
    ms = int((t - _EPOCH) / _dt.timedelta(milliseconds=1) + delta_hours * 3600 * 1000)
    t = _EPOCH + _dt.timedelta(milliseconds=ms)
    return t, t.hour + t.minute / 60.0 + t.second / 3600.0, t.month + t.day / 30.5 + t.hour / 720.0


class WeatherSim:
    def __init__(self, X: int, Y: int, base, water, wall, droplets=None, settings: Optional[Dict[str, Any]] = None, *,
                 sun_angle_deg: Optional[float] = None, quad_scale: int = 0, pass_mask: int = params.PASS_ALL, columns=None):
        """``mainScript``: take the four initial arrays + saved settings (app.js:1495, 3375-3399, 5189-5317), or -- for a
        new simulation -- the 1-D setup descriptors ``columns`` (synth.terrain_columns) that the device expands."""
        self.X, self.Y = int(X), int(Y)
        self.gui = params.merge_settings(settings)
        n_drops = 0 if droplets is None else int(np.asarray(droplets).size // 5)
        self._h = Handle(self.X, self.Y, n_drops)
        if columns is not None:
            self._h.setup_columns(columns, droplets)
        else:
            self._h.upload(base, water, wall, droplets)
        self._quad_scale = int(quad_scale)
        self._pass_mask = int(pass_mask)
        self._manual_sun = sun_angle_deg
        self._inactive_pushed = False
        self._placement_told = False
        self.verbose = True
        # startSimulation(): clock from the saved month / time of day (app.js:3902-3910)
        self.sim_datetime = initial_sim_datetime(float(self.gui["month"]), float(self.gui["timeOfDay"]), bool(self.gui.get("dayNightCycle")))
        self.brush = {"userInputType": -1, "userInputValues": (0.0, 0.0, 0.0, 0.0), "userInputMove": (0.0, 0.0)}
        self.airplane = (0.0, 0.0, 0.0, 0.0)
        self._push_uniforms()

    # ---- construction helpers ----
    @classmethod
    def new_simulation(cls, X: int, Y: int, settings: Optional[Dict[str, Any]] = None, *, n_droplets: Optional[int] = None,
                       seed: float = 0.5, height_mult: float = 0.3, **kw) -> "WeatherSim":
        """Start-up without a save file: the setup pass (setupShader.frag:36-92) + ``initRainDrops`` (app.js:4901-4913),
        one droplet per 25 cells like the reference (the save format relies on that count)."""
        from . import synth
        gui = params.merge_settings(settings)
        n = codec.num_droplets(X, Y) if n_droplets is None else int(n_droplets)
        cols = synth.terrain_columns(X, Y, gui, seed=seed, height_mult=height_mult)
        return cls(X, Y, None, None, None, synth.init_rain_drops(n) if n else None, settings, columns=cols, **kw)

    @classmethod
    def from_save(cls, sf: "codec.SaveFile | str", **kw) -> "WeatherSim":
        """``loadData()`` (app.js:1256-1366)."""
        if isinstance(sf, str):
            sf = codec.load(sf)
        return cls(sf.X, sf.Y, sf.base, sf.water, sf.wall, sf.droplets, sf.settings, **kw)

    # ---- parameters ----
    def uniforms(self) -> Dict[str, Any]:
        u = params.uniforms_from_gui(self.gui, self.Y, sun_angle_deg=self._manual_sun, quad_scale=self._quad_scale,
                                     pass_mask=self._pass_mask)
        u.update(self.brush)
        u["airplaneValues"] = self.airplane
        # keep the engine's own 600-iteration measurement after the first push (app.js:5957-5966)
        u["inactiveDroplets"] = -1.0 if self._inactive_pushed else 0.0
        if getattr(self, "_sounding", None) is not None:
            u["sounding_T"], u["sounding_W"], u["sounding_Vel"] = self._sounding
        return u

    def _push_uniforms(self):
        u = self.uniforms()
        p = params.fill_struct(params.WxParams(), u)
        self._h.set_params(p, u["initial_T"], u.get("sounding_T"), u.get("sounding_W"), u.get("sounding_Vel"))
        self._inactive_pushed = True

    def set_sounding(self, raw_sounding):
        """Load a real sounding for the ``soundingForcing`` slider (app.js:5444-5463); ``raw_sounding`` as in
        ``params.sounding_arrays`` (scraper order: top of the sounding first)."""
        sim_h = float(self.gui["simHeight"])
        self._sounding = params.sounding_arrays(raw_sounding, self.Y, sim_h, sim_h * float(self.gui["dryLapseRate"]) / 1000.0)
        self._push_uniforms()

    def set_gui(self, **changes):
        """Change guiControls entries and push the uniforms (dat.GUI onChange + setGuiUniforms, app.js:3401-3443)."""
        for k in changes:
            if k not in params.GUI_DEFAULTS:
                raise KeyError(k)
        self.gui.update(changes)
        self._push_uniforms()

    def set_brush(self, input_type: int, x: float, y: float, intensity: float, brush_size: float, move=(0.0, 0.0)):
        """Per-frame brush uniforms (app.js:5749-5808); input_type -1 = mouse released."""
        self.brush = {"userInputType": int(input_type), "userInputValues": (x, y, intensity, brush_size * 0.5),
                      "userInputMove": tuple(move)}
        self._push_uniforms()

    def update_sunlight(self, delta_hours: Optional[float]):
        """``updateSunlight(deltaT_hours)`` (app.js:6510-6561): advance the clock, recompute the sun."""
        if delta_hours is not None:
            self.sim_datetime, self.gui["timeOfDay"], self.gui["month"] = advance_sim_datetime(self.sim_datetime, delta_hours)
        self.gui["sunAngle"] = params.sun_angle_from_time(self.gui["timeOfDay"], self.gui["month"], self.gui["latitude"])
        self._manual_sun = None
        self._push_uniforms()

    # ---- the frame loop ----
    def step(self, n_iter: Optional[int] = None):
        """Simulation part of ``draw()``: sun update for the frame, then n iterations (app.js:5814-6005)."""
        n = int(self.gui["IterPerFrame"]) if n_iter is None else int(n_iter)
        if self.gui.get("dayNightCycle") and self._manual_sun is None:
            self.update_sunlight(TIME_PER_ITERATION * n)
        self._h.step(n)
        if not self._placement_told:  # the engine looked for a fast placement of its planes inside the first step of a big grid: say so once
            self._placement_told = True
            pi = self._h.placement_info()
            if pi is not None and self.verbose:
                print(f"[wxsim] placement search: {pi[0]:.4f} ms / iteration on the first allocations, {pi[1]:.4f} kept", file=sys.stderr)

    def sync(self):
        self._h.sync()

    @property
    def iter_num(self) -> int:
        return self._h.iter

    @iter_num.setter
    def iter_num(self, v: int):
        self._h.iter = v

    # ---- readback (gl.readPixels / getBufferSubData call sites) ----
    def read_rect(self, field: str, x=0, y=0, w=None, h=None, **kw):
        return self._h.read_rect(field, x, y, w, h, **kw)

    def read_particles(self, first=0, count=None):
        return self._h.read_particles(first, count)

    def measure_station(self, x: int, y: int):
        """Weatherstation.measure (app.js:1084-1092): FB0 base 1x3 and water 1x2 starting at (x, y-1)."""
        return self.read_rect("BASE_CUR", x, y - 1, 1, 3), self.read_rect("WATER_0", x, y - 1, 1, 2)

    def sounding_column(self, x: int):
        """soundingGraph.draw (app.js:3931-3943): FB1 column reads; wall as Int32."""
        return (self.read_rect("BASE_DISP", x, 0, 1, self.Y), self.read_rect("WATER_CUR", x, 0, 1, self.Y),
                self.read_rect("WALL_DISP", x, 0, 1, self.Y, int32=True))

    def inactive_droplets(self) -> float:
        """readPixels(0,0) of the feedback texture (app.js:5958-5961)."""
        return float(self.read_rect("PRECIP_FB", 0, 0, 1, 1)[0, 0, 0])

    def lightning(self):
        return self.read_rect("LIGHTNING")

    def to_save(self) -> codec.SaveFile:
        """``prepareDownload()`` (app.js:6584-6613): FB0 = base_0, water_0 (post-boundary!), wall_0 + particles.
        Deviation: the reference always stores particle buffer 0; this stores the current buffer."""
        gui = {k: v for k, v in self.gui.items()}
        return codec.SaveFile(self.X, self.Y, self.read_rect("BASE_CUR"), self.read_rect("WATER_0"), self.read_rect("WALL_CUR"),
                              self.read_particles() if self._h.n_droplets else np.zeros((0, 5), np.float32), [], gui)

    # ---- profiling ----
    def profile(self, enable: bool):
        self._h.profile(enable)

    def profile_read(self):
        return self._h.profile_read()

    @property
    def handle(self) -> Handle:
        return self._h
