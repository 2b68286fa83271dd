/*
 * sim_host.js -- Node.js host for the MI355X engine: the reference's simulation seam in JavaScript.
 *
 * app.js cannot be require()d under Node (DOM everywhere), so this file restates, call for call, the parts
 * of it that belong to the hot path and binds them to the N-API addon instead of WebGL:
 *   loadData()            app.js:1256-1366   -> loadSave()
 *   settings merge        app.js:3378-3399 + libraries/dat.gui.min.js:136-150 -> mergeSettings()
 *   derived constants     app.js:5436-5476   -> uniformsFromGui()
 *   setGuiUniforms()      app.js:3401-3443   -> WeatherSim.pushUniforms()
 *   updateSunlight()      app.js:6510-6561   -> WeatherSim.updateSunlight()
 *   draw() simulation     app.js:5814-6005   -> WeatherSim.frame()
 *   readPixels consumers  SURVEY.md 3.5      -> WeatherSim.readRect() & friends
 *   prepareDownload()     app.js:6575-6628   -> WeatherSim.toSave()
 * Rendering / UI / audio stay where they are in the reference.
 *
 * CLI:  node host/sim_host.js <in.weathersandbox> <iterations> [out.weathersandbox] [--sun-fixed] [--dump out.bin]
 */
'use strict';
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');

const SAVE_FILE_VERSION_ID = 263574036; // app.js:345
const LEGACY_VERSION_ID = 1939327491;
const TIME_PER_ITERATION = 0.00008; // app.js:449
const DEG2RAD = 0.0174533, RAD2DEG = 57.2957795; // app.js:340-341

const GUI_DEFAULTS = { // app.js:347-407
  vorticity: 0.005, dragMultiplier: 0.001, wind: 0.0, globalEffectsStartAlt: 0, globalEffectsEndAlt: 10000, globalDrying: 0.0,
  globalHeating: 0.0, soundingForcing: 0.0, sunIntensity: 1.0, waterTemperature: 25.0, dynamicWaterTemperature: true,
  landEvaporation: 0.00005, waterEvaporation: 0.0001, evapHeat: 2.90, meltingHeat: 0.43, condensationRate: 0.0050, waterWeight: 0.25,
  inactiveDroplets: 0, aboveZeroThreshold: 1.0, subZeroThreshold: 0.005, spawnChance: 0.00005, snowDensity: 0.2, fallSpeed: 0.0003,
  growthRate0C: 0.0001, growthRate_30C: 0.001, freezingRate: 0.01, meltingRate: 0.01, evapRate: 0.0008, displayMode: 'DISP_REAL',
  wrapHorizontally: true, SmoothCam: true, camSpeed: 0.01, exposure: 1.0, timeOfDay: 9.9, latitude: 45.0, month: 6.65, sunAngle: 9.9,
  dayNightCycle: true, greenhouseGases: 0.001, waterGreenHouseEffect: 0.0015, IR_rate: 1.0, tool: 'TOOL_NONE', brushSize: 20,
  wholeWidth: false, intensity: 0.01, showGraph: false, realDewPoint: false, enablePrecipitation: true, showDrops: false, paused: false,
  IterPerFrame: 10, auto_IterPerFrame: true, sound: true, dryLapseRate: 10.0, simHeight: 12000, twelveHourClock: false,
  lengthUnit: 'LENGTH_UNIT_METRIC', tempUnit: 'TEMP_UNIT_C', windUnit: 'SPEED_UNIT_KMH'
};

const FIELD = {BASE_CUR: 0, BASE_DISP: 1, WATER_0: 2, WATER_CUR: 3, WALL_CUR: 4, WALL_DISP: 5, LIGHT_0: 6, LIGHT_1: 7, CURL: 8, VORT: 9,
               PRECIP_FB: 10, PRECIP_DEP: 11, LIGHTNING: 12, EMITTED: 13};

function loadAddon()
{
  return require(path.join(__dirname, 'wxsim_napi.node')); // throws if libwxsim.so / the GPU is missing
}

// ---- .weathersandbox codec (app.js:1261-1344, 6610-6621; pako 1.0.3 == RFC 1950 zlib) ----
function decodeSave(buf)
{
  const version = buf.readUInt32LE(0);
  if (version !== SAVE_FILE_VERSION_ID && version !== LEGACY_VERSION_ID) throw new Error('Incompatible file!');
  const raw = zlib.inflateSync(buf.slice(4));
  const X = raw.readUInt16LE(0), Y = raw.readUInt16LE(2);
  let off = 4;
  const n = X * Y * 4;
  function f32(count) { const a = new Float32Array(raw.buffer.slice(raw.byteOffset + off, raw.byteOffset + off + count * 4)); off += count * 4; return a; }
  const base = f32(n), water = f32(n);
  const wall = new Int8Array(raw.buffer.slice(raw.byteOffset + off, raw.byteOffset + off + n));
  off += n;
  const ndBytes = Math.floor((X * Y) / 25 * 4 * 5);
  const nd = Math.floor(ndBytes / 20);
  const droplets = f32(nd * 5);
  off += ndBytes - nd * 20;
  const stations = [];
  let settings = null;
  if (version === SAVE_FILE_VERSION_ID) {
    const ns = raw.readInt16LE(off);
    off += 2;
    for (let i = 0; i < ns; i++) stations.push([raw.readInt16LE(off + 4 * i), raw.readInt16LE(off + 4 * i + 2)]);
    off += ns * 4;
    const txt = raw.slice(off).toString('utf8');
    settings = txt.trim().length ? JSON.parse(txt) : null;
  }
  return {X: X, Y: Y, base: base, water: water, wall: wall, droplets: droplets, stations: stations, settings: settings, version: version};
}
function loadSave(file) { return decodeSave(fs.readFileSync(file)); }

function encodeSave(sf)
{
  const head = Buffer.alloc(4);
  head.writeUInt16LE(sf.X, 0);
  head.writeUInt16LE(sf.Y, 2);
  const ns = Buffer.alloc(2 + sf.stations.length * 4);
  ns.writeUInt16LE(sf.stations.length, 0);
  sf.stations.forEach(function(s, i) { ns.writeInt16LE(s[0], 2 + 4 * i); ns.writeInt16LE(s[1], 4 + 4 * i); });
  const b = function(t) { return Buffer.from(t.buffer, t.byteOffset, t.byteLength); };
  const body = Buffer.concat([head, b(sf.base), b(sf.water), b(sf.wall), b(sf.droplets), ns, Buffer.from(JSON.stringify(sf.settings || {}), 'utf8')]);
  const ver = Buffer.alloc(4);
  ver.writeUInt32LE(SAVE_FILE_VERSION_ID, 0);
  return Buffer.concat([ver, zlib.deflateSync(body)]);
}

// ---- settings merge: missing numeric (or a saved -1) -> default, missing boolean -> false ----
function mergeSettings(saved, simHeight)
{
  // new simulation (app.js:3378-3391): the defaults, with simHeight AND globalEffectsEndAlt = the start dialog's height (app.js:437, 3380-3381)
  if (saved == null) return Object.assign({}, GUI_DEFAULTS, {simHeight: simHeight || 12000, globalEffectsEndAlt: simHeight || 12000});
  const out = Object.assign({}, saved);
  Object.keys(GUI_DEFAULTS).forEach(function(k) {
    const d = GUI_DEFAULTS[k];
    if (typeof d === 'boolean') { if (!(k in out)) out[k] = false; }
    else if (typeof d === 'number') { if (!(k in out) || out[k] === -1) out[k] = d; }
    else if (!(k in out)) out[k] = d;
  });
  return out;
}

// ---- derived parameters ----
function initialTemperatureProfile(Y, simHeight, dryLapse)
{ // app.js:5467-5474 (float64 math, then Float32Array as gl.uniform4fv)
  const T = new Float32Array(Y + 1);
  for (let y = 0; y < Y + 1; y++) {
    const altitude = y / (Y + 1) * simHeight;
    const realTemp = Math.max(15.0 + (altitude - 0) * (-70.0 - 15.0) / (12000 - 0), -60);
    T[y] = (realTemp + 273.15) + (y / Y) * dryLapse;
  }
  return T;
}
// ---- new simulation: the 1-D part of the setup pass (setupShader.frag:26-92), expanded on the device by wx_setup_columns ----
function fract(x) { return x - Math.floor(x); }
function srand(n) { return fract(Math.sin(n) * 43758.5453123); }                      // setupShader.frag:26
function snoise(p) { const fl = Math.floor(p), fc = p - fl; return srand(fl) * (1.0 - fc) + srand(fl + 1.0) * fc - 0.5; } // :28-33
function terrainColumns(X, Y, gui, opts)
{
  opts = opts || {};
  const seed = opts.seed != null ? opts.seed : 0.5, heightMult = opts.heightMult != null ? opts.heightMult : 0.3;
  const snap = opts.snap != null ? opts.snap : 2; // even x / y steps (see SURVEY Appendix C); 1 = the reference's raw terrain
  const simH = gui.simHeight, dryLapse = simH * gui.dryLapseRate / 1000.0, texY = 1.0 / Y;
  const h = new Float64Array(X);
  for (let x = 0; x < X; x++) { // :44-61
    if (heightMult < 0.05) { h[x] = 0.0; continue; }
    if (heightMult < 0.10) { h[x] = 0.005; continue; }
    const v = (x + 0.5) * 0.001;
    let acc = 0.0;
    for (let i = 2.0; i < 1000.0; i *= 1.5) acc += snoise(v * i + srand(seed + i) * 10.0) * 0.5 / i;
    h[x] = acc * heightMult;
  }
  if (snap > 1) for (let x = 0; x < X; x++) h[x] = h[x - x % snap];
  const d = {wallRows: new Int32Array(X), sea: new Uint8Array(X), vegNoise: new Float64Array(X), snow: new Float32Array(X),
             T_air: new Float32Array(Y), totalWater: new Float32Array(Y), cloudWater: new Float32Array(Y)};
  for (let x = 0; x < X; x++) {
    let rows = Math.max(1, Math.ceil(h[x] * Y - 0.5)); // wall: texCoord.y < texelSize.y or texCoord.y < height (:63-64)
    if (snap > 1) rows = Math.floor((rows + snap - 1) / snap) * snap;
    d.wallRows[x] = Math.min(rows, Y - 8);
    d.sea[x] = h[x] < texY ? 1 : 0;
    d.vegNoise[x] = snoise((x + 0.5) * 0.01 + srand(seed) * 10.0) * 150.0;         // :72
    d.snow[x] = Math.min(Math.max((h[x] * simH - 2000.0) * 100.0 / 3000.0, 0.0), 100.0);
  }
  const T0 = initialTemperatureProfile(Y, simH, dryLapse);
  for (let y = 0; y < Y; y++) { // initial sounding :78-89
    const tcy = (y + 0.5) / Y, realT = T0[y] - tcy * dryLapse, dew = tcy < 0.20 ? realT - 2.0 : realT - 20.0;
    const tot = Math.pow(dew / 250.0, 17);
    d.T_air[y] = T0[y];
    d.totalWater[y] = tot;
    d.cloudWater[y] = Math.max(tot - Math.pow(realT / 250.0, 17), 0.0);
  }
  return d;
}
function initRainDrops(n, rng)
{ // initRainDrops(), app.js:4901-4913: inactive droplets whose fields are random seeds
  rng = rng || Math.random;
  const d = new Float32Array(5 * n);
  for (let i = 0; i < n; i++) {
    d[5 * i] = rng(); d[5 * i + 1] = rng(); d[5 * i + 2] = -10.0 + rng(); d[5 * i + 3] = rng(); d[5 * i + 4] = rng();
  }
  return d;
}
// realWorldSounding_T / _W / _Vel from a raw sounding: rawSoundingToSimSounding (app.js:149-186) + app.js:5444-5463.
// `raw`: samples ordered from the top of the sounding down to the ground, {alt [m], t, td [deg C], vel [km/h], angle [deg]}.
// msToRawVelocity divides by the GLOBAL cellHeight, which still holds its page-load value 12000 / 300 = 40 m (app.js:435) when mainScript
// builds these arrays -- it is assigned simHeight / sim_res_y only afterwards (app.js:5476): the default reproduces that.
function soundingArrays(raw, Y, simHeight, dryLapse, cellHeight)
{
  if (cellHeight == null) cellHeight = 12000.0 / 300.0;
  const bad = function(d) { return isNaN(d.t) || isNaN(d.td) || isNaN(d.vel); };
  const T = new Float32Array(Y + 1), W = new Float32Array(Y + 1), V = new Float32Array(Y + 1);
  let idx = raw.length - 1;
  for (let y = 0; y < Y + 1; y++) {
    const alt = y * (simHeight / Y);
    while (raw[idx].alt < alt || bad(raw[idx])) {
      idx--;
      if (idx < 0) throw new Error('sounding ends below the simulated altitude ' + alt + ' m');
    }
    const above = raw[idx], below = raw[Math.min(idx + 1, raw.length - 1)];
    let s = above;
    if (above.alt != alt && alt >= raw[raw.length - 1].alt) {
      const a = (alt - below.alt) / (above.alt - below.alt);
      s = {};
      ['t', 'td', 'vel', 'angle'].forEach(function(k) { s[k] = below[k] * (1 - a) + above[k] * a; });
    }
    const velMs = s.vel * Math.cos(s.angle * DEG2RAD) / 3.6;
    T[y] = (s.t + 273.15) + (y / Y) * dryLapse;
    W[y] = Math.pow((s.td + 273.15) / 250.0, 17);
    V[y] = velMs * 3600 / cellHeight * TIME_PER_ITERATION;
  }
  return {T: T, W: W, Vel: V};
}
function sunFromAngle(sunAngleDeg, sunIntensityGui)
{ // app.js:6538-6550
  return {
    zenith: (sunAngleDeg - 90) * DEG2RAD,
    intensity: sunIntensityGui * Math.pow(Math.max(Math.sin((180.0 - sunAngleDeg) * DEG2RAD), 0.0), 0.1) * 1300.0
  };
}
function sunAngleFromTime(timeOfDay, month, latitude)
{ // app.js:6522-6536
  const tod = (timeOfDay / 24.0) * 2.0 * Math.PI - Math.PI / 2.0;
  const tiltDeg = Math.sin(month * 0.5236 - 1.92) * 23.5;
  const t = tiltDeg * DEG2RAD, l = latitude * DEG2RAD;
  let ang = Math.asin(Math.sin(t) * Math.sin(l) + Math.cos(t) * Math.cos(l) * Math.sin(tod)) * RAD2DEG;
  if (latitude - tiltDeg < 0.0) ang = 180.0 - ang;
  return ang;
}
// startSimulation()'s clock (app.js:3902-3910) + onUpdateTimeOfDaySlider / onUpdateMonthSlider (app.js:6494-6507); the Date
// methods truncate their fractional arguments
function initialSimDateTime(month, timeOfDay, dayNightCycle)
{
  const t = new Date(2000, Math.floor(month) - 1, (month % 1) * 30.417);
  if (dayNightCycle) {
    t.setHours(timeOfDay, (timeOfDay % 1) * 60);
    const m = month - 0.96;
    t.setMonth(m, (m % 1) * 30);
  }
  return t;
}
// the clock part of updateSunlight(deltaT_hours), app.js:6513-6516
function advanceSimDateTime(t0, deltaHours)
{
  const t = new Date(t0.getTime() + deltaHours * 3600 * 1000);
  return {t: t, timeOfDay: t.getHours() + t.getMinutes() / 60. + t.getSeconds() / 3600., month: t.getMonth() + 1 + t.getDate() / 30.5 + t.getHours() / 720.};
}
function uniformsFromGui(gui, Y, opts)
{
  opts = opts || {};
  const simH = gui.simHeight;
  const dryLapse = simH * gui.dryLapseRate / 1000.0; // app.js:5439
  const sun = sunFromAngle(opts.sunAngleDeg != null ? opts.sunAngleDeg : gui.sunAngle, gui.sunIntensity);
  return {
    dragMultiplier: gui.dragMultiplier, wind: gui.wind, vorticity: gui.vorticity, landEvaporation: gui.landEvaporation,
    waterEvaporation: gui.waterEvaporation, dynamicWaterTemperature: gui.dynamicWaterTemperature ? 1.0 : 0.0, evapHeat: gui.evapHeat,
    waterWeight: gui.waterWeight, sunAngle: sun.zenith, dryLapse: dryLapse, meltingHeat: gui.meltingHeat, condensationRate: gui.condensationRate,
    globalDrying: gui.globalDrying, globalHeating: gui.globalHeating, soundingForcing: gui.soundingForcing,
    globalEffectsStartAlt: gui.globalEffectsStartAlt / simH, globalEffectsEndAlt: gui.globalEffectsEndAlt / simH,
    waterTemperature: gui.waterTemperature + 273.15, sunIntensity: sun.intensity, greenhouseGases: gui.greenhouseGases,
    waterGreenHouseEffect: gui.waterGreenHouseEffect, IR_rate: gui.IR_rate, aboveZeroThreshold: gui.aboveZeroThreshold,
    subZeroThreshold: gui.subZeroThreshold, spawnChanceMult: gui.spawnChance, snowDensity: gui.snowDensity, fallSpeed: gui.fallSpeed,
    growthRate0C: gui.growthRate0C, growthRate_30C: gui.growthRate_30C, freezingRate: gui.freezingRate, meltingRate: gui.meltingRate,
    evapRate: gui.evapRate, inactiveDroplets: 0.0, userInputValues: [0, 0, 0, 0], userInputMove: [0, 0], userInputType: -1,
    wrapHorizontally: gui.wrapHorizontally ? 1 : 0, airplaneValues: [0, 0, 0, 0], enablePrecipitation: gui.enablePrecipitation ? 1 : 0,
    quad_scale: opts.quadScale || 0, pass_mask: opts.passMask != null ? opts.passMask : 0x7F,
    initial_T: initialTemperatureProfile(Y, simH, dryLapse)
  };
}

// ---- the simulation object: mainScript() + draw()'s simulation block + readbacks ----
function WeatherSim(sf, opts)
{
  opts = opts || {};
  this.addon = opts.addon || loadAddon();
  this.X = sf.X;
  this.Y = sf.Y;
  this.gui = mergeSettings(sf.settings);
  this.nDroplets = sf.droplets ? Math.floor(sf.droplets.length / 5) : (sf.dropletSeed != null ? (sf.nDroplets | 0) : 0);
  this.opts = opts;
  this.manualSun = opts.sunFixed ? this.gui.sunAngle : null; // updateSunlight('MANUAL_ANGLE')
  this.slabs = null;
  if (opts.gpus > 1) {
    // The domain cut into opts.gpus column slabs, one per GPU (periodic in x like the textures' REPEAT wrap), each with `halo` ghost columns
    // per side; the library exchanges the halos itself (RCCL send / recv between the devices, or device-to-device copies where several
    // slabs share a device) every halo / 6 iterations, overlapped with compute: wx_group_* in include/wxsim.h. Same results as one
    // handle, bit for bit. With droplets the pool is PARTITIONED (include/wxsim.h): every slab is handed the whole pool once, an active
    // droplet is then tracked by the slab that contains it; slab widths and the halo must be multiples of the 64-column splat tile, so
    // a grid that does not allow that (or opts.particles === false) runs with the particles off and carries the pool through unchanged.
    if (this.X % opts.gpus) throw new Error('the grid width ' + this.X + ' is not divisible by ' + opts.gpus + ' slabs');
    const xo = this.X / opts.gpus, Y = this.Y, X = this.X;
    const withDrops = this.nDroplets > 0 && opts.particles !== false && this.gui.enablePrecipitation && xo % 64 == 0;
    const halo = opts.halo != null ? opts.halo : (withDrops ? 64 : 42), wl = xo + 2 * halo;
    if (withDrops && halo % 64) throw new Error('slabs with particles need a halo that is a multiple of 64 (the splat tile)');
    this.savedDroplets = sf.droplets || null; // (written back unchanged when the particles are off: the save format carries X * Y / 25 droplets)
    if (!withDrops) {
      this.nDroplets = 0;
      this.gui.enablePrecipitation = false;
    }
    this.group = this.addon.groupCreate(opts.gpus, X, Y, halo, opts.transport || 0, this.nDroplets);
    this.slabs = [];
    for (let i = 0; i < opts.gpus; i++) {
      const h = this.addon.groupSlab(this.group, i, Y), x0 = i * xo;
      const gx = new Int32Array(wl); // global column of every local one
      for (let c = 0; c < wl; c++) gx[c] = ((x0 - halo + c) % X + X) % X;
      if (sf.terrain) { // every slab generates its own columns on its own device
        const c = sf.columns, g = sf.terrain;
        this.addon.setupTerrain(h, g.seed, g.heightMult, g.snap, this.gui.simHeight, c.T_air, c.totalWater, c.cloudWater, this.nDroplets ? sf.droplets : null);
      } else if (sf.columns) {
        const c = sf.columns, pick = function(a) { const o = new a.constructor(wl); for (let k = 0; k < wl; k++) o[k] = a[gx[k]]; return o; };
        this.addon.setupColumns(h, pick(c.wallRows), pick(c.sea), pick(c.vegNoise), pick(c.snow), c.T_air, c.totalWater, c.cloudWater, this.nDroplets ? sf.droplets : null);
      } else {
        const cut = function(a) {
          const o = new a.constructor(wl * Y * 4);
          for (let y = 0; y < Y; y++)
            for (let k = 0; k < wl; k++) o.set(a.subarray((y * X + gx[k]) * 4, (y * X + gx[k]) * 4 + 4), (y * wl + k) * 4);
          return o;
        };
        this.addon.upload(h, cut(sf.base), cut(sf.water), cut(sf.wall), this.nDroplets ? sf.droplets : null);
      }
      if (sf.dropletSeed != null && this.nDroplets) this.addon.initDroplets(h, sf.dropletSeed >>> 0); // the same pool on every slab
      this.slabs.push({h: h, x0: x0, xo: xo, halo: halo, wl: wl});
    }
    this.addon.groupAgree(this.group);
    this.h = null;
  } else {
    this.h = this.addon.create(this.X, this.Y, this.nDroplets);
    if (sf.terrain) { // new simulation generated entirely on the device (wx_setup_terrain); the host keeps the per-row sounding
      const c = sf.columns, g = sf.terrain;
      this.addon.setupTerrain(this.h, g.seed, g.heightMult, g.snap, this.gui.simHeight, c.T_air, c.totalWater, c.cloudWater, this.nDroplets ? sf.droplets : null);
    } else if (sf.columns) { // ... or the textures filled on the device from 1-D descriptors computed here (setupShader.frag)
      const c = sf.columns;
      this.addon.setupColumns(this.h, c.wallRows, c.sea, c.vegNoise, c.snow, c.T_air, c.totalWater, c.cloudWater, this.nDroplets ? sf.droplets : null);
    } else {
      this.addon.upload(this.h, sf.base, sf.water, sf.wall, this.nDroplets ? sf.droplets : null);
    }
    if (sf.dropletSeed != null && this.nDroplets) this.addon.initDroplets(this.h, sf.dropletSeed >>> 0);
  }
  // startSimulation(): clock from month / timeOfDay (app.js:3902)
  const m = this.gui.month;
  this.simDateTime = initialSimDateTime(m, this.gui.timeOfDay, this.gui.dayNightCycle);
  this.brush = {userInputType: -1, userInputValues: [0, 0, 0, 0], userInputMove: [0, 0]};
  this.inactivePushed = false;
  this.pushUniforms();
}
WeatherSim.prototype.pushUniforms = function() {
  const u = uniformsFromGui(this.gui, this.Y, {sunAngleDeg: this.manualSun, quadScale: this.opts.quadScale, passMask: this.opts.passMask});
  Object.assign(u, this.brush);
  u.inactiveDroplets = this.inactivePushed ? -1.0 : 0.0; // keep the engine's 600-iteration measurement (app.js:5957-5966)
  const hs = this.slabs ? this.slabs.map(function(sl) { return sl.h; }) : [this.h];
  for (const h of hs) {
    if (this.sounding) this.addon.setParams(h, u, u.initial_T, this.sounding.T, this.sounding.W, this.sounding.Vel);
    else this.addon.setParams(h, u, u.initial_T);
  }
  this.inactivePushed = true;
};
WeatherSim.prototype.setGui = function(changes) { Object.assign(this.gui, changes); this.pushUniforms(); };
// load a real sounding for the `soundingForcing` slider (app.js:5444-5463); raw = scraper output, top of the sounding first
WeatherSim.prototype.setSounding = function(raw) {
  this.sounding = soundingArrays(raw, this.Y, this.gui.simHeight, this.gui.simHeight * this.gui.dryLapseRate / 1000.0);
  this.pushUniforms();
};
WeatherSim.prototype.setBrush = function(type, x, y, intensity, brushSize, move) { // app.js:5749-5808
  this.brush = {userInputType: type, userInputValues: [x, y, intensity, brushSize * 0.5], userInputMove: move || [0, 0]};
  this.pushUniforms();
};
WeatherSim.prototype.updateSunlight = function(deltaHours) { // app.js:6510-6561
  if (deltaHours != null) {
    const c = advanceSimDateTime(this.simDateTime, deltaHours);
    this.simDateTime = c.t;
    this.gui.timeOfDay = c.timeOfDay;
    this.gui.month = c.month;
  }
  this.gui.sunAngle = sunAngleFromTime(this.gui.timeOfDay, this.gui.month, this.gui.latitude);
  this.manualSun = null;
  this.pushUniforms();
};
// one animation frame: `IterPerFrame` iterations (app.js:5814-6005)
WeatherSim.prototype.frame = function(nIter) {
  const n = nIter != null ? nIter : this.gui.IterPerFrame;
  if (this.gui.paused) return;
  if (this.gui.dayNightCycle && this.manualSun == null) this.updateSunlight(TIME_PER_ITERATION * n);
  if (this.slabs) this.addon.groupStep(this.group, n); // every slab: n iterations + the halo exchanges that fall into them
  else this.addon.step(this.h, n);
  if (!this.placementTold && !this.slabs) { // the engine looked for a fast placement of its planes inside the first step of a big grid: say so once
    this.placementTold = true;
    const pi = this.addon.placementInfo(this.h);
    if (pi && this.verbose !== false) console.error(`[wxsim] placement search: ${pi[0].toFixed(4)} ms / iteration on the first allocations, ${pi[1].toFixed(4)} kept`);
  }
};
WeatherSim.prototype.placementInfo = function() { return this.slabs ? null : this.addon.placementInfo(this.h); };
WeatherSim.prototype.sync = function() { if (this.slabs) this.addon.groupSync(this.group); else this.addon.sync(this.h); };
// engine options with no counterpart in app.js: deterministic particle splats (option 1), per-launch checks (option 2), and the
// search for a fast placement of the handle's planes in device memory (returns [ms before, ms after]; the state is unchanged)
WeatherSim.prototype.setOption = function(option, value) { if (this.slabs) this.addon.groupSetOption(this.group, option, value); else this.addon.setOption(this.h, option, value); };
WeatherSim.prototype.tunePlacement = function(tries, itersPerTry) {
  if (this.slabs) throw new Error('tunePlacement: single-handle simulations only');
  return this.addon.tunePlacement(this.h, tries || 6, itersPerTry || 30);
};
WeatherSim.prototype.iterNum = function() { return this.addon.getIter(this.slabs ? this.slabs[0].h : this.h); };
WeatherSim.prototype.readRect = function(field, x, y, w, h, Type) {
  const id = typeof field === 'string' ? FIELD[field] : field;
  const ch = id === FIELD.CURL ? 1 : (id === FIELD.VORT || id === FIELD.PRECIP_DEP) ? 2 : 4;
  const isWall = id === FIELD.WALL_CUR || id === FIELD.WALL_DISP;
  const T = Type || (isWall ? Int8Array : Float32Array), dst = new T(w * h * ch);
  if (!this.slabs) return this.addon.readRect(this.h, id, x, y, w, h, dst);
  if (x < 0 || x + w > this.X) throw new RangeError('readRect: columns outside the grid (no wrap, as readPixels)');
  for (const sl of this.slabs) { // the owned columns of every slab the rectangle touches
    const a = Math.max(x, sl.x0), b = Math.min(x + w, sl.x0 + sl.xo);
    if (a >= b) continue;
    const part = this.addon.readRect(sl.h, id, a - sl.x0 + sl.halo, y, b - a, h, new T((b - a) * h * ch));
    for (let r = 0; r < h; r++) dst.set(part.subarray(r * (b - a) * ch, (r + 1) * (b - a) * ch), (r * w + (a - x)) * ch);
  }
  return dst;
};
WeatherSim.prototype.measureStation = function(x, y) { // Weatherstation.measure, app.js:1084-1092
  return {base: this.readRect('BASE_CUR', x, y - 1, 1, 3), water: this.readRect('WATER_0', x, y - 1, 1, 2)};
};
WeatherSim.prototype.soundingColumn = function(x) { // soundingGraph.draw, app.js:3931-3943 (wall as Int32Array)
  return {base: this.readRect('BASE_DISP', x, 0, 1, this.Y), water: this.readRect('WATER_CUR', x, 0, 1, this.Y),
          wall: this.readRect('WALL_DISP', x, 0, 1, this.Y, Int32Array)};
};
WeatherSim.prototype.readParticles = function() {
  if (!this.nDroplets) return new Float32Array(0);
  if (!this.slabs) return this.addon.readParticles(this.h, 0, this.nDroplets, new Float32Array(this.nDroplets * 5));
  // the partitioned pool: right after an exchange every active droplet is owned (flag 2) by exactly one slab, whose record is THE
  // record; an inactive droplet's record is the same on every slab
  this.addon.groupExchange(this.group);
  const n = this.nDroplets, out = this.addon.readParticles(this.slabs[0].h, 0, n, new Float32Array(n * 5));
  for (const sl of this.slabs) {
    const d = this.addon.readParticles(sl.h, 0, n, new Float32Array(n * 5)), f = this.addon.poolFlags(sl.h, new Uint8Array(n));
    for (let i = 0; i < n; i++) if (f[i] == 2) out.set(d.subarray(5 * i, 5 * i + 5), 5 * i);
  }
  return out;
};
WeatherSim.prototype.toSave = function() { // prepareDownload(): FB0 = base_0, water_0, wall_0 (app.js:6584-6593)
  return {X: this.X, Y: this.Y, base: this.readRect('BASE_CUR', 0, 0, this.X, this.Y), water: this.readRect('WATER_0', 0, 0, this.X, this.Y),
          wall: this.readRect('WALL_CUR', 0, 0, this.X, this.Y), droplets: this.slabs && !this.nDroplets && this.savedDroplets ? this.savedDroplets : this.readParticles(),
          stations: [], settings: this.gui};
};
// display fields of a viewport, copied asynchronously into pinned memory (what the renderer binds, app.js:6081-6219):
// returns {wait()} whose result holds typed-array views BASE_DISP, WATER_CUR, WALL_DISP, LIGHT_0, CURL, PRECIP_FB
WeatherSim.prototype.streamFrame = function(x, y, w, h) {
  if (this.slabs) throw new Error('streamFrame: single-handle simulations only (read the slabs with readRect)');
  const self = this, ab = this.addon.streamFrame(this.h, x, y, w, h), n = w * h;
  return {wait: function() {
    self.addon.streamWait(self.h);
    let off = 0;
    const f32 = function(ch) { const v = new Float32Array(ab, off, n * ch); off += 4 * n * ch; return v; };
    const out = {BASE_DISP: f32(4), WATER_CUR: f32(4)};
    out.WALL_DISP = new Int8Array(ab, off, 4 * n);
    off += 4 * n;
    out.LIGHT_0 = f32(4);
    out.CURL = f32(1);
    out.PRECIP_FB = f32(4);
    out.EMITTED = new Uint16Array(ab, off, 4 * n); // binary16 bits, as texImage2D(..., gl.RGBA16F, gl.RGBA, gl.HALF_FLOAT, u16) takes them
    return out;
  }};
};
WeatherSim.prototype.destroy = function() { if (this.slabs) { this.addon.groupDestroy(this.group); this.slabs = []; } else this.addon.destroy(this.h); };
// startup without a save file: what mainScript() does for a new simulation (setup pass + initRainDrops)
WeatherSim.newSimulation = function(X, Y, opts) {
  opts = opts || {};
  const gui = mergeSettings(opts.settings || null);
  const nDrops = opts.nDroplets != null ? opts.nDroplets : Math.floor(X * Y / 25); // app.js: one droplet per 25 cells
  // opts.dropletSeed (an integer): the droplet pool is generated on the device as well (wx_init_droplets), identically on every slab
  // opts.deviceTerrain: the terrain noise runs on the device too (wx_setup_terrain); the descriptors computed here then only serve as reference
  const terrain = opts.deviceTerrain ? {seed: opts.seed != null ? opts.seed : 0.5, heightMult: opts.heightMult != null ? opts.heightMult : 0.3, snap: opts.snap != null ? opts.snap : 2} : null;
  return new WeatherSim({X: X, Y: Y, settings: gui, columns: terrainColumns(X, Y, gui, opts), terrain: terrain, nDroplets: nDrops, dropletSeed: opts.dropletSeed != null ? opts.dropletSeed : null,
                         droplets: nDrops && opts.dropletSeed == null ? initRainDrops(nDrops, opts.rng) : null}, opts);
};

module.exports = {WeatherSim: WeatherSim, initialSimDateTime: initialSimDateTime, advanceSimDateTime: advanceSimDateTime, loadSave: loadSave, decodeSave: decodeSave, encodeSave: encodeSave, mergeSettings: mergeSettings,
                  uniformsFromGui: uniformsFromGui, initialTemperatureProfile: initialTemperatureProfile, sunFromAngle: sunFromAngle,
                  sunAngleFromTime: sunAngleFromTime, terrainColumns: terrainColumns, initRainDrops: initRainDrops, soundingArrays: soundingArrays, GUI_DEFAULTS: GUI_DEFAULTS,
                  FIELD: FIELD};

if (require.main === module) {
  // flags anywhere on the line; the value-taking ones as `--gpus 4` or `--gpus=4` (their values are NOT positional arguments)
  const VALUED = {'--gpus': 1, '--halo': 1, '--transport': 1};
  const flags = [], pos = [], opt = {};
  for (let i = 2; i < process.argv.length; i++) {
    const a = process.argv[i];
    if (!a.startsWith('--')) { pos.push(a); continue; }
    const eq = a.indexOf('=');
    const name = eq >= 0 ? a.slice(0, eq) : a;
    if (VALUED[name]) {
      const v = eq >= 0 ? a.slice(eq + 1) : process.argv[++i];
      if (v === undefined || v === '') { console.error(name + ' needs a value'); process.exit(2); }
      opt[name] = v;
    } else flags.push(name);
  }
  if (pos.length < 2) {
    console.error('usage: node sim_host.js <in.weathersandbox> <iterations> [out.weathersandbox] [--sun-fixed] [--splat-order] [--gpus N [--halo H] [--transport rccl|local] [--exact]]');
    process.exit(2);
  }
  const sf = loadSave(pos[0]);
  const num = function(flag, dflt) {
    if (opt[flag] === undefined) return dflt;
    const v = parseInt(opt[flag], 10);
    if (!(v === v)) { console.error(flag + ': not a number: ' + opt[flag]); process.exit(2); }
    return v;
  };
  const tr = opt['--transport'] !== undefined ? opt['--transport'] : 'auto';
  const sim = new WeatherSim(sf, {sunFixed: flags.indexOf('--sun-fixed') >= 0, gpus: num('--gpus', 1), halo: opt['--halo'] !== undefined ? num('--halo', 42) : null,
                                  transport: tr == 'rccl' ? 1 : tr == 'local' ? 2 : 0});
  if (flags.indexOf('--splat-order') >= 0) sim.setOption(1, 1); // WX_OPT_SPLAT_ORDER: deterministic particle splats
  if (flags.indexOf('--exact') >= 0 && sim.slabs && sim.nDroplets) sim.setOption(7, 1); // WX_OPT_POOL_EXACT: slabs with particles == one handle, exactly
  const n = parseInt(pos[1], 10);
  const t0 = Date.now();
  let left = n;
  while (left > 0) { const k = Math.min(left, sim.gui.IterPerFrame); sim.frame(k); left -= k; }
  sim.sync();
  const dt = (Date.now() - t0) / 1000;
  console.log(JSON.stringify({X: sf.X, Y: sf.Y, iterations: n, iterNum: sim.iterNum(), seconds: dt, McellStepsPerS: sf.X * sf.Y * n / dt / 1e6}));
  if (pos[2]) fs.writeFileSync(pos[2], encodeSave(sim.toSave()));
  sim.destroy();
}
