#!/usr/bin/env node
/*
 * Golden vectors for the HOST side of the simulation step (SURVEY 8a rows a11 / a12) -- TEST INFRASTRUCTURE, container only.
 *
 * app.js cannot be require()d (DOM everywhere, syntax newer than this Node), but the code that decides the simulation's uniforms and
 * the order of its draws is plain JavaScript.  This script reads /root/reference/app.js AT RUN TIME, cuts out the pieces listed in
 * PIECES below by their own text (anchors, not line numbers; the line ranges found are written into the output), and EXECUTES them
 * against a recording mock of the WebGL2 context (mockgl.js) and a stub guiControls:
 *
 *   constants / helpers             degToRad, timePerIteration, map_range, CtoK, maxWater, realToPotentialT, rawSoundingToSimSounding ...
 *   guiControls_default             app.js:347-407
 *   shader -> program table         the `await loadShader(..)` / `createProgram(..)` lines, app.js:4701-4757, 4879-4881
 *   full-screen quad                app.js:4769-4818          particle buffers / VAOs / TFs    app.js:4885-5002
 *   textures + framebuffers         app.js:5148-5317          derived parameters + constant uniforms  app.js:5436-5636
 *   setGuiUniforms                  app.js:3401-3443          updateSunlight + clock start     app.js:6494-6566, 3901-3910
 *   brush uniform block             app.js:5748-5810          the iteration loop               app.js:5830-6005
 *
 * Output (DATA only -- numbers, enum names, object names; no source text):
 *   tests/golden/host_uniforms.json    every uniform value each simulation program receives, initial_T, sounding arrays, sun state,
 *                                      for settings x grid heights x day / night
 *   tests/golden/host_call_trace.json  texture / framebuffer / VAO tables and the GL call sequence of whole iterations (even / odd,
 *                                      precipitation on / off, iterNum crossing 600, sound on)
 * The tests then hold params.py / host/sim_host.js against the first and oracle/golden/harness.js (run on the same mock) against the
 * second.  Usage:  TZ=UTC node oracle/golden/gen_host_golden.js [--out DIR] [--save-settings FILE.json]
 */
'use strict';
const fs = require('fs');
const path = require('path');
const mockgl = require('./mockgl.js');

const REF = process.env.WX_REFERENCE || '/root/reference';
const SRC = fs.readFileSync(path.join(REF, 'app.js'), 'utf8');
const L = SRC.split('\n');
const CITES = {};

// ---------------------------------------------------------------- slicing by anchors
function lineOf(marker, from, optional)
{
  for (let i = from || 0; i < L.length; i++)
    if (L[i].indexOf(marker) >= 0) return i;
  if (optional) return -1;
  throw new Error('anchor not found in app.js: ' + marker);
}
function cite(key, a, b) { CITES[key] = 'app.js:' + (a + 1) + '-' + (b + 1); }
function span(key, startMarker, endMarker, from)
{
  const a = lineOf(startMarker, from), b = lineOf(endMarker, a);
  cite(key, a, b);
  return L.slice(a, b + 1).join('\n');
}
// text from line `a` up to the line on which the first `{` opened at or after it closes (skips strings, template literals, comments)
function block(key, a)
{
  let depth = 0, paren = 0, seen = false, mode = null; // mode: ', ", `, '/*'; braces inside a parameter list (destructuring defaults) do not count
  for (let i = a; i < L.length; i++) {
    const s = L[i];
    for (let j = 0; j < s.length; j++) {
      const c = s[j], d = s[j + 1];
      if (mode == '/*') {
        if (c == '*' && d == '/') { mode = null; j++; }
        continue;
      }
      if (mode) {
        if (c == '\\') j++;
        else if (c == mode) mode = null;
        continue;
      }
      if (c == '/' && d == '/') break;
      if (c == '/' && d == '*') { mode = '/*'; j++; continue; }
      if (c == '\'' || c == '"' || c == '`') { mode = c; continue; }
      if (c == '(' && depth == 0) paren++;
      else if (c == ')' && depth == 0) paren--;
      else if (paren > 0) continue;
      else if (c == '{') { depth++; seen = true; }
      else if (c == '}') depth--;
    }
    if (mode == '\'' || mode == '"') mode = null; // (no multi-line quotes)
    if (seen && depth == 0) {
      cite(key, a, i);
      return L.slice(a, i + 1).join('\n');
    }
  }
  throw new Error('unbalanced block at app.js:' + (a + 1));
}
function func(name, nth)
{
  const re = new RegExp('^\\s*(async\\s+)?function ' + name + '\\(');
  let seen = 0;
  for (let i = 0; i < L.length; i++)
    if (re.test(L[i]) && seen++ == (nth || 0)) return block('function ' + name, i);
  throw new Error('function not found in app.js: ' + name);
}
function constLine(name)
{
  const re = new RegExp('^(const|var|let) ' + name + ' = [^;]*;');
  for (let i = 0; i < L.length; i++)
    if (re.test(L[i])) { cite('const ' + name, i, i); return re.exec(L[i])[0]; }
  throw new Error('constant not found in app.js: ' + name);
}
function matching(key, re)
{
  const out = [];
  let a = -1, b = -1;
  for (let i = 0; i < L.length; i++)
    if (re.test(L[i])) { out.push(L[i]); if (a < 0) a = i; b = i; }
  if (!out.length) throw new Error('no line matches ' + re);
  cite(key, a, b);
  return out.join('\n');
}

const P = {};
P.consts = ['degToRad', 'radToDeg', 'timePerIteration', 'wf_devider', 'wf_pow', 'cellHeight', 'minShadowLight', 'sunIsUp', 'sim_height'].map(constLine).join('\n');
P.helpers = ['clamp', 'mod', 'map_range', 'map_range_C', 'CtoK', 'maxWater', 'realToPotentialT', 'sampleIsInvalid', 'mixGeneric', 'msToRawVelocity',
             'rawSoundingToSimSounding'].map(function(n) { return func(n); }).join('\n');
// Node 12 has no `??`: the one helper that uses it (mixGeneric) is down-levelled mechanically, a ?? b -> (a != null ? a : b)
P.helpers = P.helpers.replace(/(\w+\[\w+\]) \?\? (\w+)/g, '($1 != null ? $1 : $2)');
P.defaults = block('guiControls_default', lineOf('const guiControls_default = {'));
P.shaders = matching('loadShader lines', /^\s*const \w+ = await loadShader\('[^']+'\);/);
P.programs = matching('createProgram lines', /^\s*const \w+Program = createProgram\(/);
P.fboClass = block('class FBO', lineOf('class FBO // wraps texture'));
{
  const i = lineOf('emittedLightFBO = new FBO(sim_res_x, sim_res_y');
  cite('emittedLightFBO', i, i);
  P.emitted = L[i];
}
P.quad = span('quad', 'const fluidQuadVertices = [', 'gl.bindBuffer(gl.ARRAY_BUFFER, null);');
{
  const a = lineOf('const dropPositionAttribLocation = 0;'), f = lineOf('function setupPrecipitationBuffers()', a);
  P.particles = L.slice(a, f).join('\n') + '\n' + block('particle buffers', f);
  cite('particle buffers', a, parseInt(CITES['particle buffers'].split('-')[1]) - 1);
}
P.textures = span('textures + framebuffers', 'const baseTexture_0 = gl.createTexture();', 'gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, lightningDataTexture, 0);');
P.derived = span('derived parameters + constant uniforms', 'var texelSizeX = 1.0 / sim_res_x;', 'setGuiUniforms(); // all uniforms changed by gui');
P.nested = ['setGuiUniforms', 'updateSunlight', 'onUpdateTimeOfDaySlider', 'onUpdateMonthSlider'].map(function(n) { return func(n); }).join('\n');
{
  const a = lineOf('simDateTime = new Date(2000, Math.floor(guiControls.month) - 1');
  const b = lineOf('updateSunlight(\'MANUAL_ANGLE\'); // set angle from savefile', a) + 1;
  cite('clock start', a, b);
  P.clock = L.slice(a, b + 1).join('\n');
}
P.newSimDefaults = span('new-simulation settings', 'guiControls.simHeight = sim_height;', 'guiControls.globalEffectsEndAlt = sim_height;');
P.minusOne = block('settings -1 rule', lineOf('for (const [key, value] of Object.entries(guiControls)) {'));
P.loopVars = span('loop variables', 'var srcVAO;', 'var uniformLocation_boundaryProgram_iterNum =');
P.input = span('brush uniform block', 'var inputType = -1;', 'gl.uniform1i(gl.getUniformLocation(advectionProgram, \'userInputType\'), inputType);');
{
  const a = lineOf('if (!guiControls.paused) { // Simulation part');
  const d = lineOf('if (guiControls.dayNightCycle) {', a);
  const head = block('frame head', d);
  const v = lineOf('gl.viewport(0, 0, sim_res_x, sim_res_y);', d), c = lineOf('gl.clearColor(0.0, 0.0, 0.0, 0.0);', v);
  P.frameHead = head + '\n' + L.slice(v, c + 1).join('\n');
  cite('frame head', d, c);
  P.loop = block('iteration loop', lineOf('// Simulation loop', a));
}

// identifiers the setup slices declare (GL objects get their reference names as trace tokens)
function declared(text, re)
{
  const out = [];
  let m;
  while ((m = re.exec(text)) !== null) out.push(m[1]);
  return out;
}
const OBJ_NAMES = declared(P.textures + '\n' + P.particles + '\n' + P.quad, /(?:const|var|let)?\s*(\w+) = gl\.create(?:Texture|Framebuffer|VertexArray|Buffer|TransformFeedback)\(\)/g);
const PROGRAM_NAMES = declared(P.programs, /const (\w+) = createProgram/g);

// uniform names a program's shaders declare (the reference's GLSL files, read at run time; `#include "common.glsl"` resolved like
// app.js:6668-6690 does): gl.getUniformLocation answers null for anything else and the push is a no-op
function declaredUniforms(vert, frag)
{
  const names = new Set();
  const common = fs.readFileSync(path.join(REF, 'shaders', 'common.glsl'), 'utf8');
  for (const f of [path.join('vertex', vert), path.join('fragment', frag)]) {
    const src = fs.readFileSync(path.join(REF, 'shaders', f), 'utf8').replace('#include "common.glsl"', common);
    const re = /^\s*uniform\s+(?:highp\s+|mediump\s+|lowp\s+)?\w+\s+(\w+)/gm;
    let m;
    while ((m = re.exec(src)) !== null) names.add(m[1]);
  }
  return names;
}

const AsyncFunction = Object.getPrototypeOf(async function() {}).constructor;
const BODY = [
  'var sim_res_x = __sc.X, sim_res_y = __sc.Y;',
  'var dryLapse, simDateTime, iterNum = __sc.iter0 || 0, frameBuff_0, lightFrameBuff_0;',
  'var NUM_DROPLETS = __sc.n_drops || 0, airplaneMode = false, displayWeatherStations = !!__sc.weather_stations, weatherStations = [], frameNum = 0;',
  'var soundingData = __sc.sounding || undefined, soundingDataIndex, guiControls, lastSaveTime, startDate = null, startLatitude = null;',
  'var leftMousePressed = false, ctrlPressed = false, mouseXinSim = 0, mouseYinSim = 0, prevMouseXinSim = 0, prevMouseYinSim = 0;',
  'var clockEl = {innerHTML: ""}, dateTimeStr = function() { return ""; }, console = {log: function() {}, time: function() {}, timeEnd: function() {}};',
  'var soundSystem = {soundThunder: function() { __out.thunder = (__out.thunder || 0) + 1; }};',
  P.consts, P.helpers, P.defaults,
  '__out.gui_default = JSON.parse(JSON.stringify(guiControls_default));',
  // setupDatGui(str): guiControls = JSON.parse(str); guiControls.tool = "TOOL_NONE" (app.js:3448-3450; the dat.GUI widgets themselves are
  // not run: a control missing from the saved string is created by dat.gui as -1 / false, which __sc.saved_json already contains)
  'if (__sc.saved_json == null) { guiControls = JSON.parse(JSON.stringify(guiControls_default)); guiControls.tool = "TOOL_NONE";',
  P.newSimDefaults,
  '} else { guiControls = JSON.parse(__sc.saved_json); guiControls.tool = "TOOL_NONE";',
  P.minusOne,
  '}',
  'if (__sc.gui_edit) Object.assign(guiControls, __sc.gui_edit);',
  'const loadShader = async function(n) { return n; };',
  'const createProgram = function(v, f, tf) { const p = gl.createProgram(); p.__name = v + "+" + f; p.tfVaryings = tf || null; p.declared = __declared(v, f); return p; };',
  P.shaders, P.programs,
  'const initialBaseTex = {__data: "initialBaseTex"}, initialWaterTex = {__data: "initialWaterTex"}, initialWallTex = {__data: "initialWallTex"};',
  'let emittedLightFBO;',
  P.fboClass,
  'function createAmbientLightFBOs() {', P.emitted, 'emittedLightFBO.texture.__name = "emittedLightFBO.texture"; emittedLightFBO.frameBuffer.__name = "emittedLightFBO.frameBuffer"; }',
  P.quad,
  '__out.quad = Array.from(new Float32Array(fluidQuadVertices));',
  P.particles,
  'rainDrops = __sc.rain_drops || [0, 0, -9.5, 0.5, 0.5];',
  'setupPrecipitationBuffers();',
  P.textures,
  '__names({' + OBJ_NAMES.join(', ') + '});',
  '__out.setup_ops = gl.__state.trace.length;',
  P.nested,
  P.derived,
  '__out.derived = {dryLapse: dryLapse, cellHeight: cellHeight, texelSize: [texelSizeX, texelSizeY], initial_T: Array.from(initial_T),',
  '  realWorldSounding_T: Array.from(realWorldSounding_T), realWorldSounding_W: Array.from(realWorldSounding_W), realWorldSounding_Vel: Array.from(realWorldSounding_Vel)};',
  P.clock,
  '__snapshot("after setup");',
  P.loopVars,
  'function __input() {', P.input, '}',
  'function __frameHead() {', P.frameHead, '}',
  'function __iterations(numIterations) {', P.loop, '}',
  'for (const fr of (__sc.frames || [])) {',
  '  if (fr.gui_edit) { Object.assign(guiControls, fr.gui_edit); setGuiUniforms(); }',
  '  if (fr.sun == "MANUAL_ANGLE") updateSunlight("MANUAL_ANGLE");',
  '  if (fr.mouse) { leftMousePressed = !!fr.mouse.pressed; ctrlPressed = !!fr.mouse.ctrl; prevMouseXinSim = fr.mouse.px; prevMouseYinSim = fr.mouse.py; mouseXinSim = fr.mouse.x; mouseYinSim = fr.mouse.y; }',
  '  const t0 = gl.__state.trace.length, it0 = iterNum;',
  '  if (fr.input) { gl.useProgram(advectionProgram); __input(); }',
  '  if (fr.head) __frameHead();',
  '  const t1 = gl.__state.trace.length;',
  '  if (fr.iterations) __iterations(fr.iterations);',
  '  __frame(fr, t0, t1, it0, iterNum, even);',
  '}',
  '__out.gui_final = JSON.parse(JSON.stringify(guiControls));',
  '__out.sun_state = {sunIsUp: sunIsUp, minShadowLight: minShadowLight, simDateTime_ms: simDateTime ? simDateTime.getTime() : null};',
].join('\n');

if (process.env.WX_DUMP_BODY) { fs.writeFileSync(process.env.WX_DUMP_BODY, BODY); }
async function runScenario(sc)
{
  const gl = mockgl.create({
    attribLocations: {vertPosition: 0, vertTexCoord: 1},
    onReadPixels: function(dst, st) { if (dst && dst.length >= 4) { dst[0] = sc.readback0 === undefined ? 777 : sc.readback0; dst[2] = sc.readback2 || 0; dst[3] = 0.5; } },
  });
  const out = {snapshots: {}, frames: []};
  const names = function(o) { for (const k of Object.keys(o)) if (o[k] && o[k].__id) o[k].__name = k; };
  const uniformsNow = function() { return JSON.parse(JSON.stringify(mockgl.uniforms(gl))); };
  const snapshot = function(key) { out.snapshots[key] = uniformsNow(); };
  const frame = function(fr, t0, t1, it0, it1, even) {
    out.frames.push({spec: fr, head_trace: mockgl.trace(gl, t0, t1), trace: mockgl.trace(gl, t1), iter_before: it0, iter_after: it1, even_after: even,
                     uniforms_after: uniformsNow()});
  };
  const f = new AsyncFunction('gl', '__sc', '__out', '__names', '__snapshot', '__frame', '__declared', BODY);
  await f(gl, sc, out, names, snapshot, frame, declaredUniforms);
  out.setup_trace = mockgl.trace(gl, 0, out.setup_ops);
  out.tables = mockgl.summary(gl);
  out.errors = gl.__state.errors;
  out.ignored = gl.__state.ignored.slice().sort();
  return out;
}

// ---------------------------------------------------------------- scenarios
function argOf(flag, dflt)
{
  const i = process.argv.indexOf(flag);
  return i >= 0 ? process.argv[i + 1] : dflt;
}

async function main()
{
  if ((process.env.TZ || '') != 'UTC') throw new Error('run with TZ=UTC (simDateTime uses local-time accessors)');
  const outDir = argOf('--out', path.join(__dirname, '..', '..', 'tests', 'golden'));
  const saveSettingsFile = argOf('--save-settings', null); // the settings JSON of the reference's save, written by gen_host_golden.py
  const probe = await runScenario({X: 8, Y: 8});
  const D = probe.gui_default;

  // what dat.gui makes of a saved settings string: a numeric control missing from it is created as -1, a boolean as false
  // (libraries/dat.gui.min.js, read; the -1 -> default rule itself, app.js:3394-3398, is executed)
  function asDatGuiLoads(saved)
  {
    const g = JSON.parse(JSON.stringify(saved));
    for (const k of Object.keys(D))
      if (!(k in g)) g[k] = typeof D[k] == 'number' ? -1 : typeof D[k] == 'boolean' ? false : D[k];
    return JSON.stringify(g);
  }
  const perturbed = {};
  let k = 0;
  for (const key of Object.keys(D)) {
    const v = D[key];
    if (typeof v == 'number') perturbed[key] = v == 0 ? 0.0003 * (1 + k) : v * (1.07 + 0.013 * k);
    else if (typeof v == 'boolean') perturbed[key] = !v;
    else perturbed[key] = v;
    k++;
  }
  Object.assign(perturbed, {simHeight: 9000, dryLapseRate: 9.8, globalEffectsStartAlt: 1500, globalEffectsEndAlt: 7000, waterTemperature: 18.5, latitude: -52.5,
                            month: 12.4, timeOfDay: 15.25, sunAngle: 133.7, IterPerFrame: 7, dayNightCycle: false, enablePrecipitation: true, wrapHorizontally: true});
  const settings = {'default': null, 'perturbed': JSON.stringify(perturbed)};
  if (saveSettingsFile) settings.save100 = asDatGuiLoads(JSON.parse(fs.readFileSync(saveSettingsFile, 'utf8')));

  const sounding = []; // top of the sounding first, as the scraper delivers it; one invalid sample
  for (let i = 0; i < 40; i++) {
    const alt = 16000 - i * 410;
    sounding.push({alt: alt, t: 24.5 - alt * 0.0068 + Math.sin(i) * 1.5, td: 17.0 - alt * 0.0081 + Math.cos(i * 1.7) * 2.0, vel: 12 + 0.004 * alt, angle: 200 + 3.3 * i});
  }
  sounding[17].td = NaN;
  sounding[39].alt = 0;

  const U = {meta: {generated_by: 'oracle/golden/gen_host_golden.js', reference_slices: null, tz: 'UTC', pushes_to_undeclared_uniforms_ignored_by_gl: probe.ignored}, gui_default: D, scenarios: []};
  for (const sname of Object.keys(settings))
    for (const Y of [100, 300, 500])
      for (const sun of ['manual_day', 'manual_night', 'clock', 'clock_south']) {
        const sc = {X: 2 * Y, Y: Y, saved_json: settings[sname], gui_edit: {}, frames: []};
        if (sun == 'manual_day') sc.gui_edit = {dayNightCycle: false};
        if (sun == 'manual_night') sc.gui_edit = {dayNightCycle: false, sunAngle: -20.5};
        if (sun == 'clock') sc.gui_edit = {dayNightCycle: true};
        if (sun == 'clock_south') sc.gui_edit = {dayNightCycle: true, latitude: -61.0, month: 12.6, timeOfDay: 13.37};
        if (sname == 'perturbed' && Y == 300) sc.sounding = sounding;
        // three frames of the day / night driver (no iterations: only updateSunlight(timePerIteration * IterPerFrame) of the frame head)
        if (sun.indexOf('clock') == 0) sc.frames = [{head: true}, {head: true}, {head: true}];
        const r = await runScenario(sc);
        if (r.errors.length) throw new Error(sname + ': ' + r.errors.join('; '));
        U.scenarios.push({name: sname + '/Y' + Y + '/' + sun, X: sc.X, Y: Y, saved_json: settings[sname], gui_edit: sc.gui_edit, sounding: sc.sounding || null,
                          quad: r.quad, uniforms: dedupe(r.snapshots['after setup'], r.derived), derived: trim(r.derived, Y),
                          clock_frames: r.frames.map(function(f) { return {uniforms_after: pick(f.uniforms_after), head_trace: f.head_trace}; }),
                          gui_final: r.gui_final, sun_state: r.sun_state});
      }
  // brush uniform block: idle mouse (userInputType -1) and a pressed tool
  {
    const sc = {X: 200, Y: 100, saved_json: null, gui_edit: {dayNightCycle: false, tool: 'TOOL_WALL_FIRE', brushSize: 33, intensity: 0.02, wholeWidth: false},
                frames: [{input: true, mouse: {pressed: false, x: 0.4, y: 0.3, px: 0.39, py: 0.31}},
                         {input: true, mouse: {pressed: true, x: 1.25, y: 0.3, px: 1.2, py: 0.35}},
                         {input: true, mouse: {pressed: true, ctrl: true, x: -0.25, y: 0.6, px: -0.2, py: 0.6}, gui_edit: {wrapHorizontally: false}},
                         {input: true, mouse: {pressed: true, x: 0.5, y: 0.5, px: 0.5, py: 0.5}, gui_edit: {wholeWidth: true, tool: 'TOOL_TEMPERATURE'}}]};
    const r = await runScenario(sc);
    U.brush = r.frames.map(function(f) { return {mouse: f.spec.mouse, gui_edit: f.spec.gui_edit || null, trace: f.head_trace}; });
  }
  // the four per-row arrays reach several programs: store them once (derived) and name them where they are pushed
  function dedupe(u, derived)
  {
    for (const p of Object.keys(u)) // the simulation programs only (display programs are out of scope)
      if (!/^(simShader|precipitationShader)\.vert\+/.test(p)) delete u[p];
    const same = function(a, b) { return a.length == b.length && a.every(function(v, i) { return Object.is(v, b[i]); }); };
    for (const p of Object.keys(u))
      for (const n of Object.keys(u[p]))
        if (Array.isArray(u[p][n]) && u[p][n].length > 4)
          for (const k of ['initial_T', 'realWorldSounding_T', 'realWorldSounding_W', 'realWorldSounding_Vel'])
            if (Array.isArray(u[p][n]) && same(u[p][n], derived[k])) u[p][n] = u[p][n].every(function(v) { return v === 0; }) ? '@zeros' : '@' + k;
    return u;
  }
  // the 504-entry arrays (app.js:5442-5467): entries beyond Y are never written -- checked here, stored as the first Y + 1; all-zero -> null
  function trim(d, Y)
  {
    for (const k of ['initial_T', 'realWorldSounding_T', 'realWorldSounding_W', 'realWorldSounding_Vel']) {
      if (d[k].length != 504) throw new Error(k + ' has ' + d[k].length + ' entries');
      if (!d[k].slice(Y + 1).every(function(v) { return v === 0; })) throw new Error(k + ': non-zero entry beyond Y');
      d[k] = d[k].every(function(v) { return v === 0; }) ? null : d[k].slice(0, Y + 1);
    }
    d.array_length = 504;
    return d;
  }
  function pick(u)
  {
    const o = {};
    for (const p of Object.keys(u))
      if (/boundaryShader|lightingShader/.test(p)) o[p] = {sunAngle: u[p].sunAngle, sunIntensity: u[p].sunIntensity};
    return o;
  }
  U.meta.reference_slices = CITES;

  // ---- call traces
  const T = {meta: {generated_by: 'oracle/golden/gen_host_golden.js', reference_slices: CITES}, scenarios: []};
  const drops = [];
  for (let i = 0; i < 7; i++) drops.push(0.1 * i, 0.05 * i, i % 2 ? 0.3 : -9.5 + 0.1 * i, 0.2, 0.7);
  const traceJobs = [
    {name: 'precip_on_across_600', iter0: 598, n_drops: 7, precip: true, sound: false, iterations: 4},
    {name: 'precip_on_from_0', iter0: 0, n_drops: 7, precip: true, sound: false, iterations: 3},
    {name: 'precip_off', iter0: 41, n_drops: 7, precip: false, sound: false, iterations: 2},
    {name: 'sound_on_strike', iter0: 10, n_drops: 7, precip: true, sound: true, iterations: 2, readback2: 10},
    {name: 'weather_stations_across_208', iter0: 205, n_drops: 7, precip: false, sound: false, iterations: 10, weather_stations: true},
  ];
  for (const j of traceJobs) {
    const sc = {X: 64, Y: 48, saved_json: null, n_drops: j.n_drops, iter0: j.iter0, rain_drops: drops, readback2: j.readback2, weather_stations: j.weather_stations,
                gui_edit: {dayNightCycle: false, enablePrecipitation: j.precip, sound: j.sound}, frames: [{iterations: j.iterations}]};
    const r = await runScenario(sc);
    if (r.errors.length) throw new Error(j.name + ': ' + r.errors.join('; '));
    const f = r.frames[0];
    T.scenarios.push({name: j.name, X: 64, Y: 48, n_drops: j.n_drops, precip: j.precip, sound: j.sound, iter0: j.iter0, iterations_requested: j.iterations,
                      iterations_run: f.iter_after - f.iter_before, even_after: f.even_after, thunder: r.thunder || 0, setup_trace: r.setup_trace, tables: r.tables, trace: f.trace,
                      uniforms_after_setup: r.snapshots['after setup']});
  }
  fs.writeFileSync(path.join(outDir, 'host_uniforms.json'), JSON.stringify(U));
  fs.writeFileSync(path.join(outDir, 'host_call_trace.json'), JSON.stringify(T));
  console.log('wrote host_uniforms.json (' + U.scenarios.length + ' scenarios), host_call_trace.json (' + T.scenarios.length + ' scenarios)');
  console.log(JSON.stringify(CITES, null, 1));
}

main().catch(function(e) { console.error(e && e.stack ? e.stack : e); process.exit(1); });
