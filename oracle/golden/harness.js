/*
 * Golden-vector harness (TEST INFRASTRUCTURE, container-only): runs the REFERENCE shaders, loaded at
 * run time from file:///root/reference/shaders (nothing is copied into this repo), under
 * HeadlessChrome + SwiftShader as bundled with the `kaleido` pip package, and dumps simulation state.
 *
 * It is loaded through kaleido's plotly scope (`--plotlyjs=<this file>`), which calls
 * Plotly.toImage(gd, opts); gd.layout.wx carries the job description written by gen_golden.py.
 * The GL call sequence below restates the reference host: quad app.js:4770-4818, particle buffers
 * app.js:4915-5002, textures/FBOs app.js:5189-5317, constant uniforms app.js:5479-5635,
 * setGuiUniforms app.js:3401-3443, sun uniforms app.js:6557-6561 and the iteration loop app.js:5830-6005.
 */
(function() {
'use strict';
var REF = 'file:///root/reference/';

function loadText(url)
{
  var r = new XMLHttpRequest();
  r.open('GET', url, false);
  r.send(null);
  if (r.responseText == null || r.responseText.length == 0) throw 'empty: ' + url;
  return r.responseText;
}
function loadBin(url)
{
  var r = new XMLHttpRequest();
  r.open('GET', url, false);
  r.overrideMimeType('text/plain; charset=x-user-defined');
  r.send(null);
  var t = r.responseText, u = new Uint8Array(t.length);
  for (var i = 0; i < t.length; i++) u[i] = t.charCodeAt(i) & 255;
  return u;
}
function b64(typed)
{
  var u = new Uint8Array(typed.buffer, typed.byteOffset, typed.byteLength), s = '', CH = 0x8000;
  for (var i = 0; i < u.length; i += CH) s += String.fromCharCode.apply(null, u.subarray(i, i + CH));
  return btoa(s);
}

function run(o)
{
  var X = o.X, Y = o.Y, N = o.n_drops;
  var canvas = document.createElement('canvas');
  canvas.width = X;
  canvas.height = Y;
  var gl = canvas.getContext('webgl2', {alpha: false, antialias: true, depth: false, stencil: false, preserveDrawingBuffer: false});
  if (!gl) throw 'no webgl2';
  var exts = ['EXT_color_buffer_float', 'EXT_float_blend', 'OES_texture_float_linear', 'OES_texture_half_float_linear'];
  var extOk = {};
  exts.forEach(function(e) { extOk[e] = !!gl.getExtension(e); });
  gl.disable(gl.DEPTH_TEST);

  // ---- shaders (verbatim reference sources; only the uniform-array extent is patched so that
  //      4 x vec4[126] fit SwiftShader's MAX_FRAGMENT_UNIFORM_VECTORS) ----
  var ARR = Math.ceil((Y + 2) / 4);
  var common = loadText(REF + 'shaders/common.glsl');
  function source(kind, name)
  {
    var s = loadText(REF + 'shaders/' + kind + '/' + name);
    s = s.replace('#include "common.glsl"', common);
    return s.split('[126]').join('[' + ARR + ']');
  }
  function compile(type, src, name)
  {
    var sh = gl.createShader(type);
    gl.shaderSource(sh, src);
    gl.compileShader(sh);
    if (!gl.getShaderParameter(sh, gl.COMPILE_STATUS)) throw name + ': ' + gl.getShaderInfoLog(sh);
    return sh;
  }
  var simVert = compile(gl.VERTEX_SHADER, source('vertex', 'simShader.vert'), 'simShader.vert');
  // Point-drawn variant (job option `points`): one GL_POINT per pixel instead of the full-screen quad. SwiftShader corrupts
  // advectionShader output for air pixels that share a 2x2 pixel quad with a wall pixel when the pass is drawn as two triangles
  // (SURVEY Appendix C); a point primitive has no neighbouring fragments, so unaligned terrain renders correctly. The vertex
  // stage is the reference's simShader.vert with its two attributes turned into globals that a wrapper main() fills from
  // gl_VertexID: vertTexCoord = (pixel + 0.5) * uvScale, uvScale = f32(res * 1.0000001) / res, i.e. exactly the value the
  // quad's interpolated texture coordinate has at the pixel centre in exact arithmetic (app.js:4770-4788).
  var simVertPoints = null;
  if (o.points) {
    var sv = source('vertex', 'simShader.vert');
    sv = sv.replace('in vec2 vertPosition;', 'vec2 vertPosition;').replace('in vec2 vertTexCoord;', 'vec2 vertTexCoord;').replace('void main()', 'void simMain()');
    sv += '\nuniform vec2 wxRes;\nuniform vec2 wxUvScale;\nvoid main(){ int id = gl_VertexID; int W = int(wxRes.x); float px = float(id - (id / W) * W), py = float(id / W);' +
          ' vertPosition = vec2((px + 0.5) / wxRes.x * 2.0 - 1.0, (py + 0.5) / wxRes.y * 2.0 - 1.0);' +
          ' vertTexCoord = vec2((px + 0.5) * wxUvScale.x, (py + 0.5) * wxUvScale.y); simMain(); gl_PointSize = 1.0; }\n';
    simVertPoints = compile(gl.VERTEX_SHADER, sv, 'simShader.vert (points)');
  }
  function fragProgram(name)
  {
    var p = gl.createProgram();
    gl.attachShader(p, o.points ? simVertPoints : simVert);
    gl.attachShader(p, compile(gl.FRAGMENT_SHADER, source('fragment', name), name));
    gl.bindAttribLocation(p, 0, 'vertPosition');
    gl.bindAttribLocation(p, 1, 'vertTexCoord');
    gl.linkProgram(p);
    if (!gl.getProgramParameter(p, gl.LINK_STATUS)) throw name + ' link: ' + gl.getProgramInfoLog(p);
    return p;
  }
  var P = {};
  var progNames = ['velocity', 'curl', 'vorticity', 'boundary', 'advection', 'pressure', 'lighting', 'lightningLocation'];
  if (o.setup) progNames.push('setup');
  progNames.forEach(function(n) {
    P[n] = fragProgram(n + 'Shader.frag');
    if (o.points) {
      gl.useProgram(P[n]);
      gl.uniform2f(gl.getUniformLocation(P[n], 'wxRes'), X, Y);
      gl.uniform2f(gl.getUniformLocation(P[n], 'wxUvScale'), Math.fround(Math.fround(X * 1.0000001) / X), Math.fround(Math.fround(Y * 1.0000001) / Y));
    }
  });
  // particle program with transform feedback (app.js:4879-4881)
  P.precipitation = gl.createProgram();
  gl.attachShader(P.precipitation, compile(gl.VERTEX_SHADER, source('vertex', 'precipitationShader.vert'), 'precipitationShader.vert'));
  gl.attachShader(P.precipitation, compile(gl.FRAGMENT_SHADER, source('fragment', 'precipitationShader.frag'), 'precipitationShader.frag'));
  gl.bindAttribLocation(P.precipitation, 0, 'dropPosition');
  gl.bindAttribLocation(P.precipitation, 1, 'mass');
  gl.bindAttribLocation(P.precipitation, 2, 'density');
  gl.transformFeedbackVaryings(P.precipitation, ['position_out', 'mass_out', 'density_out'], gl.INTERLEAVED_ATTRIBS);
  gl.linkProgram(P.precipitation);
  if (!gl.getProgramParameter(P.precipitation, gl.LINK_STATUS)) throw 'precip link: ' + gl.getProgramInfoLog(P.precipitation);

  // ---- full-screen quad with the 1.0000001 UV scale (app.js:4770-4788) ----
  var quad = new Float32Array([1, -1, X * 1.0000001, 0, -1, -1, 0, 0, 1, 1, X * 1.0000001, Y * 1.0000001, -1, 1, 0, Y * 1.0000001]);
  var fluidVao = gl.createVertexArray();
  gl.bindVertexArray(fluidVao);
  var qb = gl.createBuffer();
  gl.bindBuffer(gl.ARRAY_BUFFER, qb);
  gl.bufferData(gl.ARRAY_BUFFER, quad, gl.STATIC_DRAW);
  gl.enableVertexAttribArray(0);
  gl.enableVertexAttribArray(1);
  gl.vertexAttribPointer(0, 2, gl.FLOAT, false, 16, 0);
  gl.vertexAttribPointer(1, 2, gl.FLOAT, false, 16, 8);
  gl.bindVertexArray(null);

  // ---- optional probe: dump the interpolated varyings of simShader.vert (what fragCoord/texCoord
  //      really are under this rasteriser) ----
  if (o.probe) {
    var pf = '#version 300 es\nprecision highp float;\nin vec2 fragCoord;\nin vec2 texCoord;\nout vec4 o;\nvoid main(){ o = vec4(fragCoord, texCoord); }';
    var pp = gl.createProgram();
    gl.attachShader(pp, o.points ? simVertPoints : simVert);
    gl.attachShader(pp, compile(gl.FRAGMENT_SHADER, pf, 'probe'));
    gl.bindAttribLocation(pp, 0, 'vertPosition');
    gl.bindAttribLocation(pp, 1, 'vertTexCoord');
    gl.linkProgram(pp);
    gl.useProgram(pp);
    gl.uniform2f(gl.getUniformLocation(pp, 'texelSize'), 1.0 / X, 1.0 / Y);
    if (o.points) {
      gl.uniform2f(gl.getUniformLocation(pp, 'wxRes'), X, Y);
      gl.uniform2f(gl.getUniformLocation(pp, 'wxUvScale'), Math.fround(Math.fround(X * 1.0000001) / X), Math.fround(Math.fround(Y * 1.0000001) / Y));
    }
    var pt = gl.createTexture();
    gl.bindTexture(gl.TEXTURE_2D, pt);
    gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA32F, X, Y, 0, gl.RGBA, gl.FLOAT, null);
    var pfb = gl.createFramebuffer();
    gl.bindFramebuffer(gl.FRAMEBUFFER, pfb);
    gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, pt, 0);
    gl.viewport(0, 0, X, Y);
    gl.bindVertexArray(fluidVao);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0]);
    if (o.points) {
      gl.bindVertexArray(gl.createVertexArray());
      gl.drawArrays(gl.POINTS, 0, X * Y);
    } else {
      gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
    }
    var pa = new Float32Array(4 * X * Y);
    gl.readPixels(0, 0, X, Y, gl.RGBA, gl.FLOAT, pa);
    return {probe: b64(pa), renderer: gl.getParameter(gl.RENDERER)};
  }

  // ---- job option `setup`: the setup draw of a new simulation (setupShader.frag, uniforms app.js:5479-5485, 5732-5734) ----
  if (o.setup) {
    var sT = {};
    function stex(ifmt, fmt, type)
    {
      var t = gl.createTexture();
      gl.bindTexture(gl.TEXTURE_2D, t);
      gl.texImage2D(gl.TEXTURE_2D, 0, ifmt, X, Y, 0, fmt, type, null);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
      return t;
    }
    sT.base = stex(gl.RGBA32F, gl.RGBA, gl.FLOAT);
    sT.water = stex(gl.RGBA32F, gl.RGBA, gl.FLOAT);
    sT.wall = stex(gl.RGBA8I, gl.RGBA_INTEGER, gl.BYTE);
    var sfb = gl.createFramebuffer();
    gl.bindFramebuffer(gl.FRAMEBUFFER, sfb);
    gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, sT.base, 0);
    gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT1, gl.TEXTURE_2D, sT.water, 0);
    gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT2, gl.TEXTURE_2D, sT.wall, 0);
    gl.useProgram(P.setup);
    var sInit = new Float32Array(ARR * 4);
    for (var sy = 0; sy < Y + 1; sy++) sInit[sy] = o.initial_T[sy];
    gl.uniform2f(gl.getUniformLocation(P.setup, 'texelSize'), 1.0 / X, 1.0 / Y);
    gl.uniform2f(gl.getUniformLocation(P.setup, 'resolution'), X, Y);
    gl.uniform1f(gl.getUniformLocation(P.setup, 'dryLapse'), o.setup.dryLapse);
    gl.uniform1f(gl.getUniformLocation(P.setup, 'simHeight'), o.setup.simHeight);
    gl.uniform4fv(gl.getUniformLocation(P.setup, 'initial_Tv'), sInit);
    gl.uniform1f(gl.getUniformLocation(P.setup, 'seed'), o.setup.seed);
    gl.uniform1f(gl.getUniformLocation(P.setup, 'heightMult'), o.setup.heightMult);
    gl.viewport(0, 0, X, Y);
    gl.bindVertexArray(fluidVao);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1, gl.COLOR_ATTACHMENT2]);
    if (o.points) {
      gl.bindVertexArray(gl.createVertexArray());
      gl.drawArrays(gl.POINTS, 0, X * Y);
    } else {
      gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
    }
    var so = {renderer: gl.getParameter(gl.RENDERER), err: 0};
    var sa = new Float32Array(4 * X * Y);
    gl.readBuffer(gl.COLOR_ATTACHMENT0);
    gl.readPixels(0, 0, X, Y, gl.RGBA, gl.FLOAT, sa);
    so.base = b64(sa);
    gl.readBuffer(gl.COLOR_ATTACHMENT1);
    gl.readPixels(0, 0, X, Y, gl.RGBA, gl.FLOAT, sa);
    so.water = b64(sa);
    var si = new Int32Array(4 * X * Y), s8 = new Int8Array(4 * X * Y);
    gl.readBuffer(gl.COLOR_ATTACHMENT2);
    gl.readPixels(0, 0, X, Y, gl.RGBA_INTEGER, gl.INT, si);
    for (var q = 0; q < si.length; q++) s8[q] = si[q];
    so.wall = b64(s8);
    so.err = gl.getError();
    return so;
  }

  // ---- fixture ----
  var dir = o.dir;
  function f32(name) { var u = loadBin(dir + name); return new Float32Array(u.buffer, 0, u.length >> 2); }
  var base0 = f32('base.f32'), water0 = f32('water.f32');
  var wall0 = new Int8Array(loadBin(dir + 'wall.i8').buffer);
  var drops0 = N > 0 ? f32('drops.f32') : new Float32Array(5);

  // ---- particle buffers / VAOs / TFs (app.js:4891-5002) ----
  function dropSet()
  {
    var vao = gl.createVertexArray(), buf = gl.createBuffer(), tf = gl.createTransformFeedback();
    gl.bindVertexArray(vao);
    gl.bindBuffer(gl.ARRAY_BUFFER, buf);
    gl.bufferData(gl.ARRAY_BUFFER, drops0, gl.STATIC_DRAW);
    for (var a = 0; a < 3; a++) gl.enableVertexAttribArray(a);
    gl.vertexAttribPointer(0, 2, gl.FLOAT, false, 20, 0);
    gl.vertexAttribPointer(1, 2, gl.FLOAT, false, 20, 8);
    gl.vertexAttribPointer(2, 1, gl.FLOAT, false, 20, 16);
    gl.bindTransformFeedback(gl.TRANSFORM_FEEDBACK, tf);
    gl.bindBufferBase(gl.TRANSFORM_FEEDBACK_BUFFER, 0, buf);
    gl.bindTransformFeedback(gl.TRANSFORM_FEEDBACK, null);
    gl.bindBufferBase(gl.TRANSFORM_FEEDBACK_BUFFER, 0, null);
    gl.bindBuffer(gl.ARRAY_BUFFER, null);
    gl.bindVertexArray(null);
    return {vao: vao, buf: buf, tf: tf};
  }
  var D = [dropSet(), dropSet()];

  // ---- textures / FBOs (app.js:5189-5317) ----
  function tex(ifmt, w, h, fmt, type, data, filter, clampT)
  {
    var t = gl.createTexture();
    gl.bindTexture(gl.TEXTURE_2D, t);
    gl.texImage2D(gl.TEXTURE_2D, 0, ifmt, w, h, 0, fmt, type, data);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, filter);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, filter);
    if (clampT) gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.CLAMP_TO_EDGE);
    return t;
  }
  var T = {};
  for (var k = 0; k < 2; k++) {
    T['base' + k] = tex(gl.RGBA32F, X, Y, gl.RGBA, gl.FLOAT, base0, gl.NEAREST);
    T['water' + k] = tex(gl.RGBA32F, X, Y, gl.RGBA, gl.FLOAT, water0, gl.NEAREST);
    T['wall' + k] = tex(gl.RGBA8I, X, Y, gl.RGBA_INTEGER, gl.BYTE, wall0, gl.NEAREST);
    T['light' + k] = tex(gl.RGBA32F, X, Y, gl.RGBA, gl.FLOAT, null, gl.LINEAR, true);
  }
  T.curl = tex(gl.R32F, X, Y, gl.RED, gl.FLOAT, null, gl.NEAREST);
  T.vort = tex(gl.RG32F, X, Y, gl.RG, gl.FLOAT, null, gl.NEAREST);
  T.emitted = tex(gl.RGBA16F, X, Y, gl.RGBA, gl.HALF_FLOAT, null, gl.LINEAR, true); // new FBO(..., gl.LINEAR), app.js:786-801, 838: clamped in S and T
  gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.CLAMP_TO_EDGE);
  T.fb = tex(gl.RGBA32F, X, Y, gl.RGBA, gl.FLOAT, null, gl.NEAREST);
  T.dep = tex(gl.RG32F, X, Y, gl.RG, gl.FLOAT, null, gl.NEAREST);
  T.lightning = tex(gl.RGBA32F, 1, 1, gl.RGBA, gl.FLOAT, null, gl.NEAREST);
  function fbo(list)
  {
    var f = gl.createFramebuffer();
    gl.bindFramebuffer(gl.FRAMEBUFFER, f);
    for (var i = 0; i < list.length; i++) gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0 + i, gl.TEXTURE_2D, list[i], 0);
    return f;
  }
  var F = {
    fb0: fbo([T.base0, T.water0, T.wall0]),
    fb1: fbo([T.base1, T.water1, T.wall1]),
    curl: fbo([T.curl]),
    vort: fbo([T.vort]),
    light0: fbo([T.light0, T.emitted]),
    light1: fbo([T.light1, T.emitted]),
    precip: fbo([T.fb, T.dep]),
    lightning: fbo([T.lightning])
  };
  // zero the textures created with null data (GL leaves them zero-initialised in WebGL, but be explicit)
  gl.clearColor(0, 0, 0, 0);

  // ---- uniforms ----
  var U = o.uniforms, initT = new Float32Array(ARR * 4), sndZero = new Float32Array(ARR * 4);
  for (var y = 0; y < Y + 1; y++) initT[y] = o.initial_T[y];
  // realWorldSounding_* (app.js:5444-5463): zero unless the job carries per-row arrays
  var sndT = new Float32Array(ARR * 4), sndW = new Float32Array(ARR * 4), sndV = new Float32Array(ARR * 4);
  if (o.sounding)
    for (var ys = 0; ys < Y + 1; ys++) {
      sndT[ys] = o.sounding.T[ys];
      sndW[ys] = o.sounding.W[ys];
      sndV[ys] = o.sounding.Vel[ys];
    }
  function setU(prog, list)
  {
    gl.useProgram(P[prog]);
    list.forEach(function(e) {
      var loc = gl.getUniformLocation(P[prog], e[0]);
      if (loc === null) return;
      var v = e[2];
      if (e[1] == '1i') gl.uniform1i(loc, v);
      else if (e[1] == '1f') gl.uniform1f(loc, v);
      else if (e[1] == '2f') gl.uniform2f(loc, v[0], v[1]);
      else if (e[1] == '4f') gl.uniform4f(loc, v[0], v[1], v[2], v[3]);
      else if (e[1] == '4fv') gl.uniform4fv(loc, v);
    });
  }
  var texel = [1.0 / X, 1.0 / Y], res = [X, Y];
  setU('advection', [
    ['baseTex', '1i', 0], ['waterTex', '1i', 1], ['wallTex', '1i', 2], ['texelSize', '2f', texel], ['resolution', '2f', res],
    ['initial_Tv', '4fv', initT], ['dryLapse', '1f', U.dryLapse], ['waterTemperature', '1f', U.waterTemperature],
    ['realWorldSounding_Tv', '4fv', sndT], ['realWorldSounding_Wv', '4fv', sndW], ['realWorldSounding_Velv', '4fv', sndV],
    ['evapHeat', '1f', U.evapHeat], ['meltingHeat', '1f', U.meltingHeat], ['condensationRate', '1f', U.condensationRate],
    ['globalDrying', '1f', U.globalDrying], ['globalHeating', '1f', U.globalHeating], ['soundingForcing', '1f', U.soundingForcing],
    ['globalEffectsStartAlt', '1f', U.globalEffectsStartAlt], ['globalEffectsEndAlt', '1f', U.globalEffectsEndAlt],
    ['userInputType', '1i', U.userInputType], ['userInputValues', '4f', U.userInputValues], ['userInputMove', '2f', U.userInputMove],
    ['wrapHorizontally', '1i', U.wrapHorizontally], ['airplaneValues', '4f', U.airplaneValues]
  ]);
  setU('pressure', [['baseTex', '1i', 0], ['wallTex', '1i', 1], ['texelSize', '2f', texel]]);
  setU('velocity', [
    ['baseTex', '1i', 0], ['wallTex', '1i', 1], ['texelSize', '2f', texel], ['initial_Tv', '4fv', initT],
    ['dragMultiplier', '1f', U.dragMultiplier], ['wind', '1f', U.wind]
  ]);
  setU('vorticity', [['texelSize', '2f', texel], ['curlTex', '1i', 0]]);
  setU('curl', [['texelSize', '2f', texel], ['baseTex', '1i', 0]]);
  setU('boundary', [
    ['baseTex', '1i', 0], ['waterTex', '1i', 1], ['vortForceTex', '1i', 2], ['wallTex', '1i', 3], ['lightTex', '1i', 4],
    ['precipFeedbackTex', '1i', 5], ['precipDepositionTex', '1i', 6], ['resolution', '2f', res], ['texelSize', '2f', texel],
    ['vorticity', '1f', U.vorticity], ['dryLapse', '1f', U.dryLapse], ['initial_Tv', '4fv', initT],
    ['landEvaporation', '1f', U.landEvaporation], ['waterEvaporation', '1f', U.waterEvaporation],
    ['dynamicWaterTemperature', '1f', U.dynamicWaterTemperature], ['evapHeat', '1f', U.evapHeat], ['waterWeight', '1f', U.waterWeight],
    ['sunAngle', '1f', U.sunAngle]
  ]);
  setU('lighting', [
    ['resolution', '2f', res], ['texelSize', '2f', texel], ['baseTex', '1i', 0], ['waterTex', '1i', 1], ['wallTex', '1i', 2],
    ['lightTex', '1i', 3], ['dryLapse', '1f', U.dryLapse], ['greenhouseGases', '1f', U.greenhouseGases],
    ['waterGreenHouseEffect', '1f', U.waterGreenHouseEffect], ['IR_rate', '1f', U.IR_rate], ['sunIntensity', '1f', U.sunIntensity],
    ['sunAngle', '1f', U.sunAngle]
  ]);
  setU('precipitation', [
    ['baseTex', '1i', 0], ['waterTex', '1i', 1], ['lightningDataTex', '1i', 2], ['resolution', '2f', res], ['texelSize', '2f', texel],
    ['dryLapse', '1f', U.dryLapse], ['evapHeat', '1f', U.evapHeat], ['meltingHeat', '1f', U.meltingHeat],
    ['aboveZeroThreshold', '1f', U.aboveZeroThreshold], ['subZeroThreshold', '1f', U.subZeroThreshold],
    ['spawnChanceMult', '1f', U.spawnChanceMult], ['snowDensity', '1f', U.snowDensity], ['fallSpeed', '1f', U.fallSpeed],
    ['growthRate0C', '1f', U.growthRate0C], ['growthRate_30C', '1f', U.growthRate_30C], ['freezingRate', '1f', U.freezingRate],
    ['meltingRate', '1f', U.meltingRate], ['evapRate', '1f', U.evapRate], ['inactiveDroplets', '1f', U.inactiveDroplets]
  ]);
  setU('lightningLocation', [['precipFeedbackTex', '1i', 0], ['resolution', '2f', res], ['texelSize', '2f', texel]]);

  // ---- readback helpers ----
  function readF(fb, att, ch)
  {
    gl.bindFramebuffer(gl.FRAMEBUFFER, fb);
    gl.readBuffer(gl.COLOR_ATTACHMENT0 + att);
    var a = new Float32Array(4 * X * Y);
    gl.readPixels(0, 0, X, Y, gl.RGBA, gl.FLOAT, a);
    if (ch == 4) return a;
    var r = new Float32Array(ch * X * Y);
    for (var i = 0; i < X * Y; i++)
      for (var c = 0; c < ch; c++) r[i * ch + c] = a[i * 4 + c];
    return r;
  }
  function readI8(fb, att)
  {
    gl.bindFramebuffer(gl.FRAMEBUFFER, fb);
    gl.readBuffer(gl.COLOR_ATTACHMENT0 + att);
    var a = new Int32Array(4 * X * Y);
    gl.readPixels(0, 0, X, Y, gl.RGBA_INTEGER, gl.INT, a); // SwiftShader only allows INT
    var r = new Int8Array(4 * X * Y);
    for (var i = 0; i < a.length; i++) r[i] = a[i];
    return r;
  }
  function readDrops(set)
  {
    var a = new Float32Array(5 * Math.max(N, 1));
    gl.bindBuffer(gl.ARRAY_BUFFER, set.buf);
    gl.getBufferSubData(gl.ARRAY_BUFFER, 0, a);
    gl.bindBuffer(gl.ARRAY_BUFFER, null);
    return a;
  }
  function readLightning()
  {
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.lightning);
    gl.readBuffer(gl.COLOR_ATTACHMENT0);
    var a = new Float32Array(4);
    gl.readPixels(0, 0, 1, 1, gl.RGBA, gl.FLOAT, a);
    return a;
  }
  var out = {ext: extOk, dumps: {}, perpass: {}, renderer: gl.getParameter(gl.RENDERER), version: gl.getParameter(gl.VERSION)};
  function put(dst, key, arr) { dst[key] = b64(arr); }

  function bind(unit, t)
  {
    gl.activeTexture(gl.TEXTURE0 + unit);
    gl.bindTexture(gl.TEXTURE_2D, t);
  }
  var emptyVao = gl.createVertexArray();
  function quadDraw()
  {
    if (o.points) { // attribute-less draw: the wrapper vertex stage derives everything from gl_VertexID
      gl.bindVertexArray(emptyVao);
      gl.drawArrays(gl.POINTS, 0, X * Y);
      gl.bindVertexArray(fluidVao);
    } else {
      gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
    }
  }

  // ---- the iteration (app.js:5830-6005) ----
  var even = true, iterNum = o.iter0 || 0, lastDst = 0;
  gl.viewport(0, 0, X, Y);
  gl.bindVertexArray(fluidVao);
  function iteration(pp)
  {
    gl.useProgram(P.velocity);
    bind(0, T.base0);
    bind(1, T.wall0);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.fb1);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.NONE, gl.COLOR_ATTACHMENT2]);
    quadDraw();
    if (pp) { put(pp, 'velocity_base', readF(F.fb1, 0, 4)); put(pp, 'velocity_wall', readI8(F.fb1, 2)); }

    gl.useProgram(P.curl);
    bind(0, T.base1);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.curl);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0]);
    quadDraw();
    if (pp) put(pp, 'curl', readF(F.curl, 0, 1));

    gl.useProgram(P.vorticity);
    bind(0, T.curl);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.vort);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0]);
    quadDraw();
    if (pp) put(pp, 'vort', readF(F.vort, 0, 2));

    gl.useProgram(P.boundary);
    gl.uniform1f(gl.getUniformLocation(P.boundary, 'iterNum'), iterNum);
    bind(0, T.base1);
    bind(1, T.water1);
    bind(2, T.vort);
    bind(3, T.wall1);
    bind(4, T.light0);
    bind(5, T.fb);
    bind(6, T.dep);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.fb0);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1, gl.COLOR_ATTACHMENT2]);
    quadDraw();
    if (pp) { put(pp, 'boundary_base', readF(F.fb0, 0, 4)); put(pp, 'boundary_water', readF(F.fb0, 1, 4)); put(pp, 'boundary_wall', readI8(F.fb0, 2)); }

    gl.useProgram(P.advection);
    bind(0, T.base0);
    bind(1, T.water0);
    bind(2, T.wall0);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.fb1);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1, gl.COLOR_ATTACHMENT2]);
    quadDraw();
    if (pp) { put(pp, 'advection_base', readF(F.fb1, 0, 4)); put(pp, 'advection_water', readF(F.fb1, 1, 4)); put(pp, 'advection_wall', readI8(F.fb1, 2)); }

    gl.useProgram(P.pressure);
    bind(0, T.base1);
    bind(1, T.wall1);
    gl.bindFramebuffer(gl.FRAMEBUFFER, F.fb0);
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.NONE, gl.COLOR_ATTACHMENT2]);
    quadDraw();
    if (pp) { put(pp, 'pressure_base', readF(F.fb0, 0, 4)); put(pp, 'pressure_wall', readI8(F.fb0, 2)); }

    gl.useProgram(P.lighting);
    bind(0, T.base1);
    bind(1, T.water1);
    bind(2, T.wall1);
    gl.activeTexture(gl.TEXTURE3);
    var src, dst, dstLightFb;
    if (even) {
      gl.bindTexture(gl.TEXTURE_2D, T.light0);
      dstLightFb = F.light1;
      src = D[0];
      dst = D[1];
      lastDst = 1;
    } else {
      gl.bindTexture(gl.TEXTURE_2D, T.light1);
      dstLightFb = F.light0;
      src = D[1];
      dst = D[0];
      lastDst = 0;
    }
    gl.bindFramebuffer(gl.FRAMEBUFFER, dstLightFb);
    even = !even;
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1]);
    quadDraw();
    if (pp) put(pp, 'lighting_light', readF(dstLightFb, 0, 4));

    gl.bindFramebuffer(gl.FRAMEBUFFER, F.precip);
    gl.clear(gl.COLOR_BUFFER_BIT);

    if (o.precip && N > 0) {
      gl.useProgram(P.precipitation);
      gl.uniform1f(gl.getUniformLocation(P.precipitation, 'iterNum'), iterNum);
      gl.enable(gl.BLEND);
      gl.blendFunc(gl.ONE, gl.ONE);
      bind(0, T.base1);
      bind(1, T.water1);
      bind(2, T.lightning);
      gl.bindVertexArray(src.vao);
      gl.bindTransformFeedback(gl.TRANSFORM_FEEDBACK, dst.tf);
      gl.beginTransformFeedback(gl.POINTS);
      gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1]);
      gl.drawArrays(gl.POINTS, 0, N);
      gl.endTransformFeedback();
      if (iterNum % 600 == 0) {
        gl.readBuffer(gl.COLOR_ATTACHMENT0);
        var sv = new Float32Array(4);
        gl.readPixels(0, 0, 1, 1, gl.RGBA, gl.FLOAT, sv);
        gl.uniform1f(gl.getUniformLocation(P.precipitation, 'inactiveDroplets'), sv[0]);
        out.inactiveDroplets = sv[0];
      }
      gl.bindTransformFeedback(gl.TRANSFORM_FEEDBACK, null);
      gl.disable(gl.BLEND);
      gl.bindVertexArray(fluidVao);
      if (pp) { put(pp, 'precip_fb', readF(F.precip, 0, 4)); put(pp, 'precip_dep', readF(F.precip, 1, 2)); put(pp, 'precip_drops', readDrops(dst)); }

      gl.useProgram(P.lightningLocation);
      gl.uniform1f(gl.getUniformLocation(P.lightningLocation, 'iterNum'), iterNum);
      bind(0, T.fb);
      gl.bindFramebuffer(gl.FRAMEBUFFER, F.lightning);
      gl.drawBuffers([gl.COLOR_ATTACHMENT0]);
      quadDraw();
    }
    iterNum++;
  }

  function dumpState(key)
  {
    var d = {};
    put(d, 'base_cur', readF(F.fb0, 0, 4));
    put(d, 'base_disp', readF(F.fb1, 0, 4));
    put(d, 'water_0', readF(F.fb0, 1, 4));
    put(d, 'water_cur', readF(F.fb1, 1, 4));
    put(d, 'wall_cur', readI8(F.fb0, 2));
    put(d, 'wall_disp', readI8(F.fb1, 2));
    put(d, 'light_0', readF(F.light0, 0, 4));
    put(d, 'light_1', readF(F.light1, 0, 4));
    if (o.dump_emitted) put(d, 'emitted', readF(F.light0, 1, 4)); // emittedLight (lightingShader's second output, RGBA16F)
    if (o.precip && N > 0) {
      put(d, 'drops', readDrops(D[lastDst]));
      put(d, 'lightning', readLightning());
      put(d, 'precip_fb', readF(F.precip, 0, 4));
      put(d, 'precip_dep', readF(F.precip, 1, 2));
    }
    out.dumps[key] = d;
  }

  var dumpAt = {};
  (o.dump_iters || []).forEach(function(i) { dumpAt[i] = true; });
  var t0 = 0, t1 = 0;
  for (var it = 0; it < o.niter; it++) {
    if (it == 1) { gl.finish(); t0 = performance.now(); }
    var pp = null;
    if (o.perpass_iter === it) pp = out.perpass;
    if (typeof o.mark == 'function') o.mark('iteration', it, gl); // (run_harness_mock.js: cuts the recorded GL calls into iterations)
    iteration(pp);
    if (typeof o.mark == 'function') o.mark('iteration_end', it, gl);
    if (dumpAt[it + 1]) dumpState(String(it + 1));
  }
  // a 1x1 readback forces completion (gl.finish is not a reliable sync under ANGLE)
  readLightning();
  t1 = performance.now();
  out.ms_after_first = t1 - t0;
  out.niter = o.niter;
  out.err = gl.getError();
  return out;
}

window.Plotly = {
  version: '2.0.0',
  purge: function() {},
  toImage: function(gd, opts) {
    var res;
    try {
      res = run(gd.layout.wx);
    } catch (e) {
      res = {error: String(e && e.stack ? e.stack : e)};
    }
    return Promise.resolve(JSON.stringify(res));
  }
};
})();
