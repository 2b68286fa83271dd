/*
 * Recording mock of the WebGL2 context (TEST INFRASTRUCTURE, shared by gen_host_golden.js and run_harness_mock.js).
 *
 * Nothing is rendered: every call is appended to a trace with its arguments decoded (enum numbers -> names, GL objects -> tokens),
 * and the state a later comparison needs is tracked: texture storage / sampler parameters, framebuffer attachments, uniform values per
 * program, vertex-array layout, transform-feedback bindings.  Enum values are the WebGL2 specification's (arithmetic such as
 * gl.TEXTURE0 + unit and gl.COLOR_ATTACHMENT0 + i must work).
 */
'use strict';

const ENUM = {
  NONE: 0, POINTS: 0, ONE: 1, TRIANGLE_STRIP: 5, FALSE: 0, TRUE: 1,
  DEPTH_TEST: 0x0B71, BLEND: 0x0BE2, TEXTURE_2D: 0x0DE1, BYTE: 0x1400, UNSIGNED_BYTE: 0x1401, INT: 0x1404, FLOAT: 0x1406, HALF_FLOAT: 0x140B,
  RED: 0x1903, RGBA: 0x1908, RENDERER: 0x1F01, VERSION: 0x1F02, NEAREST: 0x2600, LINEAR: 0x2601, NEAREST_MIPMAP_LINEAR: 0x2702,
  TEXTURE_MAG_FILTER: 0x2800, TEXTURE_MIN_FILTER: 0x2801, TEXTURE_WRAP_S: 0x2802, TEXTURE_WRAP_T: 0x2803, REPEAT: 0x2901,
  COLOR_BUFFER_BIT: 0x4000, CLAMP_TO_EDGE: 0x812F, RG: 0x8227, R32F: 0x822E, RG32F: 0x8230, TEXTURE0: 0x84C0,
  RGBA32F: 0x8814, RGBA16F: 0x881A, ARRAY_BUFFER: 0x8892, STATIC_DRAW: 0x88E4, FRAGMENT_SHADER: 0x8B30, VERTEX_SHADER: 0x8B31,
  COMPILE_STATUS: 0x8B81, LINK_STATUS: 0x8B82, INTERLEAVED_ATTRIBS: 0x8C8C, TRANSFORM_FEEDBACK_BUFFER: 0x8C8E, COLOR_ATTACHMENT0: 0x8CE0,
  FRAMEBUFFER: 0x8D40, RGBA8I: 0x8D8E, RGBA_INTEGER: 0x8D99, TRANSFORM_FEEDBACK: 0x8E22,
};
for (let i = 1; i < 16; i++) {
  ENUM['TEXTURE' + i] = ENUM.TEXTURE0 + i;
  ENUM['COLOR_ATTACHMENT' + i] = ENUM.COLOR_ATTACHMENT0 + i;
}
const NAME = {}; // value -> name, for the enum classes that do not collide
for (const k of Object.keys(ENUM))
  if (!['NONE', 'POINTS', 'ONE', 'FALSE', 'TRUE', 'TRIANGLE_STRIP'].includes(k)) NAME[ENUM[k]] = k;
function en(v) { return NAME[v] !== undefined ? NAME[v] : v; }
function attach(v) { return v === 0 ? 'NONE' : en(v); }
function prim(v) { return v === 0 ? 'POINTS' : v === 5 ? 'TRIANGLE_STRIP' : v; }

function create(opts)
{
  opts = opts || {};
  const st = {
    trace: [], objects: {}, counters: {}, program: null, unit: 0, units: {}, fbo: null, vao: null, arrayBuffer: null, tf: null,
    errors: [], ignored: [],
  };
  function obj(kind, extra)
  {
    const n = (st.counters[kind] = (st.counters[kind] || 0) + 1);
    const o = Object.assign({__kind: kind, __id: kind + '@' + n}, extra || {});
    st.objects[o.__id] = o;
    return o;
  }
  function tok(o)
  {
    if (o === null || o === undefined) return null;
    if (o.__id !== undefined) return o.__name || o.__id;
    return o;
  }
  // (GL objects are stored as such and turned into tokens when the trace is read: the caller may name them after creating them)
  function rec() { st.trace.push(Array.prototype.slice.call(arguments)); }
  function data(d)
  {
    if (d === null || d === undefined) return null;
    if (d.__data) return d.__data;
    if (ArrayBuffer.isView(d)) return {typed: d.constructor.name, length: d.length};
    return String(d);
  }
  function boundTex() { return st.units[st.unit] || null; }
  function setUniform(fn, loc, value)
  {
    if (loc === null || loc === undefined) return; // (GL ignores a null location)
    if (loc.undeclared) {
      const key = tok(loc.program) + '.' + loc.name;
      if (st.ignored.indexOf(key) < 0) st.ignored.push(key);
      return;
    }
    if (loc.program !== st.program) st.errors.push('uniform ' + loc.name + ' set while ' + tok(st.program) + ' is current (INVALID_OPERATION)');
    (loc.program.uniforms = loc.program.uniforms || {})[loc.name] = {fn: fn, value: value};
    rec(fn, {__uniform: loc}, value);
  }
  const f32 = Math.fround;
  const gl = {
    __state: st,
    canvas: opts.canvas || {width: 0, height: 0},
    getExtension: function(n) { return {}; },
    getParameter: function(p) { return p === ENUM.RENDERER ? 'mockgl (no rendering)' : p === ENUM.VERSION ? 'WebGL 2.0 (mock)' : 0; },
    getError: function() { return 0; },
    finish: function() {},
    // ---- objects
    createTexture: function() { return obj('texture', {params: {}}); },
    createFramebuffer: function() { return obj('framebuffer', {attachments: {}}); },
    createVertexArray: function() { return obj('vao', {attribs: {}, enabled: []}); },
    createBuffer: function() { return obj('buffer', {}); },
    createTransformFeedback: function() { return obj('tf', {buffers: {}}); },
    createProgram: function() { return obj('program', {shaders: []}); },
    createShader: function(type) { return obj('shader', {type: en(type), file: null}); },
    shaderSource: function(sh, src)
    {
      const m = /\/\/@@file:(\S+)/.exec(src);
      sh.file = m ? m[1] : null;
    },
    compileShader: function() {},
    getShaderParameter: function() { return true; },
    getShaderInfoLog: function() { return ''; },
    attachShader: function(p, sh)
    {
      p.shaders.push(sh);
      if (!p.__name && p.shaders.length == 2 && p.shaders.every(function(s) { return s.file; })) {
        const v = p.shaders.find(function(s) { return s.type == 'VERTEX_SHADER'; }), f = p.shaders.find(function(s) { return s.type == 'FRAGMENT_SHADER'; });
        if (v && f) p.__name = v.file + '+' + f.file;
      }
    },
    bindAttribLocation: function(p, i, n) { (p.attribs = p.attribs || {})[n] = i; },
    transformFeedbackVaryings: function(p, names, mode) { p.tfVaryings = names.slice(); },
    linkProgram: function() {},
    getProgramParameter: function() { return true; },
    getProgramInfoLog: function() { return ''; },
    // a program that knows its shaders' uniform declarations (p.declared, a Set) answers null for any other name, as GL does; pushes
    // to a null location are ignored by GL -- they are listed in st.ignored
    getUniformLocation: function(p, name)
    {
      if (p.declared && !p.declared.has(name)) return {program: p, name: name, undeclared: true};
      return {program: p, name: name};
    },
    getAttribLocation: function(p, name) { return (opts.attribLocations || {})[name] !== undefined ? opts.attribLocations[name] : -1; },
    // ---- state
    useProgram: function(p) { st.program = p; rec('useProgram', p); },
    activeTexture: function(u) { st.unit = u - ENUM.TEXTURE0; rec('activeTexture', st.unit); },
    bindTexture: function(target, t) { st.units[st.unit] = t; rec('bindTexture', en(target), t); },
    bindFramebuffer: function(target, f) { st.fbo = f; rec('bindFramebuffer', f); },
    drawBuffers: function(list) { rec('drawBuffers', list.map(attach)); },
    readBuffer: function(a) { rec('readBuffer', attach(a)); },
    viewport: function(x, y, w, h) { rec('viewport', x, y, w, h); },
    clearColor: function(r, g, b, a) { rec('clearColor', r, g, b, a); },
    clear: function(mask) { rec('clear', en(mask)); },
    enable: function(c) { rec('enable', en(c)); },
    disable: function(c) { rec('disable', en(c)); },
    blendFunc: function(s, d) { rec('blendFunc', s === 1 ? 'ONE' : en(s), d === 1 ? 'ONE' : en(d)); },
    bindVertexArray: function(v) { st.vao = v; rec('bindVertexArray', v); },
    bindBuffer: function(target, b)
    {
      if (target === ENUM.ARRAY_BUFFER) st.arrayBuffer = b;
      rec('bindBuffer', en(target), b);
    },
    bufferData: function(target, d, usage)
    {
      const b = target === ENUM.ARRAY_BUFFER ? st.arrayBuffer : null;
      if (b) b.data = ArrayBuffer.isView(d) ? Array.from(d) : data(d);
      if (b && d && d.__data) b.data = d.__data;
      rec('bufferData', en(target), b, en(usage));
    },
    enableVertexAttribArray: function(i) { if (st.vao) st.vao.enabled.push(i); rec('enableVertexAttribArray', i); },
    vertexAttribPointer: function(i, size, type, norm, stride, off)
    {
      if (st.vao) st.vao.attribs[i] = {size: size, type: en(type), normalized: !!norm, stride: stride, offset: off, buffer: st.arrayBuffer};
      rec('vertexAttribPointer', i, size, en(type), !!norm, stride, off);
    },
    bindTransformFeedback: function(target, t) { st.tf = t; rec('bindTransformFeedback', t); },
    bindBufferBase: function(target, idx, b)
    {
      if (target === ENUM.TRANSFORM_FEEDBACK_BUFFER && st.tf && b) st.tf.buffers[idx] = b;
      rec('bindBufferBase', en(target), idx, b);
    },
    beginTransformFeedback: function(m) { rec('beginTransformFeedback', prim(m)); },
    endTransformFeedback: function() { rec('endTransformFeedback'); },
    // ---- textures / framebuffers
    texImage2D: function(target, level, ifmt, w, h, border, fmt, type, d)
    {
      const t = boundTex();
      if (t) t.storage = {internalformat: en(ifmt), width: w, height: h, format: en(fmt), type: en(type), data: data(d)};
      rec('texImage2D', t, en(ifmt), w, h, en(fmt), en(type), data(d));
    },
    texParameteri: function(target, pname, v)
    {
      const t = boundTex();
      if (t) t.params[en(pname)] = en(v);
      rec('texParameteri', t, en(pname), en(v));
    },
    generateMipmap: function() { rec('generateMipmap', boundTex()); },
    framebufferTexture2D: function(target, att, textarget, t, level)
    {
      if (st.fbo) st.fbo.attachments[en(att)] = t;
      rec('framebufferTexture2D', st.fbo, en(att), t);
    },
    // ---- uniforms (values as the GL stores them: fp32 / int32)
    uniform1i: function(l, v) { setUniform('uniform1i', l, typeof v === 'boolean' ? (v ? 1 : 0) : v | 0); },
    uniform1f: function(l, v) { setUniform('uniform1f', l, f32(v)); },
    uniform2f: function(l, a, b) { setUniform('uniform2f', l, [f32(a), f32(b)]); },
    uniform4f: function(l, a, b, c, d) { setUniform('uniform4f', l, [f32(a), f32(b), f32(c), f32(d)]); },
    uniform4fv: function(l, v) { setUniform('uniform4fv', l, Array.from(v, f32)); },
    uniform1fv: function(l, v) { setUniform('uniform1fv', l, Array.from(v, f32)); },
    // ---- draws / reads
    drawArrays: function(mode, first, count) { rec('drawArrays', prim(mode), first, count); },
    readPixels: function(x, y, w, h, fmt, type, dst)
    {
      rec('readPixels', x, y, w, h, en(fmt), en(type));
      if (opts.onReadPixels) opts.onReadPixels(dst, st);
    },
    getBufferSubData: function(target, off, dst) { rec('getBufferSubData', en(target), off); },
  };
  for (const k of Object.keys(ENUM)) gl[k] = ENUM[k];
  return new Proxy(gl, {
    get: function(t, p)
    {
      if (p in t || typeof p === 'symbol') return t[p];
      throw new Error('mockgl: gl.' + String(p) + ' is not modelled');
    }
  });
}

function tok(o) { return o === null || o === undefined ? null : o.__id !== undefined ? (o.__name || o.__id) : o; }
function resolve(v)
{
  if (v === null || v === undefined) return null;
  if (v.__uniform) return tok(v.__uniform.program) + '.' + v.__uniform.name;
  if (v.__id !== undefined) return tok(v);
  if (Array.isArray(v)) return v.map(resolve);
  if (typeof v == 'object' && !ArrayBuffer.isView(v)) {
    const o = {};
    for (const k of Object.keys(v)) o[k] = resolve(v[k]);
    return o;
  }
  return v;
}
// the recorded calls [a, b) with GL objects replaced by their tokens (reference names where the caller assigned them)
function trace(gl, a, b) { return gl.__state.trace.slice(a || 0, b === undefined ? gl.__state.trace.length : b).map(resolve); }
// uniform values per program, as the GL holds them
function uniforms(gl)
{
  const out = {};
  for (const id of Object.keys(gl.__state.objects)) {
    const o = gl.__state.objects[id];
    if (o.__kind != 'program' || !o.uniforms) continue;
    const u = (out[tok(o)] = {});
    for (const n of Object.keys(o.uniforms)) u[n] = o.uniforms[n].value;
  }
  return out;
}

// tables a comparison needs, with object tokens instead of object references
function summary(gl)
{
  const st = gl.__state, out = {textures: {}, framebuffers: {}, vaos: {}, tfs: {}, buffers: {}, programs: {}};
  for (const id of Object.keys(st.objects)) {
    const o = st.objects[id], name = o.__name || o.__id;
    if (o.__kind == 'texture') out.textures[name] = {storage: o.storage || null, params: o.params};
    else if (o.__kind == 'framebuffer') out.framebuffers[name] = resolve(o.attachments);
    else if (o.__kind == 'vao') out.vaos[name] = {enabled: o.enabled.slice().sort(), attribs: resolve(o.attribs)};
    else if (o.__kind == 'tf') out.tfs[name] = resolve(o.buffers);
    else if (o.__kind == 'buffer') out.buffers[name] = {data: o.data === undefined ? null : o.data};
    else if (o.__kind == 'program') out.programs[name] = {tfVaryings: o.tfVaryings || null};
  }
  return out;
}

module.exports = {create: create, summary: summary, trace: trace, uniforms: uniforms, ENUM: ENUM};
