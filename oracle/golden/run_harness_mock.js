#!/usr/bin/env node
/*
 * Runs oracle/golden/harness.js -- the script that drives the reference's shaders under HeadlessChrome + SwiftShader for every golden
 * fixture -- against the recording mock of the WebGL2 context (mockgl.js) instead of a browser (TEST INFRASTRUCTURE).
 *
 * Purpose: harness.js restates the reference's host loop (app.js:5830-6005), texture / framebuffer set-up and uniform pushes by hand.
 * gen_host_golden.js records what app.js ITSELF does with a GL context; this runner records what harness.js does with the same mock,
 * so that tests/test_host_golden.py can require the two to be the same sequence of GL calls on the same object graph.
 * No reference file is needed: shader sources are stand-ins carrying only their file name (the mock compiles nothing).
 *
 * usage: node run_harness_mock.js job.json   (job = what gen_golden.py puts into layout.wx, minus `dir`)  -> JSON on stdout
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const mockgl = require('./mockgl.js');

const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const X = job.X, Y = job.Y, N = job.n_drops || 0;
const sizes = {'base.f32': X * Y * 16, 'water.f32': X * Y * 16, 'wall.i8': X * Y * 4, 'drops.f32': Math.max(N, 1) * 20};

const gl = mockgl.create({
  canvas: {width: X, height: Y},
  onReadPixels: function(dst) { if (dst && dst.length == 4) { dst[0] = job.readback0 === undefined ? 777 : job.readback0; dst[2] = job.readback2 || 0; dst[3] = 0.5; } },
});
function XHR() { this.responseText = null; }
XHR.prototype.open = function(m, url) { this.url = url; };
XHR.prototype.overrideMimeType = function() {};
XHR.prototype.send = function()
{
  const u = this.url, name = u.split('/').pop();
  if (u.indexOf('/shaders/') >= 0) this.responseText = name == 'common.glsl' ? '// common.glsl stand-in\n' : '//@@file:' + name + '\n#include "common.glsl"\nuniform vec4 a[126];\nvoid main() {}\n';
  else if (sizes[name] !== undefined) this.responseText = '\0'.repeat(sizes[name]);
  else throw new Error('run_harness_mock: unexpected URL ' + u);
};
const marks = [];
job.dir = 'mock://fixture/';
job.mark = function(what, it) { marks.push([what, it, gl.__state.trace.length]); };
const windowObj = {};
const sandbox = {
  window: windowObj, XMLHttpRequest: XHR, performance: {now: function() { return 0; }},
  document: {createElement: function(tag) { if (tag != 'canvas') throw new Error(tag); return {width: 0, height: 0, getContext: function() { return gl; }}; }},
  btoa: function(s) { return Buffer.from(s, 'binary').toString('base64'); },
  Float32Array: Float32Array, Int8Array: Int8Array, Int32Array: Int32Array, Uint8Array: Uint8Array, Math: Math, JSON: JSON, String: String, Promise: Promise,
};
vm.runInNewContext(fs.readFileSync(path.join(__dirname, 'harness.js'), 'utf8'), sandbox, {filename: 'harness.js'});
windowObj.Plotly.toImage({layout: {wx: job}}, {}).then(function(txt) {
  const res = JSON.parse(txt);
  if (res.error) throw new Error(res.error);
  const iters = [];
  for (let i = 0; i + 1 < marks.length; i += 2) iters.push(mockgl.trace(gl, marks[i][2], marks[i + 1][2]));
  const first = marks.length ? marks[0][2] : gl.__state.trace.length;
  process.stdout.write(JSON.stringify({setup_trace: mockgl.trace(gl, 0, first), iterations: iters, tables: mockgl.summary(gl), uniforms: mockgl.uniforms(gl),
                                       errors: gl.__state.errors, inactiveDroplets: res.inactiveDroplets === undefined ? null : res.inactiveDroplets}));
}).catch(function(e) { console.error(e && e.stack ? e.stack : e); process.exit(1); });
