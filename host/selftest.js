// CPU-side self test of the JS host logic (no GPU): codec round trip, settings merge, derived parameters.
'use strict';
const assert = require('assert');
const H = require('./sim_host.js');
const X = 20, Y = 10, n = X * Y * 4;
const base = new Float32Array(n).map(function(_, i) { return Math.fround(Math.sin(i)); });
const water = new Float32Array(n).map(function(_, i) { return Math.fround(i * 0.25); });
const wall = new Int8Array(n).map(function(_, i) { return (i * 7) % 256 - 128; });
const drops = new Float32Array(Math.floor(X * Y / 25) * 5).map(function(_, i) { return -i; });
const sf = {X: X, Y: Y, base: base, water: water, wall: wall, droplets: drops, stations: [[3, 4]], settings: {vorticity: 0.007, waterWeight: -1}};
const sf2 = H.decodeSave(H.encodeSave(sf));
assert.strictEqual(sf2.X, X); assert.strictEqual(sf2.Y, Y);
assert.deepStrictEqual(Array.from(sf2.base), Array.from(base)); assert.deepStrictEqual(Array.from(sf2.wall), Array.from(wall));
assert.deepStrictEqual(Array.from(sf2.droplets), Array.from(drops)); assert.deepStrictEqual(sf2.stations, [[3, 4]]);
const gui = H.mergeSettings(sf2.settings);
assert.strictEqual(gui.vorticity, 0.007); assert.strictEqual(gui.waterWeight, 0.25); assert.strictEqual(gui.dynamicWaterTemperature, false);
assert.strictEqual(gui.condensationRate, 0.005);
const u = H.uniformsFromGui(H.mergeSettings({sunAngle: 67.45275198770811, sunIntensity: 1}), 100);
assert.strictEqual(u.dryLapse, 120); assert.ok(Math.abs(u.sunAngle + 0.39352388) < 1e-7); assert.ok(Math.abs(u.sunIntensity - 1289.7039) < 1e-3);
assert.strictEqual(u.initial_T[0], Math.fround(288.15)); assert.ok(Math.abs(u.initial_T[100] - 333.15) < 1e-4);
// the clock of startSimulation() for a few (month, timeOfDay, dayNightCycle) settings, as [month index, date, h, min, s]
const clocks = [[6.65, 9.9, true], [6.65, 9.9, false], [6.67, 11.44416, true], [1.0, 0.0, true], [12.99, 23.99, true], [13.016, 5.5, true], [3.5, 12.25, false]]
  .map(function(c) { const t = H.initialSimDateTime(c[0], c[1], c[2]); return [t.getFullYear(), t.getMonth(), t.getDate(), t.getHours(), t.getMinutes(), t.getSeconds()]; });
console.log(JSON.stringify({ok: true, clocks: clocks, initial_T: Array.from(u.initial_T.slice(0, 4)), sunAngle: u.sunAngle, sunIntensity: u.sunIntensity}));
