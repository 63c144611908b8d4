'use strict';
// Developer tool: cube.drillUp of the SHARDED dimension from Node (setDevices; on a one-GPU box the device is
// named several times and the shards exchange directly), against the same cube on one device.
const olap = require('../olap-in-memory_amd/js');
const { Cube, GenericDimension } = olap;

function build(devices, shape) {
  olap.setDevices(devices);
  const dims = shape.map((n, d) => new GenericDimension(`dimension${d}`, 'root', Array.from({ length: n }, (_, i) => `dimension${d}-item${i}`)));
  const cube = new Cube(dims);
  cube.createStoredMeasure('measure0', {}, 'float32', 0);
  const n = shape.reduce((a, b) => a * b, 1);
  const data = new Float32Array(n);
  for (let i = 0; i < n; ++i) data[i] = 0.5 + (i % 97) / 97;
  cube.setData('measure0', data);
  olap.setDevices(null);
  return cube;
}

function time(label, fn, reps = 200) {
  for (let i = 0; i < 10; ++i) fn();
  const t0 = process.hrtime.bigint();
  let last;
  for (let i = 0; i < reps; ++i) last = fn();
  last.getSingleData ? null : null;
  const v = last.storedMeasures.measure0.getValue(0); // closes the loop: waits for the device
  const us = Number(process.hrtime.bigint() - t0) / 1e3 / reps;
  console.log(`${label.padEnd(64)} ${us.toFixed(1).padStart(9)} us   (cell 0 = ${v})`);
}

const shape = [320, 50, 625]; // 10^7 cells
const devices = (process.env.OLAP_BENCH_DEVICES || '0,0').split(',').map(Number);
const plain = build(null, shape);
const sharded = build(devices, shape);
console.log(`shape [${shape}] sharded over devices [${devices}] (${sharded.storedMeasures.measure0._native.isSharded ? 'sharded' : 'NOT sharded'})`);
time('one device:  cube.drillUp(dimension0, all)', () => plain.drillUp('dimension0', 'all'));
time('sharded:     cube.drillUp(dimension0, all)   partial + collective', () => sharded.drillUp('dimension0', 'all'));
time('one device:  cube.drillUp(dimension2, all)', () => plain.drillUp('dimension2', 'all'));
time('sharded:     cube.drillUp(dimension2, all)   per shard', () => sharded.drillUp('dimension2', 'all'));
