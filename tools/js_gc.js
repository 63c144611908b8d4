'use strict';
// Developer tool: device memory held by Store wrappers across a long synchronous loop.
const { Cube, GenericDimension } = require('../olap-in-memory_amd/js');
const backend = require('../olap-in-memory_amd/js/backend').load();
const gb = () => (backend.heldBytes() / 1e9).toFixed(1);
const dims = [];
for (let i = 0; i < 8; ++i) dims.push(new GenericDimension(`d${i}`, 'root', Array.from({ length: 10 }, (_x, j) => `i${j}`)));
const cube = new Cube(dims);
cube.createStoredMeasure('mm', {}, 'float32', 0);
cube.fillData('mm', 1);
let peak = 0;
const t0 = process.hrtime.bigint();
const N = 4000;
for (let i = 0; i < N; ++i) {
  const r = cube.drillUp('d0', 'all');
  if (i % 500 === 0) console.log(`iteration ${i}: ${gb()} GB held`);
  peak = Math.max(peak, backend.heldBytes());
  if (i === N - 1) console.log('last result cell:', r.getData('mm')[0]);
}
console.log(`${N} x cube.drillUp (40 MB results, ${(N * 0.04).toFixed(0)} GB in total): ${(Number(process.hrtime.bigint() - t0) / 1e3 / N).toFixed(1)} us each, peak ${(peak / 1e9).toFixed(1)} GB held`);
