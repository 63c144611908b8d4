'use strict';
// The reference's benchmark scenarios (test/cube-benchmark.js:35-128: slice, collapse, reorder, dice,
// addDimension, removeDimension, compose on 4^10-cell cubes with 3 float32 measures at 100 / 50 / 25 /
// 10 % fill, mean of 10 runs) through this repository's Cube on the GPU.  The reference prints
// milliseconds and asserts nothing; its own figures for the dense cube, measured in the survey
// container (BASELINE.md §2), are printed beside ours.  Every run is closed by reading one cell of
// every measure of the result, so queued device work is included.
const { Cube, GenericDimension } = require('../olap-in-memory_amd/js');

function largeCube(nDims, size, nMeasures, fill, firstMeasure = 0) {
  const dims = [];
  for (let i = 0; i < nDims; ++i) dims.push(new GenericDimension(`dimension${i}`, 'root', Array.from({ length: size }, (_x, j) => `dimension${i}-item${j}`)));
  const cube = new Cube(dims);
  const n = cube.storeSize;
  const values = new Float32Array(n);
  // test/helpers/create-large-test-cube.js:31-46 marks `fill * n` random cells with 1; seeded here
  let s = 20240807;
  let left = Math.round(fill * n);
  if (fill >= 1) values.fill(1);
  else
    while (left > 0) {
      s = (Math.imul(s, 1664525) + 1013904223) | 0;
      const i = (s >>> 0) % n;
      if (values[i] === 0) {
        values[i] = 1;
        --left;
      }
    }
  for (let m = firstMeasure; m < firstMeasure + nMeasures; ++m) {
    cube.createStoredMeasure(`measure${m}`, {}, 'float32', 0);
    cube.setData(`measure${m}`, values);
  }
  return cube;
}

const finish = (cube) => {
  for (const id of cube.storedMeasureIds) cube.storedMeasures[id].getValue(0);
};
function meanMs(fn, times = 10) {
  finish(fn());
  let total = 0;
  for (let i = 0; i < times; ++i) {
    const t0 = process.hrtime.bigint();
    finish(fn());
    total += Number(process.hrtime.bigint() - t0) / 1e6;
  }
  return total / times;
}

const fills = [1.0, 0.5, 0.25, 0.1];
const cubes = fills.map((f, i) => largeCube(10, 4, 3, f, i === 3 ? 3 : 0));
const extra = new GenericDimension('dimension-new', 'root', Array.from({ length: 5 }, (_x, j) => `dimension-new-item${j}`));
const scenarios = [
  [':38  slice(dimension0, all, all)', 1236, (c) => c.slice('dimension0', 'all', 'all')],
  [':47  slice(dimension3, root, item2)', 1788, (c) => c.slice('dimension3', 'root', 'dimension3-item2')],
  [':59  collapse()', 1122, (c) => c.collapse()],
  [':71  reorderDimensions(reverse)', 3403, (c) => c.reorderDimensions(c.dimensionIds.slice().reverse())],
  [':83  dice(dimension2, [item2, item3])', 1616, (c) => c.dice('dimension2', 'root', ['dimension2-item2', 'dimension2-item3'])],
  [':100 addDimension(5 items)', null, (c) => c.addDimension(extra)],
  [':113 removeDimension(dimension4)', 1049, (c) => c.removeDimension('dimension4')],
];
console.log('scenario (test/cube-benchmark.js)            fill: 100%      50%      25%      10%   [ms]   reference, dense [ms]');
for (const [label, ref, fn] of scenarios) {
  const ms = cubes.map((c) => meanMs(() => fn(c)));
  console.log(`${label.padEnd(46)} ${ms.map((v) => v.toFixed(3).padStart(8)).join(' ')}          ${ref === null ? 'n/a' : String(ref)}`);
}
console.log(`${':122 compose(10 % cube, 50 % cube)'.padEnd(46)} ${meanMs(() => cubes[3].compose(cubes[1])).toFixed(3).padStart(8)}`);
