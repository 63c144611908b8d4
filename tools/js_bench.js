'use strict';
// Developer tool: end-to-end wall time of Cube operations from Node (N-API + allocation + kernel + sync).
const { Cube, GenericDimension } = require('../olap-in-memory_amd/js');

function cubeOf(nDims, size) {
  const dims = [];
  for (let i = 0; i < nDims; ++i) dims.push(new GenericDimension(`dimension${i}`, 'root', Array.from({ length: size }, (_x, j) => `dimension${i}-item${j}`)));
  const cube = new Cube(dims);
  cube.createStoredMeasure('measure0', {}, 'float32', 0);
  return cube;
}

// Operations are enqueued on the device stream and not waited for, so the loop is closed by a
// one-cell read of the last result (a blocking copy behind everything queued): the figure is the
// sustained time per operation, not just the enqueue time.
function time(label, fn, iters) {
  const wait = (r) => (r && r.storedMeasures ? Object.values(r.storedMeasures)[0].getValue(0) : r);
  for (let i = 0; i < 3; ++i) wait(fn());
  const t0 = process.hrtime.bigint();
  let last;
  for (let i = 0; i < iters; ++i) last = fn();
  wait(last);
  const us = Number(process.hrtime.bigint() - t0) / 1e3 / iters;
  console.log(`${label.padEnd(58)} ${us.toFixed(1).padStart(10)} us`);
  return us;
}

const big = cubeOf(8, 10);
{
  const values = new Float32Array(1e8);
  let s = 20240807;
  for (let i = 0; i < values.length; ++i) {
    s = (Math.imul(s, 1664525) + 1013904223) | 0;
    values[i] = 0.5 + (s >>> 8) / 16777216;
  }
  for (const label of ['first call, cold runtime', 'second call']) {
    const t0 = process.hrtime.bigint();
    big.setData('measure0', values);
    console.log(`setData(1e8 Float32Array), ${label}: ${(Number(process.hrtime.bigint() - t0) / 1e6).toFixed(1)} ms`);
  }
}
time('[10]^8 drillUp(dimension0, all)', () => big.drillUp('dimension0', 'all'), 50);
time('[10]^8 drillUp(dimension7, all)', () => big.drillUp('dimension7', 'all'), 50);
time('[10]^8 slice(dimension1, root, item3)', () => big.slice('dimension1', 'root', 'dimension1-item3'), 50);
time('[10]^8 slice -> dice(3 of 10) -> drillUp (config 3 chain)', () => big.slice('dimension1', 'root', 'dimension1-item3').dice('dimension4', 'root', ['dimension4-item1', 'dimension4-item4', 'dimension4-item7']).drillUp('dimension0', 'all'), 50);
time('[10]^8 removeDimension(dimension4)', () => big.removeDimension('dimension4'), 50);
const collapsed = big.collapse();
console.log(`[10]^8 collapse() -> ${collapsed.getData('measure0')[0]}`);
time('[10]^8 collapse()  (additive rules: one float64 total)', () => big.collapse(), 20);
const small = cubeOf(10, 4); // the reference benchmark's cube: 4^10 cells
small.fillData('measure0', 1);
time('4^10 slice(dimension0, all, all)   [test/cube-benchmark.js:38]', () => small.slice('dimension0', 'all', 'all'), 200);
time('4^10 dice(dimension2, 2 of 4)      [test/cube-benchmark.js:83]', () => small.dice('dimension2', 'root', ['dimension2-item2', 'dimension2-item3']).getTotal('measure0'), 50);
time('4^10 dice(...).getData() -> plain Array of 524288 numbers     ', () => small.dice('dimension2', 'root', ['dimension2-item2', 'dimension2-item3']).getData('measure0').length, 20);
time('4^10 collapse()                    [test/cube-benchmark.js:59]', () => small.collapse(), 50);
time('4^10 reorderDimensions(reverse)    [test/cube-benchmark.js:71]', () => small.reorderDimensions(small.dimensionIds.slice().reverse()), 50);

// four stored measures with four rules on a 10^6-cell cube: Cube.drillUp hands them to the device together
// (HipStore.drillUpMany -> olap_store_drillup_multi: one mixed-rule launch); per-measure store calls for comparison
{
  const dims = [];
  for (let i = 0; i < 6; ++i) dims.push(new GenericDimension(`dimension${i}`, 'root', Array.from({ length: 10 }, (_x, j) => `dimension${i}-item${j}`)));
  const multi = new Cube(dims);
  const rules = ['sum', 'average', 'highest', 'lowest'];
  rules.forEach((rule, m) => {
    multi.createStoredMeasure(`measure${m}`, Object.fromEntries(dims.map((d) => [d.id, rule])), 'float32', 0);
    multi.fillData(`measure${m}`, m + 1);
  });
  for (const dim of ['dimension0', 'dimension3']) {
    time(`10^6 cells x 4 measures / 4 rules: cube.drillUp(${dim}, all)`, () => multi.drillUp(dim, 'all'), 200);
    const rolled = multi.drillUp(dim, 'all');
    time(`   the same as 4 store.drillUp calls (one launch each)`, () => {
      let last;
      rules.forEach((rule, m) => {
        last = multi.storedMeasures[`measure${m}`].drillUp(multi.dimensions, rolled.dimensions, rule);
      });
      return last.getValue(0);
    }, 200);
  }
}
