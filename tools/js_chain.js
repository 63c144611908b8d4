'use strict';
// Developer tool: where the host time of the config-3 chain goes (enqueue-only vs closed by a read).
const { Cube, GenericDimension } = require('../olap-in-memory_amd/js');
const dims = [];
for (let i = 0; i < 8; ++i) dims.push(new GenericDimension(`dimension${i}`, 'root', Array.from({ length: 10 }, (_x, j) => `dimension${i}-item${j}`)));
const big = new Cube(dims);
big.createStoredMeasure('measure0', {}, 'float32', 0);
big.fillData('measure0', 1);
const t = (label, fn, n = Number(process.env.N || 200)) => {
  for (let i = 0; i < 5; ++i) fn();
  const t0 = process.hrtime.bigint();
  let last;
  for (let i = 0; i < n; ++i) last = fn();
  const us = Number(process.hrtime.bigint() - t0) / 1e3 / n;
  console.log(`${label.padEnd(60)} ${us.toFixed(1).padStart(8)} us`);
  return last;
};
const store = big.storedMeasures.measure0;
t('getValue(0) on a resident store (blocking 1-cell read)', () => store.getValue(0));
t('slice (lazy, host only)', () => big.slice('dimension1', 'root', 'dimension1-item3'));
const sliced = big.slice('dimension1', 'root', 'dimension1-item3');
t('dice on the sliced cube (lazy, host only)', () => sliced.dice('dimension4', 'root', ['dimension4-item1', 'dimension4-item4', 'dimension4-item7']));
const diced = sliced.dice('dimension4', 'root', ['dimension4-item1', 'dimension4-item4', 'dimension4-item7']);
t('drillUp of the diced cube (one launch, enqueue only)', () => diced.drillUp('dimension0', 'all'));
t('whole chain, enqueue only', () => big.slice('dimension1', 'root', 'dimension1-item3').dice('dimension4', 'root', ['dimension4-item1', 'dimension4-item4', 'dimension4-item7']).drillUp('dimension0', 'all'));
t('whole chain + 1-cell read', () => big.slice('dimension1', 'root', 'dimension1-item3').dice('dimension4', 'root', ['dimension4-item1', 'dimension4-item4', 'dimension4-item7']).drillUp('dimension0', 'all').storedMeasures.measure0.getValue(0));
t('drillUp(dimension0) of the full cube, enqueue only', () => big.drillUp('dimension0', 'all'), 50);
// raw addon call with prebuilt arguments (what is left is N-API marshalling + plan cache + pool + launch)
{
  const lens = Uint32Array.from({ length: 8 }, () => 10);
  const newLens = Uint32Array.from(lens);
  newLens[0] = 1;
  const maps = Array.from({ length: 8 }, (_x, d) => (d === 0 ? new Uint32Array(10) : Uint32Array.from({ length: 10 }, (_y, i) => i)));
  const native = store._native;
  t('addon Store.drillUp with prebuilt arguments, enqueue only', () => native.drillUp(lens, newLens, maps, 0), 50);
  const small = new Cube(dims.slice(0, 4));
  small.createStoredMeasure('measure0', {}, 'float32', 0);
  small.fillData('measure0', 1);
  t('10^4-cell cube: cube.drillUp(dimension0, all), enqueue only', () => small.drillUp('dimension0', 'all'));
  const sn = small.storedMeasures.measure0._native;
  const l4 = Uint32Array.from({ length: 4 }, () => 10);
  const n4 = Uint32Array.from(l4);
  n4[0] = 1;
  t('10^4-cell cube: addon Store.drillUp, enqueue only', () => sn.drillUp(l4, n4, maps.slice(0, 4), 0));
  t('10^4-cell cube: cube.drillUp + read of one cell', () => small.drillUp('dimension0', 'all').storedMeasures.measure0.getValue(0));
}
