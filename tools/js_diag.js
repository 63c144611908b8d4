'use strict';
const { Cube, GenericDimension } = require('../olap-in-memory_amd/js');
const backend = require('../olap-in-memory_amd/js/backend').load();
function t(label, fn, iters) {
  for (let i = 0; i < 3; ++i) fn();
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < iters; ++i) fn();
  console.log(label.padEnd(50), (Number(process.hrtime.bigint() - t0) / 1e3 / iters).toFixed(1), 'us');
}
const dims = [];
for (let i = 0; i < 8; ++i) dims.push(new GenericDimension(`d${i}`, 'root', Array.from({ length: 10 }, (_x, j) => `i${j}`)));
const cube = new Cube(dims);
cube.createStoredMeasure('mm', {}, 'float32', 0);
cube.fillData('mm', 1);
const store = cube.storedMeasures.mm;
const oldLen = Uint32Array.from(dims, () => 10);
const newLen = Uint32Array.from(dims, (_d, i) => (i === 0 ? 1 : 10));
const maps = dims.map((_d, i) => (i === 0 ? new Uint32Array(10) : Uint32Array.from({ length: 10 }, (_x, k) => k)));
t('native.drillUp only', () => store._native.drillUp(oldLen, newLen, maps, 0), 50);
t('native.countSet (sync + tiny kernel)', () => store._native.countSet(), 50);
t('native.getValue', () => store._native.getValue(5), 200);
t('dimension.drillUp only', () => dims[0].drillUp('all'), 200);
t('cube.drillUp', () => cube.drillUp('d0', 'all'), 50);
const small = new backend.Store(1000, 2, 0);
const ol = Uint32Array.from([10, 100]), nl = Uint32Array.from([1, 100]);
const mp = [new Uint32Array(10), Uint32Array.from({ length: 100 }, (_x, k) => k)];
t('native.drillUp on 1000 cells', () => small.drillUp(ol, nl, mp, 0), 200);
t('new Store(1000)', () => new backend.Store(1000, 2, 0), 200);
t('new Store(1e7)', () => new backend.Store(1e7, 2, 0), 50);
const newDims = dims.slice();
newDims[0] = dims[0].drillUp('all');
t('HipStore.drillUp (JS wrapper)', () => store.drillUp(dims, newDims, 'sum'), 50);
t('maps build only', () => newDims.map((dim, i) => Uint32Array.from(dims[i].getGroupIndexFromRootIndexMap(dim.rootAttribute))), 200);
t('methodFromName', () => backend.methodFromName('sum'), 200);
t('lengthsOf', () => Uint32Array.from(dims, (d) => d.numItems), 200);
const HipStore = require('../olap-in-memory_amd/js/store/hip');
const nat = store._native.drillUp(oldLen, newLen, maps, 0);
t('new HipStore(wrap)', () => new HipStore(nat.size, 'float32', 0, nat), 200);
t('nat.size', () => nat.size, 200);
// memory stays bounded: 3000 results of 40 MB = 120 GB if nothing were reclaimed
{
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < 3000; ++i) cube.drillUp('d0', 'all');
  console.log('3000 x cube.drillUp:', (Number(process.hrtime.bigint() - t0) / 1e3 / 3000).toFixed(1), 'us each');
}
