'use strict';
/*
 * The Node.js host on a sharded cube: setDevices([0, 0, 0]) splits every stored measure along the cube's
 * outermost dimension (three shards on the one GPU of the test box, direct exchange).  Every query is run
 * on the sharded cube and on an identical single-device cube; results must be identical cell for cell
 * (float sums of exactly representable values, so the order of addition does not matter).
 */
const { describe, it, assert, run } = require('./harness');
const olap = require('../../olap-in-memory_amd/js');
const { Cube, GenericDimension } = olap;

function mulberry32(a) {
  return function () {
    a |= 0; a = (a + 0x6d2b79f5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

function build(sharded) {
  olap.setDevices(sharded ? [0, 0, 0] : null);
  const dims = [14, 6, 10].map((n, d) => {
    const items = Array.from({ length: n }, (_, i) => `dimension${d}-item${i}`);
    const dim = new GenericDimension(`dimension${d}`, 'root', items);
    const parent = {};
    items.forEach((item, i) => { parent[item] = `group${i % 3}`; });
    dim.addAttribute('root', 'bucket', parent);
    return dim;
  });
  const cube = new Cube(dims);
  const rnd = mulberry32(42);
  for (const [id, rule, type, def] of [['m_sum', 'sum', 'float32', 0], ['m_avg', 'average', 'float32', NaN], ['m_last', 'last', 'int32', 0], ['m_high', 'highest', 'float64', NaN]]) {
    cube.createStoredMeasure(id, { dimension0: rule, dimension1: rule, dimension2: rule }, type, def);
    const data = Array.from({ length: 840 }, () => (rnd() < 0.7 ? Math.round(rnd() * 64 - 32) / 4 : (Number.isNaN(def) ? null : 0)));
    cube.setData(id, type === 'int32' ? data.map((v) => (v === null ? v : Math.round(v))) : data);
  }
  olap.setDevices(null);
  return cube;
}

const plain = build(false);
const sharded = build(true);
// A measure with a first / last rule tracks the reference Map's insertion order (tests/js/gpu_test.js,
// tests/test_insertion_order.py) and therefore stays on ONE device whatever the device list says: `m_last` is the same
// kind of store in both cubes, and the same program answers the same with and without setDevices().
const MEASURES = ['m_sum', 'm_avg', 'm_last', 'm_high'];
const SPLIT = ['m_sum', 'm_avg', 'm_high'];

// Every rank ships its float64 accumulators and the sum is rounded ONCE, as on one device; the cells here are
// multiples of 1/4, whose float64 sums are exact in any order: identical cell for cell, every measure.
function same(a, b, what) {
  for (const m of MEASURES) assert.deepEqual(a.getData(m), b.getData(m), `${what} ${m}`);
  assert.deepEqual(a.dimensionIds, b.dimensionIds);
}

describe('sharded cube', () => {
  it('measures really are sharded, three shards of 5 + 5 + 4 rows', () => {
    for (const m of SPLIT) {
      assert.ok(sharded.storedMeasures[m]._native.isSharded, m);
      assert.ok(!plain.storedMeasures[m]._native.isSharded, m);
      assert.deepEqual(Array.from(sharded.storedMeasures[m]._native.bounds), [0, 5, 10, 14]);
    }
    same(sharded, plain, 'data');
  });
  it('a measure with a first / last rule keeps the insertion order and stays on one device', () => {
    assert.ok(!sharded.storedMeasures.m_last._native.isSharded && sharded.storedMeasures.m_last.orderTracked);
    assert.ok(plain.storedMeasures.m_last.orderTracked);
    // out-of-order writes, then a roll-up of the (would-be sharded) dimension: `last` = the LAST INSERTED member, with
    // and without a device list alike (reference: in-memory.js:298 iterates the Map)
    for (const devices of [[0, 0, 0], null]) {
      olap.setDevices(devices);
      const dim = new GenericDimension('rows', 'root', ['r0', 'r1', 'r2', 'r3', 'r4', 'r5']);
      const c = new Cube([dim]);
      c.createStoredMeasure('seen', { rows: 'last' }, 'float32', 0);
      c.createStoredMeasure('total', { rows: 'sum' }, 'float32', 0);
      olap.setDevices(null);
      for (const [item, v] of [['r4', 40], ['r1', 10], ['r5', 50], ['r2', 20]]) {
        c.setSingleData('seen', { rows: item }, v);
        c.setSingleData('total', { rows: item }, v);
      }
      assert.equal(!!c.storedMeasures.total._native.isSharded, devices !== null);
      assert.ok(!c.storedMeasures.seen._native.isSharded);
      assert.deepEqual(c.drillUp('rows', 'all').getData('seen'), [20]); // r2 was written last
      assert.deepEqual(c.drillUp('rows', 'all').getData('total'), [120]);
      assert.deepEqual(Array.from(c.getStatusMap('seen').keys()), [4, 1, 5, 2]);
    }
  });
  it("drillUp of the sharded dimension to 'all' and to an attribute: partial + one collective", () => {
    const a = sharded.drillUp('dimension0', 'all');
    assert.ok(!a.storedMeasures.m_sum._native.isSharded); // K0 times smaller: arrives whole on the first device
    same(a, plain.drillUp('dimension0', 'all'), 'dim0 -> all');
    same(sharded.drillUp('dimension0', 'bucket'), plain.drillUp('dimension0', 'bucket'), 'dim0 -> bucket');
    same(sharded.removeDimension('dimension0'), plain.removeDimension('dimension0'), 'removeDimension(dimension0)');
  });
  it('float64 partials: cancellation across shards gives the one-device (and reference) answer', () => {
    // VERDICT r02: [2^24, 1 | -2^24, unset] rolled up over two shards gave 0 / unset with Float32 partials
    for (const devices of [[0, 0], [0, 0, 0], null]) {
      olap.setDevices(devices);
      const c = new Cube([new GenericDimension('rows', 'root', ['r0', 'r1', 'r2', 'r3'])]);
      c.createStoredMeasure('m_sum', { rows: 'sum' }, 'float32', 0);
      c.createStoredMeasure('m_avg', { rows: 'average' }, 'float32', 0);
      olap.setDevices(null);
      for (const m of ['m_sum', 'm_avg']) c.setData(m, [16777216, 1, -16777216, 0]);
      assert.equal(!!c.storedMeasures.m_sum._native.isSharded, devices !== null);
      const all = c.drillUp('rows', 'all');
      assert.deepEqual(all.getData('m_sum'), [1]);
      assert.deepEqual(all.getData('m_avg'), [Math.fround(1 / 3)]);
      assert.deepEqual(Array.from(all.getStatusMap('m_sum').keys()), [0]);
    }
  });
  it('operations on other dimensions stay sharded', () => {
    const u = sharded.drillUp('dimension2', 'bucket');
    assert.ok(u.storedMeasures.m_sum._native.isSharded);
    same(u, plain.drillUp('dimension2', 'bucket'), 'dim2 -> bucket');
    same(u.drillUp('dimension0', 'all'), plain.drillUp('dimension2', 'bucket').drillUp('dimension0', 'all'), 'dim2 -> bucket, dim0 -> all');
    const d = sharded.dice('dimension1', 'root', ['dimension1-item4', 'dimension1-item1']);
    assert.ok(d.storedMeasures.m_sum._native.isSharded);
    same(d, plain.dice('dimension1', 'root', ['dimension1-item4', 'dimension1-item1']), 'dice dim1');
    same(sharded.slice('dimension2', 'root', 'dimension2-item7'), plain.slice('dimension2', 'root', 'dimension2-item7'), 'slice dim2');
  });
  it('dice of the sharded dimension: ascending rows stay sharded, anything else is gathered', () => {
    const rows = ['dimension0-item1', 'dimension0-item2', 'dimension0-item7', 'dimension0-item13'];
    const d = sharded.dice('dimension0', 'root', rows);
    assert.ok(d.storedMeasures.m_sum._native.isSharded);
    assert.deepEqual(Array.from(d.storedMeasures.m_sum._native.bounds), [0, 2, 3, 4]);
    same(d, plain.dice('dimension0', 'root', rows), 'dice rows');
    same(d.drillUp('dimension0', 'all'), plain.dice('dimension0', 'root', rows).drillUp('dimension0', 'all'), 'dice rows then all');
    const rev = rows.slice().reverse();
    same(sharded.dice('dimension0', 'root', rev, true), plain.dice('dimension0', 'root', rev, true), 'dice reordered rows');
    same(sharded.slice('dimension0', 'root', 'dimension0-item9'), plain.slice('dimension0', 'root', 'dimension0-item9'), 'slice dim0');
  });
  it('reorderDimensions: the sharded dimension stays in front or the cube is gathered', () => {
    same(sharded.reorderDimensions(['dimension0', 'dimension2', 'dimension1']), plain.reorderDimensions(['dimension0', 'dimension2', 'dimension1']), 'swap 1, 2');
    same(sharded.reorderDimensions(['dimension2', 'dimension1', 'dimension0']), plain.reorderDimensions(['dimension2', 'dimension1', 'dimension0']), 'reverse');
  });
  it('accessors, totals, nested objects, single cells', () => {
    for (const m of MEASURES) {
      assert.equal(sharded.getTotal(m), plain.getTotal(m));
      assert.deepEqual(sharded.getNestedObject(m, true), plain.getNestedObject(m, true));
      assert.deepEqual(Array.from(sharded.getStatusMap(m).keys()), Array.from(plain.getStatusMap(m).keys()));
    }
    const where = { dimension0: 'dimension0-item11', dimension1: 'dimension1-item2', dimension2: 'dimension2-item5' };
    assert.equal(sharded.getSingleData('m_sum', where), plain.getSingleData('m_sum', where));
    const c = sharded.clone();
    c.setSingleData('m_sum', where, 99.5);
    assert.equal(c.getSingleData('m_sum', where), 99.5);
    assert.equal(sharded.getSingleData('m_sum', where), plain.getSingleData('m_sum', where)); // the clone is independent
  });
  it('serialize / deserialize and hydrateFromCube', () => {
    const back = Cube.deserialize(sharded.serialize());
    same(back, plain, 'round trip');
    const target = build(true);
    target.setData('m_sum', new Array(840).fill(0));
    target.hydrateFromCube(plain.dice('dimension1', 'root', ['dimension1-item0', 'dimension1-item3']));
    const expect = build(false);
    expect.setData('m_sum', new Array(840).fill(0));
    expect.hydrateFromCube(plain.dice('dimension1', 'root', ['dimension1-item0', 'dimension1-item3']));
    same(target, expect, 'hydrateFromCube');
  });
  it('computed measures over sharded measures', () => {
    for (const c of [sharded, plain]) {
      if (!c.computedMeasureIds.includes('ratio')) c.createComputedMeasure('ratio', 'm_sum * 2 + m_last');
    }
    assert.deepEqual(sharded.getData('ratio'), plain.getData('ratio'));
  });
});

run();
