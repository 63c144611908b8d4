'use strict';
/*
 * Host-side logic that needs no GPU: dimensions (the index-map producers, SURVEY §8(a8)), the
 * calendar restatement, formatters and the errors raised before any device work.
 * Expected values are the literals the reference's own tests assert
 * (test/dimension-generic.js, test/dimension-time.js) plus independent Gregorian checks.
 */
const { describe, it, beforeEach, assert, run } = require('./harness');
const { GenericDimension, TimeDimension, TimeSlot, HipStore, Cube } = require('../../olap-in-memory_amd/js');
const fmt = require('../../olap-in-memory_amd/js/formatter');

describe('GenericDimension', () => {
  let dimension;
  beforeEach(() => {
    dimension = new GenericDimension('location', 'zipCode', ['75018', '75019', '59000', '59133', '31000']);
    dimension.addAttribute('zipCode', 'city', { 75018: 'paris', 75019: 'paris', 59000: 'lille', 59133: 'phalempin', 31000: 'toulouse' });
    dimension.addAttribute('city', 'region', (city) => ({ paris: 'idf', lille: 'hdf', phalempin: 'hdf', toulouse: 'occitanie' }[city]));
  });

  it('numbers groups by first appearance', () => {
    assert.deepEqual(dimension.getItems('city'), ['paris', 'lille', 'phalempin', 'toulouse']);
    assert.deepEqual(Array.from(dimension.getGroupIndexFromRootIndexMap('city')), [0, 0, 1, 2, 3]);
    assert.deepEqual(Array.from(dimension.getGroupIndexFromRootIndexMap('region')), [0, 0, 1, 1, 2]);
    assert.deepEqual(Array.from(dimension.getGroupIndexFromRootIndexMap('all')), [0, 0, 0, 0, 0]);
    assert.deepEqual(Array.from(dimension.getGroupIndexFromRootIndexMap('zipCode')), [0, 1, 2, 3, 4]);
    assert.ok(dimension.getGroupIndexFromRootIndexMap('city') instanceof Uint32Array);
  });

  it('lists attributes and sizes', () => {
    assert.sameMembers(dimension.attributes, ['all', 'zipCode', 'city', 'region']);
    assert.equal(dimension.numItems, 5);
    assert.equal(dimension.rootAttribute, 'zipCode');
    assert.equal(dimension.getGroupItemFromRootItem('region', '59133'), 'hdf');
    assert.equal(dimension.getRootIndexFromRootItem('nope'), -1);
  });

  it('drillUp keeps only attributes that are functions of the new root', () => {
    const cities = dimension.drillUp('city');
    assert.equal(cities.rootAttribute, 'city');
    assert.deepEqual(cities.getItems(), ['paris', 'lille', 'phalempin', 'toulouse']);
    assert.sameMembers(cities.attributes, ['all', 'city', 'region']); // zipCode is not a function of city
    assert.deepEqual(Array.from(cities.getGroupIndexFromRootIndexMap('region')), [0, 1, 1, 2]);
    assert.equal(dimension.drillUp('zipCode'), dimension);
    assert.throws(() => dimension.drillUp('planet'));
  });

  it('dice by root items and by group', () => {
    assert.equal(dimension.dice('zipCode', ['75018', '75019', '59000', '59133', '31000']), dimension);
    assert.deepEqual(dimension.dice('zipCode', ['59000', '75018']).getItems(), ['75018', '59000']);
    assert.deepEqual(dimension.dice('zipCode', ['59000', '75018'], true).getItems(), ['59000', '75018']);
    assert.deepEqual(dimension.dice('zipCode', ['nope', '31000']).getItems(), ['31000']);
    assert.deepEqual(dimension.dice('zipCode', []).getItems(), []);
    const hdf = dimension.dice('region', ['hdf']);
    assert.deepEqual(hdf.getItems(), ['59000', '59133']);
    assert.deepEqual(hdf.getItems('city'), ['lille', 'phalempin']);
    assert.throws(() => dimension.dice('region', ['hdf'], true), /Reordering is not allowed when using groups/);
  });

  it('rejects non-string mappings', () => {
    assert.throws(() => dimension.addAttribute('zipCode', 'bad', () => 3), /Mapping result must be a string/);
    assert.throws(() => dimension.getGroupIndexFromRootIndexMap('planet'), /No attribute planet was found on dimension location/);
  });

  it('intersects and unions', () => {
    const other = new GenericDimension('location', 'zipCode', ['75018', '13000']);
    assert.deepEqual(dimension.intersect(other).getItems(), ['75018']);
    assert.deepEqual(dimension.union(other).getItems(), ['13000', '31000', '59000', '59133', '75018', '75019']);
    assert.throws(() => dimension.union(new GenericDimension('other', 'zipCode', [])), /not the same dimension/);
  });
});

describe('TimeDimension', () => {
  let dimension;
  beforeEach(() => {
    dimension = new TimeDimension('time', 'month', '2009-12', '2010-02');
  });

  it('items, attributes, maps', () => {
    assert.equal(dimension.numItems, 3);
    assert.sameMembers(dimension.attributes, ['month', 'quarter', 'semester', 'year', 'all']);
    assert.deepEqual(dimension.getItems(), ['2009-12', '2010-01', '2010-02']);
    assert.deepEqual(dimension.getItems('year'), ['2009', '2010']);
    assert.equal(dimension.getGroupItemFromRootItem('year', '2010-01'), '2010');
    assert.equal(dimension.getGroupIndexFromRootIndex('year', 0), 0);
    assert.equal(dimension.getGroupIndexFromRootIndex('year', 1), 1);
    assert.deepEqual(dimension.getGroupIndexFromRootIndexMap('quarter'), [0, 1, 1]);
    assert.deepEqual(dimension.getGroupIndexFromRootIndexMap('all'), [0, 0, 0]);
  });

  it('drills up and down', () => {
    const up = dimension.drillUp('quarter');
    assert.sameMembers(up.attributes, ['quarter', 'semester', 'year', 'all']);
    assert.deepEqual(up.getItems(), ['2009-Q4', '2010-Q1']);
    const down = dimension.drillDown('week_mon');
    assert.sameMembers(down.attributes, ['week_mon', 'month', 'quarter', 'semester', 'year', 'all']);
    assert.deepEqual(down.getItems(), ['2009-W49-mon', '2009-W50-mon', '2009-W51-mon', '2009-W52-mon', '2009-W53-mon', '2010-W01-mon',
      '2010-W02-mon', '2010-W03-mon', '2010-W04-mon', '2010-W05-mon', '2010-W06-mon', '2010-W07-mon', '2010-W08-mon']);
    assert.throws(() => dimension.drillDown('year'), /Invalid periodicity/);
    assert.equal(dimension.drillUp('month'), dimension);
  });

  it('intersect / union', () => {
    const same = new TimeDimension('time', 'month', '2010-01', '2010-02');
    assert.deepEqual(dimension.intersect(same).getItems(), ['2010-01', '2010-02']);
    const quarters = new TimeDimension('time', 'quarter', '2010-Q1', '2010-Q2');
    assert.equal(dimension.intersect(quarters).rootAttribute, 'quarter');
    assert.deepEqual(dimension.intersect(quarters).getItems(), ['2010-Q1']);
    const later = new TimeDimension('time', 'quarter', '2010-Q3', '2010-Q4');
    assert.deepEqual(dimension.intersect(later).getItems(), []);
    assert.equal(dimension.intersect(later).numItems, 0);
    assert.deepEqual(dimension.union(later).getItems(), ['2009-Q4', '2010-Q1', '2010-Q2', '2010-Q3', '2010-Q4']);
  });

  it('diceRange clamps', () => {
    assert.deepEqual(dimension.diceRange('month', '2010-01', '2010-01').getItems(), ['2010-01']);
    assert.deepEqual(dimension.diceRange('month', '2000-01', '2020-01').getItems(), ['2009-12', '2010-01', '2010-02']);
    assert.deepEqual(dimension.diceRange('month', '2010-01', '2020-01').getItems(), ['2010-01', '2010-02']);
    assert.deepEqual(dimension.diceRange('month', '2010-01', null).getItems(), ['2010-01', '2010-02']);
    assert.deepEqual(dimension.diceRange('month', null, '2010-01').getItems(), ['2009-12', '2010-01']);
    assert.deepEqual(dimension.dice('quarter', ['2010-Q1']).getItems(), ['2010-01', '2010-02']);
    assert.throws(() => dimension.dice('month', ['2009-12', '2010-02']), /Unsupported: follow/);
  });

  it('labels', () => {
    assert.deepEqual(dimension.getEntries(), [['2009-12', 'December 2009'], ['2010-01', 'January 2010'], ['2010-02', 'February 2010']]);
    assert.deepEqual(dimension.getEntries('quarter', 'fr'), [['2009-Q4', '4ème trim. 2009'], ['2010-Q1', '1er trim. 2010']]);
  });

  it('day -> month map is plain Gregorian arithmetic (config 5 shape)', () => {
    const days = new TimeDimension('time', 'day', '2010-01-01', '2019-12-31');
    assert.equal(days.numItems, 3652);
    assert.equal(days.getItems('month').length, 120);
    const map = days.getGroupIndexFromRootIndexMap('month');
    // independent recomputation with Date
    let t = Date.UTC(2010, 0, 1);
    for (let i = 0; i < 3652; ++i, t += 86400000) {
      const d = new Date(t);
      assert.equal(map[i], (d.getUTCFullYear() - 2010) * 12 + d.getUTCMonth());
    }
    assert.equal(days.getItems()[59], '2010-03-01');
    assert.equal(days.getItems()[3651], '2019-12-31');
    assert.equal(new TimeDimension('t', 'day', '2012-02', '2012-02').numItems, 29); // leap year
  });

  it('weeks', () => {
    assert.equal(TimeSlot.fromDate(new Date(Date.UTC(2010, 0, 3)), 'week_mon').value, '2009-W53-mon');
    assert.equal(TimeSlot.fromDate(new Date(Date.UTC(2010, 0, 4)), 'week_mon').value, '2010-W01-mon');
    assert.equal(TimeSlot.fromValue('2009-W53-mon').toParentPeriodicity('month').value, '2009-12'); // middle day is Dec 31
    assert.equal(TimeSlot.fromValue('2010-01-W1-mon').lastDate.toISOString().slice(0, 10), '2010-01-03');
    assert.equal(TimeSlot.fromValue('2010-01-W2-mon').firstDate.toISOString().slice(0, 10), '2010-01-04');
    const mw = new TimeDimension('time', 'month_week_mon', '2010-01-W1-mon', '2010-02-W1-mon');
    assert.deepEqual(mw.getItems(), ['2010-01-W1-mon', '2010-01-W2-mon', '2010-01-W3-mon', '2010-01-W4-mon', '2010-01-W5-mon', '2010-02-W1-mon']);
    assert.equal(mw.drillDown('day').numItems, 38);
    const w = new TimeDimension('time2', 'week_mon', '2010-W01-mon', '2010-W08-mon');
    assert.equal(w.numItems, 8);
  });
});

describe('formatters', () => {
  const dims = [{ numItems: 3, getItems: () => ['a', 'b', 'c'] }, { numItems: 2, getItems: () => ['x', 'y'] }];
  it('nests and flattens', () => {
    assert.deepEqual(fmt.toNestedArray([1, 2, 4, 8, 16, 32], dims), [[1, 2], [4, 8], [16, 32]]);
    assert.deepEqual(fmt.fromNestedArray([[1, 2], [4, 8], [16, 32]], dims), [1, 2, 4, 8, 16, 32]);
    assert.deepEqual(fmt.toNestedObject([1, 2, 4, 8, 16, 32], dims), { a: { x: 1, y: 2 }, b: { x: 4, y: 8 }, c: { x: 16, y: 32 } });
    assert.deepEqual(fmt.fromNestedObject({ a: { x: 1, y: 2 }, b: { x: 4, y: 8 }, c: { x: 16, y: 32 } }, dims), [1, 2, 4, 8, 16, 32]);
    assert.equal(fmt.toNestedArray([63], []), 63);
  });
});

describe('wire format (interoperability with blobs written by the reference)', () => {
  const fs = require('fs');
  const path = require('path');
  const { wire } = require('../../olap-in-memory_amd/js');
  const golden = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', 'wire.json'), 'utf8')).cases;
  const bytes = (b64) => wire.toArrayBuffer(Buffer.from(b64, 'base64'));
  const same = (ab, b64) => Buffer.from(ab).toString('base64') === b64;

  it('decodes and re-encodes the reference bytes of a mixed value', () => {
    const value = wire.fromBuffer(bytes(golden.primitives));
    assert.deepEqual(value, [Number.NaN, 32, new Int32Array([255]), 'totot', new Float32Array([666]), { toto: { tata: new Float32Array([666]) } }, null, true, [1.5, 'é']]);
    assert.ok(same(wire.toBuffer(value), golden.primitives), 'encoder output is byte-identical to the reference');
  });

  it('reads a GenericDimension written by the reference and writes the same bytes back', () => {
    const g = golden.genericDimension;
    const dim = GenericDimension.deserialize(bytes(g.blob));
    assert.equal(dim.id, g.id);
    assert.equal(dim.rootAttribute, g.rootAttribute);
    assert.equal(dim.label, g.label);
    assert.deepEqual(dim.attributes, g.attributes);
    for (const attr of Object.keys(g.items)) assert.deepEqual(dim.getItems(attr), g.items[attr]);
    assert.deepEqual(Array.from(dim.getGroupIndexFromRootIndexMap('continent')), g.continentMap);
    assert.deepEqual(dim.getEntries(), [['paris', 'Paris'], ['toledo', 'Toledo'], ['tokyo', 'Tokyo']]);
    assert.ok(same(dim.serialize(), g.blob), 'GenericDimension.serialize() is byte-identical to the reference');
    const own = new GenericDimension('location', 'city', ['paris', 'toledo', 'tokyo'], 'Location', { paris: 'Paris', toledo: 'Toledo', tokyo: 'Tokyo' });
    own.addAttribute('city', 'continent', { paris: 'europe', toledo: 'europe', tokyo: 'asia' });
    assert.ok(same(own.serialize(), g.blob));
  });

  it('TimeDimension round trip', () => {
    const t = new TimeDimension('time', 'month', '2009-12', '2010-02');
    const back = TimeDimension.deserialize(t.serialize());
    assert.deepEqual(back.getItems(), t.getItems());
    assert.deepEqual(back.getItems('quarter'), t.getItems('quarter'));
  });
});

describe('formula parser (no device)', () => {
  const { getParser } = require('../../olap-in-memory_amd/js');
  const parse = (t) => getParser().parse(t);
  it('precedence, associativity, printing', () => {
    assert.equal(parse('routers / antennas').toString(), '(routers / antennas)');
    assert.equal(parse('a + b * 2 ^ 3 ^ 2 - -c').toString(), '((a + (b * (2 ^ (3 ^ 2)))) - (-c))');
    assert.equal(parse('a || b + c').toString(), '((a || b) + c)'); // || sits on the additive level, as in expr-eval
    assert.equal(parse('-x ^ 2').evaluate({ x: 3 }), -9);
    assert.equal(parse('2 ^ -1').evaluate({}), 0.5);
    assert.equal(parse('sqrt 16 + abs(-2)').evaluate({}), 6);
    assert.equal(parse('a ? b : c ? 1 : 2').evaluate({ a: 0, b: 5, c: 0 }), 2);
    assert.deepEqual(parse('max(a__total, b.c) + PI').variables(), ['a__total', 'b.c']);
    assert.equal(parse(parse('min(a, 2) % 3 + if(a, 1, 0)').toString()).evaluate({ a: 7 }), 3);
  });
  it('the reference configuration: || is a NaN-coalescing sum, comparisons and logic are off', () => {
    const or = parse('a || b');
    assert.equal(or.evaluate({ a: Number.NaN, b: 3 }), 3);
    assert.equal(or.evaluate({ a: 2, b: Number.NaN }), 2);
    assert.equal(or.evaluate({ a: 2, b: 3 }), 5);
    assert.ok(Number.isNaN(or.evaluate({ a: Number.NaN, b: Number.NaN })));
    assert.throws(() => parse('a > 1'));
    assert.throws(() => parse('a and b'));
    assert.throws(() => parse('a = 1'));
    assert.throws(() => parse('(a + 1'));
    assert.throws(() => parse('nosuch(a)'));
  });
  it('substitute renames or inlines', () => {
    const e = parse('a / (a + b)');
    assert.equal(e.substitute('a', 'x').toString(), '(x / (x + b))');
    assert.equal(e.substitute('b', parse('c * 2')).toString(), '(a / (a + (c * 2)))');
    assert.equal(e.toString(), '(a / (a + b))');
  });
});

describe('pending-selection bookkeeping (host side of the lazy dice)', () => {
  const { visibleDims, effectiveSelection } = HipStore._internals;
  it('of two new items naming one old item only the last receives the cells', () => {
    assert.deepEqual(Array.from(effectiveSelection([2, 0, 2, -1, 1, 0])), [-1, -1, 2, -1, 1, 0]);
    assert.deepEqual(Array.from(effectiveSelection([])), []);
  });
  it('lines up the dimensions a cube still has with those of the pending selection', () => {
    const pending = (midLen) => ({ midLen: Uint32Array.from(midLen) });
    assert.deepEqual(visibleDims(pending([3, 1, 5]), [3, 5]), [0, 2]); // the sliced-away dimension is skipped
    assert.deepEqual(visibleDims(pending([3, 1, 5]), [3, 1, 5]), [0, 1, 2]); // nothing dropped
    assert.deepEqual(visibleDims(pending([1, 1, 4]), [1, 4]), [0, 2]); // two single-item dimensions: either is fine
    assert.deepEqual(visibleDims(pending([1, 1]), []), []); // everything sliced away
    assert.equal(visibleDims(pending([3, 2, 5]), [3, 5]), null); // a dropped dimension must have one item
    assert.equal(visibleDims(pending([3, 1, 5]), [3, 5, 2]), null); // more dimensions than the selection has
  });
});

describe('errors raised before any device work', () => {
  it('store constructor', () => {
    assert.throws(() => new HipStore(4, 'float32', 1), /Invalid default value, only NaN and 0 are supported/);
    assert.throws(() => new HipStore(4, 'float16', 0), /Invalid type/);
  });
  it('cube measure ids', () => {
    const cube = new Cube([new GenericDimension('d', 'root', ['a'])]);
    assert.throws(() => cube.createStoredMeasure('x'), /Invalid measureId: x/);
    assert.throws(() => cube.getData('nope'), /getData: no such measure nope/);
    assert.throws(() => cube.slice('nope', 'all', 'all'), /slice: no such dimension: nope/);
    assert.equal(cube.storeSize, 1);
  });
});

run();
