#!/usr/bin/env node
/*
 * gen_golden.js — TEST INFRASTRUCTURE, runs only in the authoring container.
 *
 * Executes the reference's own store (`/root/reference/src/store/in-memory.js`) and
 * generic dimension (`/root/reference/src/dimension/generic.js`, `catch-all.js`) under
 * Node and writes input/expected-output vectors to tests/golden/.  Nothing from the
 * reference is copied: the files are `require`d where they lie.  Node 12 cannot parse
 * three ES2020 tokens in those files (`??` at in-memory.js:119, `?.[` at
 * generic.js:279-280); they are down-levelled in memory at load time.  No stand-ins for
 * absent third-party modules are used: cube.js / time.js (which need `timeslot-dag` and
 * `@growblocks/expr-eval`) are NOT loaded.
 *
 *   node oracle/gen_golden.js            # rewrites tests/golden/*.json, *.f32
 *
 * The vectors are data: seeded inputs, per-dimension index maps, expected cells and the
 * expected key set (Map insertion order) produced by the reference.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const Module = require('module');

const REF = '/root/reference/src/';
const OUT = path.join(__dirname, '..', 'tests', 'golden');

const origCompile = Module.prototype._compile;
Module.prototype._compile = function (content, filename) {
  let src = content;
  if (filename.startsWith(REF)) {
    // `return a ?? b;`  ->  `{ const t = a; return t != null ? t : b; }`
    src = src.replace(/return ([^;\n]+?) \?\? ([^;\n]+);/g, '{ const __t = $1; return __t != null ? __t : $2; }');
    // `x?.[k]`  ->  `(x || {})[k]`
    src = src.replace(/([\w.]+(?:\[\w+\])*)\?\.\[/g, '($1 || {})[');
  }
  return origCompile.call(this, src, filename);
};

const InMemoryStore = require(REF + 'store/in-memory.js');
const GenericDimension = require(REF + 'dimension/generic.js');
const CatchAll = require(REF + 'dimension/catch-all.js');

/* ---------- seeded PRNG shared with the C oracle / numpy / HIP generators ---------- */
function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a = (a + 0x6d2b79f5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

/* ---------- JSON helpers (NaN / Infinity / -0 survive) ---------- */
function encNum(v) {
  if (v === undefined) return 'undefined';
  if (v === null) return null;
  if (Number.isNaN(v)) return 'NaN';
  if (v === Infinity) return 'Infinity';
  if (v === -Infinity) return '-Infinity';
  if (Object.is(v, -0)) return '-0';
  return v;
}
const encArr = (a) => Array.from(a, encNum);

function dumpStore(store) {
  const n = store._dataMap.size;
  if (n > 4096) {
    // compact form for long constant runs (the Uint16 contribution-counter cases)
    const vals = Array.from(store._dataMap.values());
    const keys = Array.from(store._dataMap.keys());
    if (keys.every((k, i) => k === i) && vals.every((v) => v === vals[0])) return { size: store._size, iota: n, value: encNum(vals[0]) };
  }
  return {
    size: store._size,
    keys: Array.from(store._dataMap.keys()),
    values: encArr(Array.from(store._dataMap.values())),
  };
}

function makeStore(size, type, def, entries) {
  // entries: [[idx, value], ...] inserted in the given order through setValue
  const s = new InMemoryStore(size, type, def);
  for (const [i, v] of entries) s.setValue(i, v);
  return s;
}

/* ---------- dimension builders ---------- */
function genericDim(id, n, groupMap /* array root idx -> group label idx, or null */) {
  const items = Array.from({ length: n }, (_, j) => `${id}-item${j}`);
  const d = new GenericDimension(id, 'root', items);
  if (groupMap) {
    const m = {};
    items.forEach((it, j) => {
      m[it] = `${id}-grp${groupMap[j]}`;
    });
    d.addAttribute('root', 'grp', m);
  }
  return d;
}

function mapsOf(oldDims, newDims) {
  return newDims.map((nd, i) => Array.from(oldDims[i].getGroupIndexFromRootIndexMap(nd.rootAttribute)));
}

function randomEntries(rnd, size, frac, valueFn, shuffle) {
  const idx = [];
  for (let i = 0; i < size; ++i) if (rnd() < frac) idx.push(i);
  if (shuffle) {
    for (let i = idx.length - 1; i > 0; --i) {
      const j = Math.floor(rnd() * (i + 1));
      const t = idx[i];
      idx[i] = idx[j];
      idx[j] = t;
    }
  }
  return idx.map((i) => [i, valueFn(i)]);
}

const METHODS = ['sum', 'average', 'highest', 'lowest', 'first', 'last', 'product'];
const defName = (d) => (Number.isNaN(d) ? 'NaN' : 0);

function drillUpCase(name, type, def, method, lens, groupMaps, entries) {
  const oldDims = lens.map((n, i) => genericDim(`d${i}`, n, groupMaps[i]));
  const newDims = oldDims.map((d, i) => (groupMaps[i] ? d.drillUp('grp') : d));
  const store = makeStore(lens.reduce((a, b) => a * b, 1), type, def, entries);
  const out = store.drillUp(oldDims, newDims, method);
  return {
    name,
    op: 'drillUp',
    type,
    default: defName(def),
    method,
    oldLen: lens,
    newLen: newDims.map((d) => d.numItems),
    maps: mapsOf(oldDims, newDims),
    in: dumpStore(store),
    out: dumpStore(out),
  };
}

/* group map where dimension -> 'all' */
function drillUpAllCase(name, type, def, method, lens, axis, entries) {
  const oldDims = lens.map((n, i) => genericDim(`d${i}`, n, null));
  const newDims = oldDims.map((d, i) => (i === axis ? d.drillUp('all') : d));
  const store = makeStore(lens.reduce((a, b) => a * b, 1), type, def, entries);
  const out = method === undefined ? store.drillUp(oldDims, newDims) : store.drillUp(oldDims, newDims, method);
  return {
    name,
    op: 'drillUp',
    type,
    default: defName(def),
    method: method === undefined ? 'sum' : method,
    oldLen: lens,
    newLen: newDims.map((d) => d.numItems),
    maps: mapsOf(oldDims, newDims),
    in: dumpStore(store),
    out: dumpStore(out),
  };
}

function drillDownCase(name, type, def, method, newLens, groupMaps, entries, distributions) {
  // new (fine) dims carry a 'grp' attribute; old (coarse) dims are their drillUp
  const newDims = newLens.map((n, i) => genericDim(`d${i}`, n, groupMaps[i]));
  const oldDims = newDims.map((d, i) => (groupMaps[i] ? d.drillUp('grp') : d));
  const oldSize = oldDims.reduce((a, d) => a * d.numItems, 1);
  const store = makeStore(oldSize, type, def, entries);
  const c = {
    name,
    op: 'drillDown',
    type,
    default: defName(def),
    method: method === undefined ? 'sum' : method,
    oldLen: oldDims.map((d) => d.numItems),
    newLen: newLens,
    // drillDown maps go new root idx -> old idx (in-memory.js:349-353)
    maps: oldDims.map((od, i) => Array.from(newDims[i].getGroupIndexFromRootIndexMap(od.rootAttribute))),
    distributions: distributions ? encArr(distributions) : null,
    in: dumpStore(store),
  };
  try {
    const out = store.drillDown(oldDims, newDims, method, distributions || null);
    c.out = dumpStore(out);
  } catch (e) {
    c.throws = e.message;
  }
  return c;
}

/* addDimension-shaped drillDown: CatchAll(1 item '_total') -> real dimension (cube.js:919-945) */
function addDimensionCase(name, type, def, method, baseLens, index, addLen, entries, distributions) {
  const base = baseLens.map((n, i) => genericDim(`d${i}`, n, null));
  const added = genericDim('added', addLen, null);
  const oldDims = base.slice();
  oldDims.splice(index, 0, new CatchAll('added', added));
  const newDims = oldDims.slice();
  newDims[index] = added;
  const store = makeStore(baseLens.reduce((a, b) => a * b, 1), type, def, entries);
  const c = {
    name,
    op: 'drillDown',
    type,
    default: defName(def),
    method: method === undefined ? 'sum' : method,
    oldLen: oldDims.map((d) => d.numItems),
    newLen: newDims.map((d) => d.numItems),
    maps: oldDims.map((od, i) => Array.from(newDims[i].getGroupIndexFromRootIndexMap(od.rootAttribute))),
    distributions: distributions ? encArr(distributions) : null,
    in: dumpStore(store),
  };
  try {
    const out = store.drillDown(oldDims, newDims, method, distributions || null);
    c.out = dumpStore(out);
  } catch (e) {
    c.throws = e.message;
  }
  return c;
}

function diceCase(name, type, def, lens, selections /* per dim: array of old idx (may include -1) or null */, entries) {
  const oldDims = lens.map((n, i) => genericDim(`d${i}`, n, null));
  // build the new dimensions directly from item lists so that order/unknown items are under test control
  const newDims = oldDims.map((d, i) => {
    if (!selections[i]) return d;
    const items = selections[i].map((j) => (j < 0 ? `d${i}-missing${-j}` : d.getItems()[j]));
    return new GenericDimension(d.id, 'root', items);
  });
  const store = makeStore(lens.reduce((a, b) => a * b, 1), type, def, entries);
  const out = store.dice(oldDims, newDims);
  return {
    name,
    op: 'dice',
    type,
    default: defName(def),
    oldLen: lens,
    newLen: newDims.map((d) => d.numItems),
    sel: newDims.map((nd, i) => {
      const toIdx = oldDims[i].getItemsToIdx();
      return nd.getItems().map((it) => (toIdx[it] === undefined ? -1 : toIdx[it]));
    }),
    in: dumpStore(store),
    out: dumpStore(out),
  };
}

function reorderCase(name, type, def, lens, perm /* new axis i = old axis perm[i] */, entries) {
  const oldDims = lens.map((n, i) => genericDim(`d${i}`, n, null));
  const newDims = perm.map((p) => oldDims[p]);
  const store = makeStore(lens.reduce((a, b) => a * b, 1), type, def, entries);
  const out = store.reorder(oldDims, newDims);
  return {
    name,
    op: 'reorder',
    type,
    default: defName(def),
    oldLen: lens,
    newLen: newDims.map((d) => d.numItems),
    perm,
    in: dumpStore(store),
    out: dumpStore(out),
  };
}

function loadCase(name, type, myDef, hisDef, myLens, hisSel /* per dim: my idx list for his items */, myEntries, hisEntries) {
  const myDims = myLens.map((n, i) => genericDim(`d${i}`, n, null));
  const hisDims = myDims.map((d, i) => new GenericDimension(d.id, 'root', hisSel[i].map((j) => d.getItems()[j])));
  const mine = makeStore(myLens.reduce((a, b) => a * b, 1), type, myDef, myEntries);
  const his = makeStore(hisDims.reduce((a, d) => a * d.numItems, 1), type, hisDef, hisEntries);
  const before = dumpStore(mine);
  mine.load(his, myDims, hisDims);
  return {
    name,
    op: 'load',
    type,
    default: defName(myDef),
    hisDefault: defName(hisDef),
    myLen: myLens,
    hisLen: hisDims.map((d) => d.numItems),
    hisToMine: hisSel,
    in: before,
    his: dumpStore(his),
    out: dumpStore(mine),
  };
}

/* ====================== 1. known-answer vectors ====================== */
function buildKats() {
  const cases = [];
  const N = Number.NaN;
  const seq = (vals) => vals.map((v, i) => [i, v]).filter(([, v]) => v !== undefined);

  // SURVEY §8(a) table, confirmed behaviours of the reference store
  cases.push(drillUpAllCase('sum_cancels_to_unset', 'float32', 0, 'sum', [2], 0, seq([5, -5])));
  cases.push(drillUpAllCase('sum_cancel_then_more', 'float32', 0, 'sum', [3], 0, seq([5, -5, 3])));
  cases.push(drillUpAllCase('highest_ignores_unset', 'float32', 0, 'highest', [2], 0, seq([-3, undefined])));
  cases.push(drillUpAllCase('product_ignores_unset', 'float32', 0, 'product', [3], 0, seq([2, undefined, 3])));
  cases.push(drillUpAllCase('highest_nan_propagates', 'float32', 0, 'highest', [3], 0, seq([1, N, 2])));
  cases.push(drillUpAllCase('lowest_nan_propagates', 'float32', 0, 'lowest', [3], 0, seq([1, N, 2])));
  cases.push(drillUpAllCase('average_counts_stored_zero_nan_default', 'float32', N, 'average', [3], 0, seq([10, 0, 20])));
  cases.push(drillUpAllCase('average_zero_is_unset_zero_default', 'float32', 0, 'average', [3], 0, seq([10, 0, 20])));
  cases.push(drillUpAllCase('uint32_sum_no_wrap', 'uint32', 0, 'sum', [2], 0, seq([4e9, 4e9])));
  cases.push(drillUpAllCase('int32_average_fraction', 'int32', 0, 'average', [2], 0, seq([7, 8])));
  cases.push(drillUpAllCase('average_65536_wraps', 'float32', 0, 'average', [65536], 0, seq(new Array(65536).fill(1))));
  cases.push(drillUpAllCase('average_65537', 'float32', 0, 'average', [65537], 0, seq(new Array(65537).fill(1))));
  cases.push(drillUpAllCase('first_insertion_order', 'float32', 0, 'first', [3], 0, [[2, 30], [0, 10], [1, 20]]));
  cases.push(drillUpAllCase('last_insertion_order', 'float32', 0, 'last', [3], 0, [[2, 30], [0, 10], [1, 20]]));
  cases.push(drillUpAllCase('default_method_is_sum', 'float32', 0, undefined, [3], 0, seq([1, 2, 3])));
  cases.push(drillUpAllCase('product_restart_after_zero', 'float32', 0, 'product', [4], 0, seq([1e-30, 1e-30 * 1e-300, 3, 4])));
  cases.push(drillUpAllCase('sum_inf_minus_inf_nan_default', 'float32', N, 'sum', [3], 0, seq([Infinity, -Infinity, 5])));
  cases.push(drillUpAllCase('sum_negative_zero', 'float32', N, 'sum', [2], 0, seq([-0, -0])));
  cases.push(drillUpAllCase('empty_store', 'float32', 0, 'sum', [4, 3], 0, []));
  cases.push(drillUpAllCase('empty_store_nan', 'float32', N, 'average', [4, 3], 1, []));

  // reference test cube 3x2 (test/helpers/create-test-cube.js:45-49), every aggregator, both axes
  const ant = seq([1, 2, 4, 8, 16, 32]);
  for (const m of METHODS) {
    cases.push(drillUpAllCase(`testcube_remove_location_${m}`, 'float32', 0, m, [3, 2], 0, ant));
    cases.push(drillUpAllCase(`testcube_remove_period_${m}`, 'uint32', 0, m, [3, 2], 1, ant));
  }
  // cities -> continents (paris,toledo -> europe ; tokyo -> asia), test/cube-drilling.js:15-24
  cases.push(drillUpCase('testcube_cities_to_continents', 'uint32', 0, 'sum', [3, 2], [[0, 0, 1], null], ant));
  // citySize numbering: big, small, big  (first-appearance numbering, generic.js:100-107)
  for (const m of METHODS) cases.push(drillUpCase(`testcube_city_size_${m}`, 'uint32', 0, m, [3, 2], [[0, 1, 0], null], ant));

  // drillDown integer remainder spreading (in-memory.js:403-417)
  cases.push(drillDownCase('int_100_to_3', 'uint32', 0, 'sum', [3], [[0, 0, 0]], [[0, 100]]));
  cases.push(drillDownCase('int_32_to_3', 'uint32', 0, 'sum', [3], [[0, 0, 0]], [[0, 32]]));
  cases.push(drillDownCase('int_67_to_3', 'int32', 0, 'sum', [3], [[0, 0, 0]], [[0, 67]]));
  cases.push(drillDownCase('int_14_to_3', 'int32', 0, 'sum', [3], [[0, 0, 0]], [[0, 14]]));
  cases.push(drillDownCase('int_neg7_to_3', 'int32', 0, 'sum', [3], [[0, 0, 0]], [[0, -7]]));
  cases.push(drillDownCase('int_5_to_7', 'int32', 0, 'sum', [7], [[0, 0, 0, 0, 0, 0, 0]], [[0, 5]]));
  cases.push(drillDownCase('float_90_q_to_m_nan', 'float32', N, 'sum', [6], [[0, 0, 0, 1, 1, 1]], [[0, 90]]));
  cases.push(drillDownCase('stored_zero_skipped_nan_default', 'float32', N, 'sum', [6], [[0, 0, 0, 1, 1, 1]], [[0, 0], [1, 9]]));
  cases.push(drillDownCase('average_copies', 'float32', 0, 'average', [6], [[0, 0, 0, 1, 1, 1]], [[0, 90], [1, 3]]));
  cases.push(drillDownCase('default_method_is_sum', 'float32', 0, undefined, [4], [[0, 0, 1, 1]], [[0, 5], [1, 7]]));
  cases.push(drillDownCase('interleaved_groups_int', 'int32', 0, 'sum', [5], [[0, 1, 0, 1, 0]], [[0, 10], [1, 7]]));
  cases.push(drillDownCase('two_axes_int', 'int32', 0, 'sum', [4, 3], [[0, 0, 1, 1], [0, 0, 0]], [[0, 100], [1, 7]]));
  cases.push(drillDownCase('nan_value_skipped', 'float32', 0, 'sum', [2], [[0, 0]], [[0, N]]));
  // addDimension with and without distributions
  cases.push(addDimensionCase('add_dim_append_float', 'float32', 0, 'sum', [2], 1, 3, [[0, 100], [1, 100]]));
  cases.push(addDimensionCase('add_dim_append_int', 'uint32', 0, 'sum', [1], 1, 3, [[0, 32]]));
  cases.push(addDimensionCase('add_dim_front_avg', 'float32', 0, 'average', [2], 0, 3, [[0, 100], [1, 50]]));
  cases.push(addDimensionCase('add_dim_no_rule', 'float32', 0, undefined, [2], 1, 3, [[0, 9], [1, 3]]));
  cases.push(addDimensionCase('add_dim_distributions_append', 'float32', 0, 'sum', [2], 1, 3, [[0, 100], [1, 10]], [0.5, 0.3, 0.2, 0.1, 0.1, 0.8]));
  cases.push(addDimensionCase('add_dim_distributions_shared', 'float32', 0, 'sum', [2, 2], 2, 2, [[0, 1], [1, 2], [2, 3], [3, 4]], [0.25, 0.75, 0.5, 0.5]));
  cases.push(addDimensionCase('add_dim_distributions_missing', 'float32', 0, 'sum', [2], 1, 3, [[0, 100], [1, 10]], [0.5, 0.3, 0.2]));

  // dice (test/cube-filtering.js:48-118 at store level)
  cases.push(diceCase('dice_keep_two_cities', 'uint32', 0, [3, 2], [[0, 1], null], ant));
  cases.push(diceCase('dice_reversed', 'uint32', 0, [3, 2], [[1, 0], null], ant));
  cases.push(diceCase('dice_winter', 'uint32', 0, [3, 2], [null, [1]], ant));
  cases.push(diceCase('dice_missing_item', 'uint32', 0, [3, 2], [[-1, 0], null], ant));
  cases.push(diceCase('dice_empty', 'uint32', 0, [3, 2], [[], null], ant));
  cases.push(diceCase('dice_nan_default', 'float32', N, [3, 2], [[2, 0], [1, 0]], seq([1, 0, undefined, 8, 16, undefined])));

  // reorder (test/cube-dimension.js:210-279 at store level)
  cases.push(reorderCase('reorder_2d', 'uint32', 0, [3, 2], [1, 0], ant));
  const eight = seq([1, 2, 3, 4, 5, 6, 7, 8]);
  for (const p of [[0, 1, 2], [0, 2, 1], [2, 1, 0], [2, 0, 1], [1, 2, 0], [1, 0, 2]]) cases.push(reorderCase(`reorder_3d_${p.join('')}`, 'float32', N, [2, 2, 2], p, eight));

  // load (test/cube-to-cube.js:403-606 at store level)
  cases.push(loadCase('load_subset', 'uint32', 0, 0, [2, 3], [[1], [0, 2]], [], [[0, 32], [1, 53]]));
  cases.push(loadCase('load_overwrites_with_default', 'float32', 0, 0, [2, 3], [[1], [0, 2]], seq([1, 1, 1, 1, 1, 1]), [[1, 53]]));
  cases.push(loadCase('load_nan_default_into_zero', 'float32', 0, N, [2, 3], [[0, 1], [2, 0]], seq([1, 1, 1, 1, 1, 1]), [[0, 7], [3, 0]]));
  return cases;
}

/* ====================== 2. seeded random vectors ====================== */
function buildRandom() {
  const cases = [];
  const rnd = mulberry32(20240807);
  const ri = (n) => Math.floor(rnd() * n);
  const shapes = [
    [7],
    [64],
    [5, 6],
    [12, 1],
    [1, 9],
    [4, 5, 6],
    [3, 17, 4],
    [2, 3, 4, 5],
    [6, 2, 3, 2, 2],
    [130, 3],
    [3, 130],
    [3, 70, 2],
  ];
  const types = ['float32', 'float64', 'int32', 'uint32'];
  let id = 0;
  function valueGen(type, style) {
    // all values are exactly representable in the declared type
    if (type === 'int32') return () => ri(41) - 20 || 1;
    if (type === 'uint32') return () => ri(50) + (style === 2 ? 4e9 : 1);
    if (style === 0) return () => Math.fround(rnd() + 0.5); // SURVEY §8(d) generator
    if (style === 1) return () => ri(9) - 4 || (rnd() < 0.5 ? 0 : 1); // small integers incl. cancellations and zeros
    return () => Math.fround((rnd() - 0.5) * 2000);
  }
  function randomGroupMap(n) {
    const g = Math.max(1, ri(n) + (n > 2 ? 0 : 1));
    // first-appearance numbering comes from addAttribute, labels are arbitrary
    return Array.from({ length: n }, () => ri(g));
  }
  for (const lens of shapes) {
    const size = lens.reduce((a, b) => a * b, 1);
    for (let rep = 0; rep < 3; ++rep) {
      for (const method of METHODS) {
        const type = types[(id + rep) % 4];
        const def = (id >> 1) % 2 === 0 ? 0 : Number.NaN;
        const style = id % 3;
        const frac = [1.0, 0.6, 0.15][(id >> 2) % 3];
        const gen = valueGen(type, style);
        const shuffle = id % 5 === 0;
        const axis = ri(lens.length);
        const multi = id % 7 === 0 && lens.length > 1;
        const groupMaps = lens.map((n, i) => (i === axis || (multi && rnd() < 0.6) ? randomGroupMap(n) : null));
        const entries = randomEntries(rnd, size, frac, gen, shuffle);
        if (def === 0 && (type === 'float32' || type === 'float64') && id % 11 === 0 && entries.length > 2) entries[1][1] = Number.NaN;
        cases.push(drillUpCase(`rnd_drillup_${id}`, type, def, method, lens, groupMaps, entries));
        ++id;
      }
    }
  }
  // drillDown
  for (const lens of shapes.filter((s) => s.reduce((a, b) => a * b, 1) <= 600)) {
    for (let rep = 0; rep < 4; ++rep) {
      const type = types[(id + rep) % 4];
      const def = id % 2 === 0 ? 0 : Number.NaN;
      const method = ['sum', 'average', 'first', undefined][id % 4];
      const axis = ri(lens.length);
      const multi = id % 5 === 0 && lens.length > 1;
      const groupMaps = lens.map((n, i) => (i === axis || (multi && rnd() < 0.6) ? randomGroupMap(n) : null));
      // old size is only known after building the dims: generate entries lazily over a generous bound
      const newDims = lens.map((n, i) => genericDim(`d${i}`, n, groupMaps[i]));
      const oldSize = newDims.reduce((a, d, i) => a * (groupMaps[i] ? d.getItems('grp').length : d.numItems), 1);
      const gen = valueGen(type, type.startsWith('float') ? [0, 1, 2][id % 3] : 0);
      const entries = randomEntries(rnd, oldSize, [1.0, 0.5][id % 2], gen, false);
      cases.push(drillDownCase(`rnd_drilldown_${id}`, type, def, method, lens, groupMaps, entries, null));
      ++id;
    }
  }
  // dice
  for (const lens of shapes) {
    const size = lens.reduce((a, b) => a * b, 1);
    for (let rep = 0; rep < 3; ++rep) {
      const type = types[(id + rep) % 4];
      const def = id % 2 === 0 ? 0 : Number.NaN;
      const sel = lens.map((n) => {
        if (rnd() < 0.35) return null;
        const keep = [];
        for (let j = 0; j < n; ++j) if (rnd() < 0.6) keep.push(j);
        if (rnd() < 0.5) keep.reverse();
        if (rnd() < 0.2) keep.splice(ri(keep.length + 1), 0, -1);
        return keep;
      });
      const entries = randomEntries(rnd, size, [1.0, 0.4][id % 2], valueGen(type, 0), id % 3 === 0);
      cases.push(diceCase(`rnd_dice_${id}`, type, def, lens, sel, entries));
      ++id;
    }
  }
  // reorder
  for (const lens of shapes.filter((s) => s.length > 1)) {
    const size = lens.reduce((a, b) => a * b, 1);
    const perm = lens.map((_, i) => i);
    for (let i = perm.length - 1; i > 0; --i) {
      const j = ri(i + 1);
      const t = perm[i];
      perm[i] = perm[j];
      perm[j] = t;
    }
    const type = types[id % 4];
    const def = id % 2 === 0 ? 0 : Number.NaN;
    cases.push(reorderCase(`rnd_reorder_${id}`, type, def, lens, perm, randomEntries(rnd, size, 0.7, valueGen(type, 0), false)));
    ++id;
  }
  return cases;
}

/* ====================== 3. BASELINE.json configs 1 and 2 ====================== */
function configCube(lens, seed, frac) {
  const size = lens.reduce((a, b) => a * b, 1);
  const rnd = mulberry32(seed);
  const store = new InMemoryStore(size, 'float32', 0);
  // one draw for the value, one for the Bernoulli mask, per cell, in index order
  for (let i = 0; i < size; ++i) {
    const v = Math.fround(0.5 + rnd());
    const keep = rnd() < frac;
    if (keep) store.setValue(i, v);
  }
  return store;
}

function buildConfigs() {
  const meta = [];
  // config 1: 10x10x10, drillUp dim0 -> all, through the reference store
  {
    const lens = [10, 10, 10];
    const dims = lens.map((n, i) => genericDim(`dimension${i}`, n, null));
    const store = configCube(lens, 20240807, 1.0);
    const out = store.drillUp(dims, dims.map((d, i) => (i === 0 ? d.drillUp('all') : d)), 'sum');
    meta.push({ name: 'config1_10x10x10_dim0', lens, axis: 0, seed: 20240807, frac: 1.0, method: 'sum', outKeys: out._dataMap.size, out: encArr(out.data) });
  }
  // config 2: [10]^6, axes 0 / 3 / 5, dense; axis 3 also at 10 % fill
  const lens6 = [10, 10, 10, 10, 10, 10];
  const dims6 = lens6.map((n, i) => genericDim(`dimension${i}`, n, null));
  const runs = [
    [0, 1.0],
    [3, 1.0],
    [5, 1.0],
    [3, 0.1],
  ];
  let cached = {};
  for (const [axis, frac] of runs) {
    const key = String(frac);
    if (!cached[key]) cached[key] = configCube(lens6, 20240807, frac);
    const store = cached[key];
    const out = store.drillUp(dims6, dims6.map((d, i) => (i === axis ? d.drillUp('all') : d)), 'sum');
    const data = out.data;
    const f32 = Float32Array.from(data); // Math.fround of the reference's float64 result
    const present = new Uint8Array(out._size);
    for (const k of out._dataMap.keys()) present[k] = 1;
    const base = `config2_axis${axis}_fill${Math.round(frac * 100)}`;
    fs.writeFileSync(path.join(OUT, base + '.f32'), Buffer.from(f32.buffer));
    fs.writeFileSync(path.join(OUT, base + '.present.u8'), Buffer.from(present.buffer));
    // float64 spot values (exact) for an accumulation-width check
    const spots = [];
    for (let s = 0; s < 64; ++s) {
      const i = Math.floor(((s + 0.5) * data.length) / 64);
      spots.push([i, encNum(data[i])]);
    }
    meta.push({ name: base, lens: lens6, axis, seed: 20240807, frac, method: 'sum', outSize: out._size, outKeys: out._dataMap.size, inKeys: store._dataMap.size, spots, total: encNum(out.total) });
  }
  return meta;
}

/* ====================== 4. wire format (src/serialization.js) ====================== */
function buildWire() {
  const { toBuffer } = require(REF + 'serialization.js');
  const b64 = (ab) => Buffer.from(ab).toString('base64');
  const primitives = [Number.NaN, 32, new Int32Array([255]), 'totot', new Float32Array([666]), { toto: { tata: new Float32Array([666]) } }, null, true, [1.5, 'é']];
  const location = new GenericDimension('location', 'city', ['paris', 'toledo', 'tokyo'], 'Location', { paris: 'Paris', toledo: 'Toledo', tokyo: 'Tokyo' });
  location.addAttribute('city', 'continent', { paris: 'europe', toledo: 'europe', tokyo: 'asia' });
  const store = makeStore(12, 'float32', 0, [[1, 1.5], [4, -2], [7, 1e10], [11, 0.25]]);
  const storeNan = makeStore(5, 'uint32', Number.NaN, [[0, 0], [3, 4000000000]]);
  return {
    primitives: b64(toBuffer(primitives)),
    genericDimension: {
      blob: b64(location.serialize()),
      id: 'location', rootAttribute: 'city', label: 'Location', attributes: location.attributes,
      items: { all: location.getItems('all'), city: location.getItems('city'), continent: location.getItems('continent') },
      continentMap: Array.from(location.getGroupIndexFromRootIndexMap('continent')),
    },
    store: { blob: b64(store.serialize()), dump: dumpStore(store), type: 'float32', default: 0 },
    storeNanDefault: { blob: b64(storeNan.serialize()), dump: dumpStore(storeNan), type: 'uint32', default: 'NaN' },
  };
}

function writeJson(file, obj) {
  fs.writeFileSync(path.join(OUT, file), JSON.stringify(obj));
  console.log(`${file}: ${Array.isArray(obj.cases) ? obj.cases.length : ''} cases`);
}

fs.mkdirSync(OUT, { recursive: true });
const header = {
  generator: 'oracle/gen_golden.js',
  reference: 'Growblocks/olap-in-memory @ 2024_08_07, src/store/in-memory.js + src/dimension/generic.js executed under node ' + process.version,
};
writeJson('store_kat.json', Object.assign({ cases: buildKats() }, header));
writeJson('store_random.json', Object.assign({ cases: buildRandom() }, header));
writeJson('configs.json', Object.assign({ cases: buildConfigs() }, header));
writeJson('wire.json', Object.assign({ cases: buildWire() }, header));
