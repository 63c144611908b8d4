'use strict';
/*
 * Cube — same public surface as the reference's Cube for the aggregation path
 * (/root/reference/src/cube.js, src/index.d.ts:37-148), with every stored measure held in a
 * HipStore (MI355X HBM).  The class is host-side orchestration only: a query rebuilds ONE
 * dimension, then asks each measure's store for the matching bulk operation with the rule
 * `storedMeasuresRules[measure][dimensionId]` (src/cube.js:1012-1020).
 *
 * Kept from the reference because callers rely on it (SURVEY.md §8(a7), §8(b)):
 *   - queries never mutate; a no-op returns the SAME cube object (src/cube.js:997, :843, :818, :968);
 *   - derived cubes share `computedMeasures` and the rules object by reference, except
 *     addDimension / removeDimension which deep-copy the rules (:931, :957);
 *   - removeDimension = drillUp(id, 'all') with the dimension dropped from the list (:950-964);
 *   - slice = dice + removeDimension (:799-807); collapse = slice every dimension to 'all' (:320-324).
 * Computed measures use ./formula.js (own parser for the arithmetic subset of expr-eval the
 * reference enables) and are evaluated by one element-wise device launch; (de)serialisation uses
 * the reference's container (./wire.js).
 */
const HipStore = require('./store/hip');
const CatchAllDimension = require('./dimension/catch-all');
const TimeSlot = require('./calendar');
const { toNestedArray, fromNestedArray, toNestedObject, fromNestedObject } = require('./formatter');
const { toBuffer, fromBuffer, toArrayBuffer } = require('./wire');
const { getParser } = require('./formula');
const backend = require('./backend');

const MEASURE_ID = /^[a-z][_a-z0-9]+$|^[_a-z0-9]+__total$/i;

const deepCopy = (value) => (value === undefined ? undefined : JSON.parse(JSON.stringify(value)));

function deepMerge(target, source) {
  if (source === null || typeof source !== 'object') return source;
  const out = target !== null && typeof target === 'object' ? target : {};
  for (const key of Object.keys(source)) out[key] = deepMerge(out[key], source[key]);
  return out;
}

function cartesian(options) {
  const keys = Object.keys(options);
  let rows = [{}];
  for (const key of keys) {
    const next = [];
    for (const row of rows) for (const item of options[key]) next.push(Object.assign({}, row, { [key]: item }));
    rows = next;
  }
  return rows;
}

class Cube {
  constructor(dimensions) {
    this.dimensions = dimensions;
    this.storedMeasures = {};
    this.storedMeasuresRules = {};
    this.computedMeasures = {};
  }

  // ------------------------------------------------------------------ introspection
  get storeSize() {
    return this.dimensions.reduce((n, d) => n * d.numItems, 1);
  }

  get byteLength() {
    return Object.values(this.storedMeasures).reduce((n, store) => n + store.byteLength, 0);
  }

  get dimensionIds() {
    return this.dimensions.map((d) => d.id);
  }

  get storedMeasureIds() {
    return Object.keys(this.storedMeasures);
  }

  get computedMeasureIds() {
    return Object.keys(this.computedMeasures);
  }

  getDimension(dimensionId) {
    return this.dimensions.find((d) => d.id === dimensionId);
  }

  getDimensionIndex(dimensionId) {
    return this.dimensions.findIndex((d) => d.id === dimensionId);
  }

  // ------------------------------------------------------------------ measures
  _checkNewMeasure(measureId) {
    if (!MEASURE_ID.test(measureId)) throw new Error(`Invalid measureId: ${measureId}`);
    if (this.storedMeasures[measureId] !== undefined) throw new Error(`This measure already exists: ${measureId}`);
  }

  createStoredMeasure(measureId, rules = {}, type = 'float32', defaultValue = 0) {
    this._checkNewMeasure(measureId);
    this.storedMeasures[measureId] = this._newStore(type, defaultValue, rules);
    this.storedMeasuresRules[measureId] = rules;
  }

  /**
   * A measure's store.  `first` / `last` follow the Map's INSERTION order in the reference (in-memory.js:298): a measure
   * with such a rule keeps that order (HipStore.trackOrder) and therefore stays on ONE device even when a device list
   * is set — the shards of a split measure combine in row order, which is the reference's order only for a cube
   * filled in ascending index order — so that the same program answers the same with and without setDevices().
   */
  _newStore(type, defaultValue, rules) {
    const ordered = Object.values(rules || {}).some((rule) => rule === 'first' || rule === 'last');
    const store = new HipStore(this.storeSize, type, defaultValue, undefined, ordered ? undefined : this.dimensions.map((d) => d.numItems));
    if (ordered) store.trackOrder();
    return store;
  }

  /**
   * A measure defined by a formula over stored measures (and `<measure>__total`), evaluated per
   * cell on demand (src/cube.js:97-140).  Formulas naming other computed measures are inlined.
   */
  createComputedMeasure(measureId, formula) {
    if (!MEASURE_ID.test(measureId)) throw new Error(`Invalid measureId: ${measureId}`);
    if (this.storedMeasures[measureId] !== undefined || this.computedMeasures[measureId] !== undefined) throw new Error(`This measure already exists ${measureId}`);
    let text = formula;
    for (const id of this.computedMeasureIds) {
      const whole = new RegExp(`\\b${id}\\b`, 'g');
      if (text.match(whole)) text = text.replace(whole, `(${this.computedMeasures[id].toString()})`);
    }
    const expression = getParser().parse(text);
    const known = this.storedMeasureIds.concat(this.storedMeasureIds.map((m) => `${m}__total`));
    const variables = expression.variables({ withMembers: true });
    if (!variables.every((v) => known.includes(v))) throw new Error(`Unknown measure(s): ${variables.filter((v) => !this.storedMeasureIds.includes(v))}`);
    this.computedMeasures[measureId] = expression;
  }

  copyToStoredMeasure(computedMeasureId, storedMeasureId, rules = {}, type = 'float32', defaultValue = 0) {
    const data = this.getData(computedMeasureId);
    this.createStoredMeasure(storedMeasureId, rules, type, defaultValue);
    this.setData(storedMeasureId, data);
  }

  convertToStoredMeasure(measureId, rules = {}, type = 'float32', defaultValue = 0) {
    if (!this.computedMeasures[measureId]) throw new Error(`convertToStoredMeasure: no such computed measure: ${measureId}`);
    const data = this.getData(measureId);
    this.dropMeasure(measureId);
    this.createStoredMeasure(measureId, rules, type, defaultValue);
    this.setData(measureId, data);
  }

  replaceStoredMeasure(toKeep, toDrop) {
    for (const id of [toKeep, toDrop]) if (this.storedMeasures[id] === undefined) throw new Error(`replaceStoredMeasure: no such measure ${id}`);
    for (const id of this.computedMeasureIds) {
      const expression = this.computedMeasures[id];
      if (expression.variables().includes(toDrop)) this.computedMeasures[id] = expression.substitute(toDrop, toKeep);
    }
    this.dropMeasure(toDrop);
  }

  /** One device launch evaluates the formula for every cell (olap_eval_formula). */
  _evaluateComputed(measureId) {
    const expression = this.computedMeasures[measureId];
    const inputs = {};
    const scalars = {};
    const stores = [];
    const totals = [];
    for (const name of expression.variables({ withMembers: true })) {
      if (name.includes('__total')) {
        scalars[name] = totals.push(this.storedMeasures[name.replace('__total', '')].total) - 1;
      } else {
        inputs[name] = stores.push(this.storedMeasures[name]) - 1;
      }
    }
    if (stores.length === 0) {
      // a formula of constants / totals only: nothing to stream, evaluate once on the host
      const params = {};
      Object.keys(scalars).forEach((name) => {
        params[name] = totals[scalars[name]];
      });
      return new Array(this.storeSize).fill(expression.evaluate(params));
    }
    const program = expression.compile(inputs, scalars);
    const addon = backend.load();
    const natives = stores.map((store) => store._native);
    // measures split over several devices alike: evaluated per shard; otherwise on one device (gathered if need be)
    if (natives.every((native) => native.isSharded)) {
      try {
        return HipStore.toPlainArray(addon.evalFormulaSharded(program.code, program.consts, natives, Float64Array.from(totals)));
      } catch (e) {
        if (!/^sharded:/.test(e.message)) throw e;
      }
    }
    return HipStore.toPlainArray(addon.evalFormula(program.code, program.consts, stores.map((store) => store._whole), Float64Array.from(totals)));
  }

  copyStoredMeasure(measureId, copyMeasureId) {
    if (!MEASURE_ID.test(copyMeasureId)) throw new Error(`Invalid measureId: ${copyMeasureId}`);
    if (this.storedMeasures[measureId] === undefined) throw new Error(`This measure does not exists: ${measureId}`);
    if (this.storedMeasures[copyMeasureId] !== undefined) throw new Error(`This measure already exists: ${copyMeasureId}`);
    this.storedMeasures[copyMeasureId] = this.storedMeasures[measureId].clone();
    this.storedMeasuresRules[copyMeasureId] = deepCopy(this.storedMeasuresRules[measureId]);
  }

  cloneStoredMeasure(originCube, measureId) {
    this._checkNewMeasure(measureId);
    const origin = originCube.storedMeasures[measureId];
    if (origin === undefined) throw new Error(`This measure does not exists in originCube: ${measureId}`);
    this.storedMeasuresRules[measureId] = Object.assign({}, originCube.storedMeasuresRules[measureId]);
    this.storedMeasures[measureId] = this._newStore(origin._type, origin._defaultValue, origin.orderTracked ? { any: 'first' } : this.storedMeasuresRules[measureId]);
  }

  renameMeasure(oldMeasureId, newMeasureId) {
    // eslint-disable-next-line eqeqeq
    if (oldMeasureId == newMeasureId) return;
    if (this.computedMeasures[oldMeasureId]) {
      this.computedMeasures[newMeasureId] = this.computedMeasures[oldMeasureId];
      delete this.computedMeasures[oldMeasureId];
      return;
    }
    if (!this.storedMeasures[oldMeasureId]) throw new Error(`renameMeasure: no such measure ${oldMeasureId} -> ${newMeasureId}`);
    this.storedMeasures[newMeasureId] = this.storedMeasures[oldMeasureId];
    this.storedMeasuresRules[newMeasureId] = this.storedMeasuresRules[oldMeasureId];
    delete this.storedMeasures[oldMeasureId];
    delete this.storedMeasuresRules[oldMeasureId];
    for (const id of this.computedMeasureIds) {
      const expression = this.computedMeasures[id];
      if (expression.variables().includes(oldMeasureId)) this.computedMeasures[id] = expression.substitute(oldMeasureId, newMeasureId);
    }
  }

  dropMeasure(measureId) {
    if (this.computedMeasures[measureId] !== undefined) {
      delete this.computedMeasures[measureId];
      return;
    }
    if (this.storedMeasures[measureId] === undefined) throw new Error(`dropMeasure: no such measure: ${measureId}`);
    delete this.storedMeasures[measureId];
    delete this.storedMeasuresRules[measureId];
    for (const id of this.computedMeasureIds) if (this.computedMeasures[id].variables().includes(measureId)) delete this.computedMeasures[id];
  }

  dropMeasures(measureIds) {
    measureIds.forEach((id) => this.dropMeasure(id));
  }

  keepMeasure(measureId) {
    this.keepMeasures([measureId]);
  }

  keepMeasures(measureIds) {
    this.computedMeasureIds.concat(this.storedMeasureIds).filter((id) => !measureIds.includes(id)).forEach((id) => {
      if (this.computedMeasures[id] !== undefined || this.storedMeasures[id] !== undefined) this.dropMeasure(id);
    });
  }

  updateStoredMeasureRules(measureId, cb) {
    this.storedMeasuresRules[measureId] = cb(this.storedMeasuresRules[measureId]);
  }

  clone(measures = []) {
    // dimension objects are immutable for every query, so the copy may share them
    const copy = new Cube(this.dimensions.slice());
    const wanted = (id) => measures.length === 0 || measures.includes(id);
    for (const id of this.computedMeasureIds.filter(wanted)) copy.computedMeasures[id] = this.computedMeasures[id];
    for (const id of this.storedMeasureIds.filter(wanted)) {
      copy.storedMeasures[id] = this.storedMeasures[id].clone();
      copy.storedMeasuresRules[id] = deepCopy(this.storedMeasuresRules[id]);
    }
    return copy;
  }

  // ------------------------------------------------------------------ cell access
  _store(measureId, caller) {
    const store = this.storedMeasures[measureId];
    if (store === undefined) throw new Error(`${caller}: no such measure ${measureId}`);
    return store;
  }

  getData(measureId) {
    if (this.computedMeasures[measureId] !== undefined && this.storedMeasures[measureId] === undefined) return this._evaluateComputed(measureId);
    return this._store(measureId, 'getData').data;
  }

  getStatusMap(measureId) {
    if (this.storedMeasures[measureId] !== undefined) return this.storedMeasures[measureId]._dataMap;
    if (this.computedMeasures[measureId] === undefined) throw new Error(`getStatusMap: no such measure ${measureId}`);
    // src/cube.js:373-386: the union of every stored measure's keys, values OR-ed together
    const result = new Map();
    for (const store of Object.values(this.storedMeasures))
      for (const [key, value] of store._dataMap.entries()) result.set(key, result.get(key) ? result.get(key) | value : value);
    return result;
  }

  getTotal(measureId) {
    return this.storedMeasures[measureId].total;
  }

  fillData(measureId, value) {
    if (!this.storedMeasures[measureId]) throw new Error(`fillData can only be called on stored measures: ${measureId}`);
    this.storedMeasures[measureId].fill(value);
  }

  setData(measureId, values) {
    if (!this.storedMeasures[measureId]) throw new Error(`setData can only be called on stored measures: ${measureId}`);
    this.storedMeasures[measureId].data = values;
  }

  getNestedArray(measureId) {
    return toNestedArray(this.getData(measureId), this.dimensions);
  }

  setNestedArray(measureId, values) {
    this.setData(measureId, fromNestedArray(values, this.dimensions));
  }

  /** With totals: the 2^D marginal cubes (drillUp to 'all' on every subset of dimensions) merged. */
  getNestedObject(measureId, withTotals = false) {
    return this.getNestedObjects([measureId], withTotals)[measureId];
  }

  getNestedObjects(measureIds, withTotals = false) {
    const plain = (cube, ids) => {
      const out = {};
      for (const id of ids) out[id] = toNestedObject(cube.getData(id), cube.dimensions);
      return out;
    };
    // eslint-disable-next-line eqeqeq
    if (!withTotals || this.dimensions.length == 0) return plain(this, measureIds);
    // The reference rebuilds every marginal from the full cube: 2^D chains of drillUp(dim, 'all')
    // (src/cube.js:429-437), merged into one object whose levels carry an extra 'all' key.  All 2^D
    // results are the cells of ONE extended cube with an 'all' item appended to every dimension; a stored
    // measure gets it from a single store call (olap_store_totals: one launch that reads the cube once
    // when the extended cube fits in LDS, D + 2 launches otherwise) — same chain order, same rounding.
    // (a measure that tracks its insertion order takes the chain too: every marginal has an order of its own)
    const direct = (id) => this.storedMeasures[id] !== undefined && !this.storedMeasures[id].orderTracked;
    const stored = measureIds.filter(direct);
    const others = measureIds.filter((id) => !direct(id));
    const extended = this.dimensions.map((d) => ({ getItems: () => d.getItems().concat(['all']) }));
    let result = {};
    for (const id of stored) {
      const rules = this.storedMeasuresRules[id] || {};
      result[id] = toNestedObject(this.storedMeasures[id].totals(this.dimensions, this.dimensions.map((d) => rules[d.id])), extended);
    }
    if (others.length) {
      // computed measures are evaluated on each marginal cube (their `__total` parameters are that cube's
      // totals).  The chain of subset s is the chain of (s without its highest dimension) plus one
      // drillUp, so marginals are memoised: same order of operations, same values.
      const marginals = [this];
      for (let subset = 0; subset < 2 ** this.dimensions.length; ++subset) {
        if (subset > 0) {
          const top = 31 - Math.clz32(subset);
          marginals[subset] = marginals[subset & ~(1 << top)].drillUp(this.dimensions[top].id, 'all');
        }
        result = deepMerge(result, plain(marginals[subset], others));
      }
    }
    return result;
  }

  setNestedObject(measureId, value) {
    this.setData(measureId, fromNestedObject(value, this.dimensions));
  }

  hydrateFromSparseNestedObject(measureId, obj, offset = 0, depth = 0) {
    if (depth === this.dimensions.length) {
      this.storedMeasures[measureId].setValue(offset, obj);
      return;
    }
    const dimension = this.dimensions[depth];
    for (const key in obj) {
      const at = dimension.getRootIndexFromRootItem(key);
      if (at !== -1) this.hydrateFromSparseNestedObject(measureId, obj[key], offset * dimension.numItems + at, depth + 1);
    }
  }

  getPosition(coords) {
    let position = 0;
    for (const dimension of this.dimensions) {
      const item = coords[dimension.id];
      if (item === undefined) throw new Error(`getPosition: no such dimension ${dimension.id}. Coords: ${JSON.stringify(coords)}`);
      const at = dimension.getRootIndexFromRootItem(item);
      if (at === -1) throw new Error(`getPosition: no such item ${item}. Dimension items: ${dimension.getItems()}`);
      position = position * dimension.numItems + at;
    }
    return position;
  }

  _checkCoords(caller, coords) {
    if (this.dimensionIds.some((id) => !coords[id]))
      throw new Error(`${caller}: no value for all dimensions. Dimensions: ${this.dimensionIds}, Coords: ${JSON.stringify(coords)}`);
  }

  setSingleData(measureId, coords, value) {
    this._checkCoords('setSingleData', coords);
    if (this.storedMeasures[measureId] === undefined) throw new Error(`setSingleData: no such stored measure ${measureId}`);
    this.storedMeasures[measureId].setValue(this.getPosition(coords), value);
  }

  getSingleData(measureId, coords) {
    this._checkCoords('getSingleData', coords);
    const position = this.getPosition(coords);
    if (this.storedMeasures[measureId] !== undefined) return this.storedMeasures[measureId].getValue(position);
    if (this.computedMeasures[measureId] === undefined) throw new Error(`getSingleData: no such measure ${measureId}`);
    const expression = this.computedMeasures[measureId];
    const params = {};
    for (const name of expression.variables({ withMembers: true })) params[name] = this.storedMeasures[name].getValue(position);
    return expression.evaluate(params);
  }

  _combinations(dimensionsFilter) {
    const options = {};
    for (const [id, value] of Object.entries(dimensionsFilter)) options[id] = typeof value === 'string' ? [value] : value;
    for (const id of this.dimensionIds) if (dimensionsFilter[id] === undefined) options[id] = this.getDimension(id).getItems();
    return cartesian(options);
  }

  getTotalForDimensionItems(measureId, dimensionsFilter = {}) {
    return this._combinations(dimensionsFilter).reduce((sum, coords) => sum + this.getSingleData(measureId, coords), 0);
  }

  getDistribution(measureId, dimensionsFilter = {}) {
    const part = this.getTotalForDimensionItems(measureId, dimensionsFilter);
    const whole = this.getTotal(measureId);
    return whole === 0 ? part : part / whole;
  }

  copyMeasureData(sourceMeasureId, targetMeasureId, dimensionsFilter = {}) {
    for (const coords of this._combinations(dimensionsFilter)) this.setSingleData(targetMeasureId, coords, this.getSingleData(sourceMeasureId, coords));
  }

  // ------------------------------------------------------------------ the derivation helper
  /**
   * Builds the cube that results from replacing the dimension list and running `storeOp` on each
   * stored measure.  `rules`: 'share' (same object, src/cube.js:1011) or a ready-made object.
   */
  _derive(newDimensions, storeOp, rules = 'share', measureFilter = null) {
    const cube = new Cube(newDimensions);
    Object.assign(cube.computedMeasures, this.computedMeasures);
    if (rules === 'share') Object.assign(cube.storedMeasuresRules, this.storedMeasuresRules);
    else cube.storedMeasuresRules = rules;
    for (const id of this.storedMeasureIds) {
      if (measureFilter && !measureFilter(id)) continue;
      cube.storedMeasures[id] = storeOp(this.storedMeasures[id], id);
    }
    return cube;
  }

  _withDimension(index, dimension) {
    const list = this.dimensions.slice();
    list[index] = dimension;
    return list;
  }

  // ------------------------------------------------------------------ drillUp / drillDown
  /** Aggregate one dimension by a parent attribute (minutes by hour, cities by region). */
  drillUp(dimensionId, attribute) {
    const index = this.getDimensionIndex(dimensionId);
    const current = this.dimensions[index];
    if (current.rootAttribute === attribute) return this;
    const rolled = current.drillUp(attribute);
    // eslint-disable-next-line eqeqeq
    if (rolled == current) {
      console.info(`drillUp: no such attribute: ${attribute} in dimension: ${dimensionId} in cube: ${this.dimensionIds.join(', ')}`);
      return this;
    }
    const newDimensions = this._withDimension(index, rolled);
    // The reference rolls its measures up one store call at a time (src/cube.js:1012-1020); here the stored measures go
    // to the device together, each with its rule for this dimension (HipStore.drillUpMany: one launch where possible).
    const ids = this.storedMeasureIds;
    const rolledUp = {};
    const Store = ids.length ? this.storedMeasures[ids[0]].constructor : null;
    if (ids.length >= 2 && typeof Store.drillUpMany === 'function') {
      const results = Store.drillUpMany(ids.map((id) => this.storedMeasures[id]), this.dimensions, newDimensions, ids.map((id) => this.storedMeasuresRules[id][dimensionId]));
      ids.forEach((id, i) => {
        rolledUp[id] = results[i];
      });
    }
    return this._derive(newDimensions, (store, id) => rolledUp[id] || store.drillUp(this.dimensions, newDimensions, this.storedMeasuresRules[id][dimensionId]));
  }

  drillDown(dimensionId, attribute) {
    const index = this.getDimensionIndex(dimensionId);
    const current = this.dimensions[index];
    if (current.rootAttribute === attribute) return this;
    const refined = current.drillDown(attribute);
    // eslint-disable-next-line eqeqeq
    if (refined == current) return this;
    const newDimensions = this._withDimension(index, refined);
    return this._derive(newDimensions, (store, id) => store.drillDown(this.dimensions, newDimensions, this.storedMeasuresRules[id][dimensionId]));
  }

  // ------------------------------------------------------------------ dice / slice
  _diced(index, dimension) {
    // eslint-disable-next-line eqeqeq
    if (dimension == this.dimensions[index]) return this;
    const newDimensions = this._withDimension(index, dimension);
    return this._derive(newDimensions, (store) => store.dice(this.dimensions, newDimensions));
  }

  dice(dimensionId, attribute, items, reorder = false) {
    const index = this.getDimensionIndex(dimensionId);
    return this._diced(index, this.dimensions[index].dice(attribute, items, reorder));
  }

  diceRange(dimensionId, attribute, start, end) {
    const index = this.getDimensionIndex(dimensionId);
    return this._diced(index, this.dimensions[index].diceRange(attribute, start, end));
  }

  slice(dimensionId, attribute, value) {
    if (this.getDimensionIndex(dimensionId) === -1) throw new Error(`slice: no such dimension: ${dimensionId}`);
    return this.dice(dimensionId, attribute, [value]).removeDimension(dimensionId);
  }

  collapse() {
    // When every roll-up rule is 'sum', slicing each dimension to 'all' in turn (src/cube.js:320-324)
    // adds up every set cell: that is the store's `total` (in-memory.js:22-28) — one float64
    // reduction per measure instead of one launch per dimension, and without the per-stage rounding
    // to the cell type that a chain of typed stores would add.
    const additive = this.storedMeasureIds.every((id) => this.dimensionIds.every((dimId) => (this.storedMeasuresRules[id][dimId] || 'sum') === 'sum'));
    if (!additive || this.dimensions.length === 0) return this.dimensionIds.reduce((cube, id) => cube.slice(id, 'all', 'all'), this);
    const cube = new Cube([]);
    Object.assign(cube.computedMeasures, this.computedMeasures);
    const rules = deepCopy(this.storedMeasuresRules);
    for (const id of Object.keys(rules)) for (const dimId of this.dimensionIds) delete rules[id][dimId];
    cube.storedMeasuresRules = rules;
    for (const id of this.storedMeasureIds) {
      const source = this.storedMeasures[id];
      const cell = new HipStore(1, source._type, source._defaultValue);
      // under a NaN default an empty measure stays unset; under 0 a zero total unsets itself
      if (!Number.isNaN(source._defaultValue) || source._dataMap.size > 0) cell.setValue(0, source.total);
      cube.storedMeasures[id] = cell;
    }
    return cube;
  }

  aggregateByDimensions(excludeDimensionIds) {
    return this.dimensionIds.filter((id) => !excludeDimensionIds.includes(id)).reduce((cube, id) => cube.slice(id, 'all', 'all'), this);
  }

  getDimensionItemsMap(dimensionIds) {
    const out = {};
    for (const id of this.dimensionIds) if (dimensionIds == null || dimensionIds.includes(id)) out[id] = this.getDimension(id).getItems();
    return out;
  }

  diceByDimensionItems(dimensionItemsMap, measures = [], reorder = false) {
    const newDimensions = this.dimensions.slice();
    for (const [dimensionId, items] of Object.entries(dimensionItemsMap)) {
      const index = this.getDimensionIndex(dimensionId);
      if (index === -1) continue;
      const list = Array.isArray(items) ? items : [items];
      // a dimension literally called 'time' is diced at the periodicity of the given slots (src/cube.js:600-603)
      const attribute = dimensionId === 'time' ? TimeSlot.fromValue(list[0]).periodicity : this.dimensions[index].rootAttribute;
      newDimensions[index] = newDimensions[index].dice(attribute, list, reorder);
    }
    if (newDimensions.every((d, i) => d === this.dimensions[i])) return this;
    const rules = {};
    const wanted = (id) => measures.length === 0 || measures.includes(id);
    for (const id of this.storedMeasureIds.filter(wanted)) rules[id] = deepCopy(this.storedMeasuresRules[id]);
    return this._derive(newDimensions, (store) => store.dice(this.dimensions, newDimensions), rules, wanted);
  }

  scan(dimensionIds, cb) {
    for (const combination of cartesian(this.getDimensionItemsMap(dimensionIds))) cb(this.diceByDimensionItems(combination), combination);
  }

  iterateOverDimension(dimension, cb) {
    const others = this.dimensionIds.filter((id) => id !== dimension);
    if (others.length === this.dimensionIds.length) throw new Error(`Cube has no ${dimension} dimension. Dimensions: ${this.dimensionIds}`);
    if (others.length === 0) {
      cb(this, {});
      return;
    }
    this.scan(others, (diced, items) => cb(diced.aggregateByDimensions([dimension]), items));
  }

  // ------------------------------------------------------------------ dimension list changes
  removeDimension(dimensionId) {
    const rules = deepCopy(this.storedMeasuresRules);
    for (const id of Object.keys(rules)) delete rules[id][dimensionId];
    // the 'all' axis has one item, so dropping it from the list leaves the flat layout unchanged
    const totals = this.drillUp(dimensionId, 'all').storedMeasures;
    const cube = new Cube(this.dimensions.filter((d) => d.id !== dimensionId));
    cube.storedMeasures = totals;
    Object.assign(cube.computedMeasures, this.computedMeasures);
    cube.storedMeasuresRules = rules;
    return cube;
  }

  removeDimensions(dimensionIds) {
    return dimensionIds.reduce((cube, id) => cube.removeDimension(id), this);
  }

  keepDimensions(dimensionIds) {
    return this.dimensionIds.filter((id) => !dimensionIds.includes(id)).reduce((cube, id) => cube.removeDimension(id), this);
  }

  /** Insert a dimension: drillDown from a one-item placeholder to the new items (src/cube.js:910-948). */
  addDimension(newDimension, aggregation = {}, index = null, distributions = {}) {
    const at = index === null ? this.dimensions.length : index;
    const oldDimensions = this.dimensions.slice();
    oldDimensions.splice(at, 0, new CatchAllDimension(newDimension.id, newDimension));
    const newDimensions = oldDimensions.slice();
    newDimensions[at] = newDimension;
    const rules = deepCopy(this.storedMeasuresRules);
    for (const id of Object.keys(rules)) rules[id][newDimension.id] = aggregation[id];
    return this._derive(newDimensions, (store, id) => store.drillDown(oldDimensions, newDimensions, aggregation[id], distributions[id]), rules);
  }

  reorderDimensions(dimensionIds) {
    if (this.dimensions.every((d, i) => dimensionIds[i] === d.id)) return this;
    const newDimensions = dimensionIds.map((id) => this.dimensions.find((d) => d.id === id));
    return this._derive(newDimensions, (store) => store.reorder(this.dimensions, newDimensions));
  }

  swapDimensions(dim1, dim2) {
    for (const id of [dim1, dim2]) if (!this.dimensionIds.includes(id)) throw new Error(`swapDimensions: no such dimension ${id}`);
    return this.reorderDimensions(this.dimensionIds.map((id) => (id === dim1 ? dim2 : id === dim2 ? dim1 : id)));
  }

  project(dimensionIds) {
    return this.keepDimensions(dimensionIds).reorderDimensions(dimensionIds);
  }

  // ------------------------------------------------------------------ cube to cube ("next" rows)
  /** Brings this cube onto `targetDims`: project, add missing dims, drill to the target roots, dice. */
  reshape(targetDims) {
    const mine = this.dimensionIds;
    let cube = this.project(targetDims.filter((d) => mine.includes(d.id)).map((d) => d.id));
    targetDims.forEach((target, i) => {
      const actual = cube.dimensions[i];
      if (!actual || actual.id !== target.id) cube = cube.addDimension(target, {}, i);
    });
    targetDims.forEach((target, i) => {
      const actual = cube.dimensions[i];
      if (actual.rootAttribute === target.rootAttribute) return;
      if (actual.attributes.includes(target.rootAttribute)) cube = cube.drillUp(target.id, target.rootAttribute);
      else if (target.attributes.includes(actual.rootAttribute)) cube = cube.drillDown(target.id, target.rootAttribute);
      else throw new Error(`The cube dimensions '${target.id}' are not compatible.`);
      cube = cube.dice(target.id, target.rootAttribute, target.getItems(), true);
    });
    return cube;
  }

  hydrateFromCube(otherCube) {
    let compatible;
    try {
      compatible = otherCube.reshape(this.dimensions);
    } catch (_e) {
      return; // no overlap between the cubes: nothing to load
    }
    for (const id of this.storedMeasureIds) {
      const source = compatible.storedMeasures[id];
      if (source) this.storedMeasures[id].load(source, this.dimensions, compatible.dimensions);
    }
  }

  compose(otherCube, union = false, fillWith = null) {
    const newDimensions = [];
    for (const dimension of this.dimensions) {
      const other = otherCube.getDimension(dimension.id);
      if (other) newDimensions.push(union ? dimension.union(other) : dimension.intersect(other));
    }
    const cube = new Cube(newDimensions);
    for (const source of [this, otherCube]) {
      for (const id of source.storedMeasureIds) {
        const store = source.storedMeasures[id];
        cube.createStoredMeasure(id, source.storedMeasuresRules[id], store._type, store._defaultValue);
        if (fillWith && fillWith[id]) cube.fillData(id, fillWith[id]);
        cube.hydrateFromCube(source);
      }
    }
    return cube;
  }

  // ------------------------------------------------------------------ wire format ("next" row f3)
  /** Same container as the reference (src/cube.js:1135-1151); computed measures are not carried. */
  serialize() {
    return toBuffer({
      dimensions: this.dimensions.map((d) => d.serialize()),
      storedMeasuresKeys: this.storedMeasureIds,
      storedMeasures: Object.values(this.storedMeasures).map((store) => store.serialize()),
      storedMeasuresRules: this.storedMeasuresRules,
      computedMeasures: this.computedMeasureIds.reduce((out, id) => {
        out[id] = this.computedMeasures[id].toString();
        return out;
      }, {}),
    });
  }

  serializeToBase64String() {
    return Buffer.from(this.serialize()).toString('base64');
  }

  static deserialize(buffer) {
    const GenericDimension = require('./dimension/generic');
    const TimeDimension = require('./dimension/time');
    const data = fromBuffer(buffer);
    // a time dimension's record carries `start` (src/dimension/factory.js:6-12)
    const cube = new Cube(data.dimensions.map((blob) => (fromBuffer(blob).start ? TimeDimension.deserialize(blob) : GenericDimension.deserialize(blob))));
    cube.storedMeasuresRules = data.storedMeasuresRules || {};
    data.storedMeasuresKeys.forEach((id, i) => {
      cube.storedMeasures[id] = HipStore.deserialize(data.storedMeasures[i]);
      // (a blob whose cells are not listed in ascending order keeps that order by itself, olap_store_from_sparse)
      if (Object.values(cube.storedMeasuresRules[id] || {}).some((rule) => rule === 'first' || rule === 'last')) cube.storedMeasures[id].trackOrder();
    });
    for (const id of Object.keys(data.computedMeasures || {})) cube.computedMeasures[id] = getParser().parse(data.computedMeasures[id]);
    return cube;
  }

  static deserializeFromBase64String(text) {
    return Cube.deserialize(toArrayBuffer(Buffer.from(text, 'base64')));
  }
}

module.exports = Cube;
