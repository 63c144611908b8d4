'use strict';
/*
 * HipStore — drop-in for the reference's InMemoryStore (/root/reference/src/store/in-memory.js)
 * whose cells live in MI355X HBM.  Same constructor, members and error messages as the members
 * `Cube` touches (SURVEY.md §8(b)): size, byteLength, total, data (get/set), clone, getValue,
 * setValue, fill, drillUp, drillDown, dice, reorder, load, _type, _defaultValue, _dataMap.
 *
 * This class only turns dimension objects into the small integer tables the C ABI takes
 * (include/olap_hip.h) and forwards; all cell work happens in the HIP kernels.
 */
const backend = require('../backend');
const { toBuffer, fromBuffer } = require('../wire');

const TYPE_CODE = { int32: 0, uint32: 1, float32: 2, float64: 3 };
const TYPE_NAME = ['int32', 'uint32', 'float32', 'float64'];
const BYTES = { int32: 4, uint32: 4, float32: 4, float64: 8 };
const TYPED_ARRAY = { int32: Int32Array, uint32: Uint32Array, float32: Float32Array, float64: Float64Array };

/**
 * The element type of the device cells of a measure DECLARED `type`.  The reference's Map holds plain
 * float64 numbers whatever the declared type and coerces only in serialize() (in-memory.js:77-92): an
 * int32 `average` of 7 and 8 is 7.5 until then, a uint32 sum may pass 2^32.  Integer measures therefore
 * live in Float64 cells (8 bytes per cell) and give the reference's values exactly; the declared type
 * still decides byteLength (:15), the remainder rule of drillDown (:343) and the serialized form.
 * backend.setCompactIntegers(true) (or OLAP_COMPACT_INT=1) stores them as 4-byte Int32 / Uint32 cells
 * instead — half the memory and traffic, values coerced after every operation.  Float32 measures are
 * Float32 cells (the tolerance the port states: 1e-5 relative).
 */
const cellTypeOf = (type) => ((type === 'int32' || type === 'uint32') && !backend.compactIntegers() ? 'float64' : type);

const lengthsOf = (dimensions) => {
  const out = new Uint32Array(dimensions.length);
  for (let i = 0; i < dimensions.length; ++i) out[i] = dimensions[i].numItems;
  return out;
};

// A dimension's index map as the Uint32Array the addon takes.  GenericDimension hands out its own Uint32Array (used as it
// is: the addon reads it during the call and keeps nothing); TimeDimension a cached plain Array, converted once per array
// (Uint32Array.from walks the iterator protocol: ~1 us per 10 items on Node 12, per dimension, per call).
const convertedMaps = new WeakMap();
const asU32 = (map) => {
  if (map instanceof Uint32Array) return map;
  let typed = convertedMaps.get(map);
  if (typed === undefined || typed.length !== map.length) {
    typed = new Uint32Array(map.length);
    for (let i = 0; i < map.length; ++i) typed[i] = map[i];
    convertedMaps.set(map, typed);
  }
  return typed;
};

// Array.from(typedArray) walks the iterator protocol (60 ms for 5e5 cells on Node 12); index loops are 20x faster
const toPlainArray = (typed) => {
  const out = new Array(typed.length);
  for (let i = 0; i < typed.length; ++i) out[i] = typed[i];
  return out;
};
const toFloat64 = (values, unset) => {
  const out = new Float64Array(values.length);
  for (let i = 0; i < values.length; ++i) {
    const v = values[i];
    out[i] = v === undefined || v === null ? unset : Number(v);
  }
  return out;
};

/** Read-only view with the Map methods the reference's callers use on `_dataMap`. */
class CellMapView {
  constructor(store) {
    this._store = store;
  }

  get size() {
    return this._store._whole.countSet();
  }

  has(index) {
    return this._store._native.getValue(index) !== undefined;
  }

  get(index) {
    return this._store._native.getValue(index);
  }

  *keys() {
    for (const k of this._store._whole.getKeys()) yield k;
  }

  *values() {
    const whole = this._store._whole;
    const data = whole.getDataF64();
    for (const k of whole.getKeys()) yield data[k];
  }

  *entries() {
    const whole = this._store._whole;
    const data = whole.getDataF64();
    for (const k of whole.getKeys()) yield [k, data[k]];
  }

  [Symbol.iterator]() {
    return this.entries();
  }
}

/**
 * Of two new items naming the same old item only the LAST receives the cells (the reference builds
 * `new Map(newItems.map((item, i) => [oldIdx, i]))`, in-memory.js:219-224): earlier ones become -1.
 */
function effectiveSelection(sel) {
  const last = new Map();
  sel.forEach((old, j) => {
    if (old >= 0) last.set(old, j);
  });
  return Int32Array.from(sel, (old, j) => (old >= 0 && last.get(old) === j ? old : -1));
}

/**
 * A pending selection keeps every dimension of its SOURCE store; the cube may meanwhile have
 * dropped dimensions that were reduced to one item (slice = dice to one item + removeDimension).
 * Returns, for each dimension the caller still sees, its position among the pending ones — the
 * others are the dropped single-item dimensions — or null when the extents do not line up.
 */
function visibleDims(pending, lengths) {
  const at = [];
  let v = 0;
  for (let d = 0; d < pending.midLen.length; ++d) {
    if (v < lengths.length && pending.midLen[d] === lengths[v]) at[v++] = d;
    else if (pending.midLen[d] !== 1) return null;
  }
  return v === lengths.length ? at : null;
}

/**
 * Calls `native[method](...args)`.  A sharded measure answers in place whatever leaves its outermost
 * dimension alone (and the roll-up of that dimension itself: one collective); for the rest the C ABI
 * refuses with a message starting "sharded:" (include/olap_hip.h) and the measure is gathered onto one
 * device first — same result, one copy more.
 */
function onShards(native, method, args) {
  if (!native.isSharded) return native[method](...args);
  try {
    return native[method](...args);
  } catch (e) {
    if (!/^sharded:/.test(e.message)) throw e;
    return native.gather()[method](...args);
  }
}

class HipStore {
  /**
   * `native` is the addon Store, or — for the result of dice() — a pending selection
   * `{ source, oldLen, sel }` that is only materialised when cells are actually needed: a drillUp
   * that follows (slice, removeDimension, drillUp after dice) runs fused and never writes the diced
   * intermediate cube (K5, DESIGN.md §3).
   */
  constructor(size, type = 'float32', defaultValue = Number.NaN, native = undefined, lengths = undefined) {
    // same checks, order and messages as in-memory.js:56-60
    if (!Number.isNaN(defaultValue) && defaultValue !== 0) throw new Error('Invalid default value, only NaN and 0 are supported');
    if (!Object.prototype.hasOwnProperty.call(TYPE_CODE, type)) throw new Error('Invalid type');
    this._size = size;
    this._type = type;
    this._defaultValue = defaultValue;
    this._pending = null;
    this._lent = false; // a pending dice elsewhere still reads this store's device buffer
    if (native && native.source) {
      this._pending = native;
      this._nativeStore = null;
      this._cells = TYPE_NAME[native.source.dtype];
    } else {
      this._nativeStore = native || HipStore._create(size, cellTypeOf(type), defaultValue, lengths);
      this._cells = TYPE_NAME[this._nativeStore.dtype];
    }
    this._dataMap = new CellMapView(this);
  }

  /**
   * A new device store.  When a device list is set (backend.setDevices / OLAP_DEVICES) and the caller says
   * how the cells are laid out (`lengths`, what Cube passes), the measure is split along dimension 0 over
   * those devices; the reference's bare `new Store(size, type, default)` stays on one device.
   */
  static _create(size, type, defaultValue, lengths) {
    const addon = backend.load();
    const def = Number.isNaN(defaultValue) ? 1 : 0;
    const world = addon.shardWorld();
    if (world >= 2 && lengths && lengths.length >= 1 && lengths[0] >= world) return new addon.ShardedStore(Uint32Array.from(lengths), TYPE_CODE[type], def);
    return new addon.Store(size, TYPE_CODE[type], def);
  }

  /** The device store (one device, or sharded); a pending dice is executed on first use. */
  get _native() {
    if (!this._nativeStore) {
      const p = this._pending;
      this._nativeStore = onShards(p.source, 'dice', [p.oldLen, p.midLen, p.sel]);
      this._pending = null;
    }
    return this._nativeStore;
  }

  /**
   * Keep the reference Map's INSERTION order for this measure (in-memory.js:298): `first` / `last`, `_dataMap.keys()`
   * and serialize() then answer exactly as the reference does after out-of-order setValue calls, roll-ups of sparse
   * cubes, permuting dices and reorders.  Costs a second pass per operation once the order leaves the flat index;
   * Cube turns it on for measures with a `first` / `last` rule.  Results of operations inherit it.  (One device only:
   * a sharded measure is gathered first.)
   */
  trackOrder(on = true) {
    if (this._native.isSharded) this._nativeStore = this._native.gather();
    this._writable.trackOrder(on);
    return this;
  }

  get orderTracked() {
    const native = this._pending ? this._pending.source : this._nativeStore;
    return native && !native.isSharded ? native.orderTracked : 0;
  }

  /** The measure as ONE device store: a sharded measure is gathered (what the shards cannot answer in place). */
  get _whole() {
    const native = this._native;
    return native.isSharded ? native.gather() : native;
  }

  /**
   * The device store for writing.  Pending dices read their source lazily, and the reference's
   * dice() returns an independent copy: a store whose buffer has been lent out writes to a fresh
   * copy and leaves the lent one to its readers (copy-on-write, at most once per lending).
   */
  get _writable() {
    const native = this._native;
    if (this._lent) {
      this._nativeStore = native.clone();
      this._lent = false;
    }
    return this._nativeStore;
  }

  _wrap(native) {
    return new HipStore(native.size, this._type, this._defaultValue, native);
  }

  get size() {
    return this._size;
  }

  get byteLength() {
    return this._size * (BYTES[this._type] || 1);
  }

  get total() {
    return this._native.total();
  }

  /** Dense plain Array, default value in unset cells (in-memory.js:30-37). */
  get data() {
    return toPlainArray(this._native.getDataF64());
  }

  set data(values) {
    if (this._size !== values.length) throw new Error(`value length is invalid: ${this._size} !== ${values.length}`);
    const widened = this._cells === 'float64' && (values instanceof Int32Array || values instanceof Uint32Array || values instanceof Float32Array);
    if (ArrayBuffer.isView(values) && !(values instanceof Float64Array) && (values.constructor === TYPED_ARRAY[this._cells] || widened) &&
        !this._native.isSharded) {
      // a typed array of the store's own element type: no conversion; a narrower one into Float64 cells: widened by the addon
      this._writable.setData(values);
      return;
    }
    const d = this._defaultValue;
    // undefined / null unset the cell, exactly like the default value does (in-memory.js:122-133)
    this._writable.setData(toFloat64(values, d));
  }

  clone() {
    return this._wrap(this._native.clone());
  }

  getValue(index) {
    const v = this._native.getValue(index);
    return v === undefined ? this._defaultValue : v;
  }

  setValue(index, value) {
    this._writable.setValue(index, value);
  }

  fill(value) {
    if (value === undefined || value === null) this._writable.fill(this._defaultValue);
    else this._writable.fill(value);
  }

  /**
   * Every marginal of this measure (src/cube.js:421-440) as the flat extended cube: one more item, 'all', at
   * the end of every dimension.  `methods[d]` is the measure's rule for dimension d (default 'sum').
   */
  totals(dimensions, methods) {
    const addon = backend.load();
    const codes = Int32Array.from(dimensions, (_, d) => addon.methodFromName(methods[d])); // throws 'Unsupported aggregation method: <m>'
    return this._whole.totals(lengthsOf(dimensions), codes);
  }

  /** in-memory.js:265-334 */
  drillUp(oldDimensions, newDimensions, method = 'sum') {
    const code = backend.load().methodFromName(method); // throws 'Unsupported aggregation method: <m>'
    const maps = newDimensions.map((dim, i) => asU32(oldDimensions[i].getGroupIndexFromRootIndexMap(dim.rootAttribute)));
    const at = this._pending && !this._pending.source.isSharded ? visibleDims(this._pending, lengthsOf(oldDimensions)) : null;
    if (at) {
      const rolled = maps.filter((map, i) => map.length !== newDimensions[i].numItems || map.some((g, k) => g !== k)).length;
      const p = this._pending;
      // every group has exactly its own single member (e.g. the roll-up to 'all' of a dimension that
      // a slice diced down to one item): the cells do not change, the selection stays pending and
      // keeps composing — slice(...).dice(...).drillUp(...) becomes ONE launch over the source cube
      if (rolled === 0) return new HipStore(this._size, this._type, this._defaultValue, { source: p.source, oldLen: p.oldLen, midLen: p.midLen, sel: p.sel });
      if (rolled <= 1) {
        // dimensions the cube has dropped keep their single item
        const newLen = Uint32Array.from(p.midLen, () => 1);
        const allMaps = Array.from(p.midLen, () => Uint32Array.of(0));
        at.forEach((d, v) => {
          newLen[d] = newDimensions[v].numItems;
          allMaps[d] = maps[v];
        });
        return this._wrap(p.source.diceDrillUp(p.oldLen, p.midLen, newLen, p.sel, allMaps, code));
      }
    }
    return this._wrap(onShards(this._native, 'drillUp', [lengthsOf(oldDimensions), lengthsOf(newDimensions), maps, code]));
  }

  /**
   * drillUp of the stored measures of one cube, each by its own rule — what Cube.drillUp asks of every stored measure
   * in turn (src/cube.js:1012-1020).  Measures held whole on one device go to the device TOGETHER (addon drillUpMulti ->
   * olap_store_drillup_multi: one launch for the measures that share cell type and default — even with different rules
   * when the roll-up streams whole rows — one launch per rule otherwise); the others (pending selections, sharded or
   * order-tracking measures, a lone measure) take drillUp one by one.  Returns the new stores in the order given.
   */
  static drillUpMany(stores, oldDimensions, newDimensions, methods) {
    const out = new Array(stores.length);
    const together = [];
    stores.forEach((store, i) => {
      const native = store._pending ? null : store._nativeStore;
      if (native && !native.isSharded && !native.orderTracked) together.push(i);
    });
    if (together.length >= 2) {
      const addon = backend.load();
      const codes = Int32Array.from(together, (i) => addon.methodFromName(methods[i] === undefined ? 'sum' : methods[i])); // throws 'Unsupported aggregation method: <m>'
      const maps = newDimensions.map((dim, i) => asU32(oldDimensions[i].getGroupIndexFromRootIndexMap(dim.rootAttribute)));
      const natives = addon.drillUpMulti(together.map((i) => stores[i]._nativeStore), codes, lengthsOf(oldDimensions), lengthsOf(newDimensions), maps);
      together.forEach((i, j) => {
        out[i] = stores[i]._wrap(natives[j]);
      });
    }
    stores.forEach((store, i) => {
      if (!out[i]) out[i] = store.drillUp(oldDimensions, newDimensions, methods[i]);
    });
    return out;
  }

  /** in-memory.js:336-430 — any method other than 'sum' copies the parent value (:421-423) */
  drillDown(oldDimensions, newDimensions, method = 'sum', distributions = null) {
    const maps = oldDimensions.map((dim, i) => asU32(newDimensions[i].getGroupIndexFromRootIndexMap(dim.rootAttribute)));
    const weights = distributions ? toFloat64(distributions, Number.NaN) : null;
    // the remainder rule goes by the DECLARED type (:343), whatever the cells are (OLAP_DRILLDOWN_INTEGER_MEASURE)
    const integerMeasure = this._type === 'int32' || this._type === 'uint32' ? 0x100 : 0;
    return this._wrap(onShards(this._native, 'drillDown', [lengthsOf(oldDimensions), lengthsOf(newDimensions), maps, (method === 'sum' ? 0 : 4) | integerMeasure, weights]));
  }

  /** in-memory.js:213-263 — the new dimensions' item ORDER decides where cells land */
  dice(oldDimensions, newDimensions) {
    const sel = newDimensions.map((dim, i) => {
      const position = oldDimensions[i].getItemsToIdx();
      return Int32Array.from(dim.getItems(), (item) => (position[item] === undefined ? -1 : position[item]));
    });
    const midLen = lengthsOf(newDimensions);
    const size = midLen.reduce((n, l) => n * l, 1);
    let composed = sel.map(effectiveSelection);
    const at = this._pending ? visibleDims(this._pending, lengthsOf(oldDimensions)) : null;
    if (at) {
      // dice of a pending dice: compose the selections, still nothing is materialised
      const p = this._pending;
      const all = p.sel.slice();
      const allLen = Uint32Array.from(p.midLen);
      at.forEach((d, v) => {
        all[d] = Int32Array.from(composed[v], (j) => (j < 0 ? -1 : p.sel[d][j]));
        allLen[d] = midLen[v];
      });
      return new HipStore(size, this._type, this._defaultValue, { source: p.source, oldLen: p.oldLen, midLen: allLen, sel: all });
    }
    const source = this._native; // (materialises a pending selection whose dimensions no longer line up)
    // a tracked measure: the diced store has an order of its own that the next operation must see
    if (!source.isSharded && source.orderTracked) return this._wrap(source.dice(lengthsOf(oldDimensions), midLen, composed));
    this._lent = true;
    return new HipStore(size, this._type, this._defaultValue, { source, oldLen: lengthsOf(oldDimensions), midLen, sel: composed });
  }

  /** in-memory.js:178-211 */
  reorder(oldDimensions, newDimensions) {
    const perm = Int32Array.from(newDimensions, (dim) => oldDimensions.indexOf(dim));
    return this._wrap(onShards(this._native, 'reorder', [lengthsOf(oldDimensions), perm]));
  }

  /** in-memory.js:139-176 — mutates this store */
  load(otherStore, myDimensions, hisDimensions) {
    const hisToMine = hisDimensions.map((dim, i) => {
      const position = myDimensions[i].getItemsToIdx();
      return Int32Array.from(dim.getItems(), (item) => (position[item] === undefined ? -1 : position[item]));
    });
    if (otherStore._cells !== this._cells) {
      // the kernels copy cells of one element type; re-type the source through float64 first
      const retyped = new HipStore(otherStore._size, this._type, otherStore._defaultValue);
      retyped.data = otherStore.data;
      otherStore = retyped;
    }
    // hydration scatters cells across the whole index space: a sharded measure is gathered for it and
    // stays on one device afterwards
    if (this._native.isSharded) this._nativeStore = this._native.gather();
    this._writable.load(otherStore._whole, lengthsOf(myDimensions), lengthsOf(hisDimensions), hisToMine);
  }

  /**
   * Same blob as the reference (in-memory.js:75-101): the set cells in sparse form.  The compaction
   * (set cells -> ascending index / value lists) runs on the device.
   */
  serialize() {
    const sparse = this._whole.toSparse();
    // `new Int32Array(map.values())` (in-memory.js:77-92): the coercion to the declared type happens here
    const values = sparse.values.constructor === TYPED_ARRAY[this._type] ? sparse.values : TYPED_ARRAY[this._type].from(sparse.values);
    return toBuffer({ size: this._size, type: this._type, defaultValue: this._defaultValue, indexes: sparse.indexes, dataBuffer: values });
  }

  /** in-memory.js:103-116; accepts blobs written by the reference. */
  static deserialize(buffer) {
    const data = fromBuffer(buffer);
    const type = data.type;
    if (!Object.prototype.hasOwnProperty.call(TYPE_CODE, type)) throw new Error('Invalid type');
    const defaultValue = Number.isNaN(data.defaultValue) ? Number.NaN : 0;
    const cells = cellTypeOf(type);
    const TA = TYPED_ARRAY[cells];
    const values = data.dataBuffer instanceof TA ? data.dataBuffer : TA.from(data.dataBuffer);
    const native = backend.load().storeFromSparse(data.size, TYPE_CODE[cells], Number.isNaN(defaultValue) ? 1 : 0, data.indexes instanceof Uint32Array ? data.indexes : new Uint32Array(data.indexes), values);
    return new HipStore(data.size, type, defaultValue, native);
  }
}

module.exports = HipStore;
module.exports.toPlainArray = toPlainArray;
module.exports._internals = { visibleDims, effectiveSelection }; // host-side logic, unit-tested without a device
