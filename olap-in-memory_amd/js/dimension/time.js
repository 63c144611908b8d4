'use strict';
/*
 * TimeDimension — a calendar range [start, end] (days) seen at a periodicity.  Items and the
 * root -> group index maps are derived lazily from calendar arithmetic.  Behaviour follows
 * /root/reference/src/dimension/time.js (ctor :16-26, getItems :49-66, drillUp/Down :75-103,
 * dice :105-139, diceRange :141-180, getGroupIndexFromRootIndexMap :182-197, union/intersect
 * :207-248) on top of ./calendar.js instead of the absent `timeslot-dag`.
 */
const AbstractDimension = require('./abstract');
const TimeSlot = require('../calendar');
const { toBuffer, fromBuffer } = require('../wire');

const dayOf = (value, edge) => TimeSlot.fromDate(TimeSlot.fromValue(value)[edge], 'day');

class TimeDimension extends AbstractDimension {
  constructor(id, rootAttribute, start, end, label = null) {
    super(id, rootAttribute, label);
    this._start = dayOf(start, 'firstDate');
    this._end = dayOf(end, 'lastDate');
    this._items = {};
    this._maps = {};
    if (this._start.periodicity !== 'day' || this._end.periodicity !== 'day') throw new Error('Start and end must be dates.');
  }

  get attributes() {
    return [this._rootAttribute].concat(TimeSlot.upperSlots[this._rootAttribute]);
  }

  getItems(attribute = null) {
    if (this._start.value > this._end.value) return [];
    const attr = attribute || this._rootAttribute;
    let items = this._items[attr];
    if (!items) {
      const last = this._end.toParentPeriodicity(attr).value;
      items = [];
      for (let slot = this._start.toParentPeriodicity(attr); ; slot = slot.next()) {
        items.push(slot.value);
        if (slot.value >= last) break;
      }
      this._items[attr] = items;
    }
    return items;
  }

  getEntries(attribute = null, language = 'en') {
    return this.getItems(attribute).map((item) => [item, TimeSlot.fromValue(item).humanizeValue(language)]);
  }

  _rebased(attribute, start = this._start.value, end = this._end.value) {
    return new TimeDimension(this.id, attribute, start, end, this.label);
  }

  drillUp(attribute) {
    // eslint-disable-next-line eqeqeq
    return attribute == this.rootAttribute ? this : this._rebased(attribute);
  }

  drillDown(attribute) {
    // eslint-disable-next-line eqeqeq
    if (attribute == this.rootAttribute) return this;
    if (!TimeSlot.upperSlots[attribute].includes(this._rootAttribute)) throw new Error('Invalid periodicity.');
    return this._rebased(attribute);
  }

  /** Only contiguous runs of slots can be kept: a time dimension is a range. */
  dice(attribute, items, reorder = false) {
    if (items.length === 1) return this.diceRange(attribute, items[0], items[0]);
    const ordered = reorder ? items : items.slice().sort();
    let previous = TimeSlot.fromValue(items[0]);
    if (previous.periodicity !== attribute) throw new Error('Unsupported: wrong periodicity');
    for (let i = 1; i < ordered.length; ++i) {
      const slot = TimeSlot.fromValue(ordered[i]);
      if (slot.periodicity !== attribute || slot.value !== previous.next().value) throw new Error('Unsupported: follow');
      previous = slot;
    }
    return this.diceRange(attribute, ordered[0], ordered[ordered.length - 1]);
  }

  diceRange(attribute, start, end) {
    if (attribute === 'all') return this;
    const bound = (value, edge, fallback) => {
      if (!value) return fallback;
      const slot = TimeSlot.fromValue(value);
      if (slot.periodicity !== attribute) throw new Error(`${value} is not a valid slot of periodicity ${attribute}`);
      return TimeSlot.fromDate(slot[edge], 'day').value;
    };
    const from = bound(start, 'firstDate', this._start.value);
    const to = bound(end, 'lastDate', this._end.value);
    if (from <= this._start.value && this._end.value <= to) return this;
    return this._rebased(this._rootAttribute, from < this._start.value ? this._start.value : from, to < this._end.value ? to : this._end.value);
  }

  /** root index -> index of the enclosing slot of `attribute`; a plain Array as in the reference. */
  getGroupIndexFromRootIndexMap(attribute) {
    if (this._maps[attribute] === undefined) {
      this._checkAttribute(attribute);
      const position = this.getItemsToIdx(attribute);
      this._maps[attribute] = this.getItems().map((item) => position[TimeSlot.fromValue(item).toParentPeriodicity(attribute).value]);
    }
    return this._maps[attribute];
  }

  getGroupIndexFromRootIndex(attribute, rootIndex) {
    return this.getGroupIndexFromRootIndexMap(attribute)[rootIndex];
  }

  union(other) {
    if (this.id !== other.id) throw new Error('Not the same dimension');
    let attribute;
    if (this.attributes.includes(other.rootAttribute)) attribute = other.rootAttribute;
    else if (other.attributes.includes(this.rootAttribute)) attribute = this.rootAttribute;
    else throw new Error('The dimensions are not compatible');
    const start = this._start.value < other._start.value ? this._start.value : other._start.value;
    const end = other._end.value < this._end.value ? this._end.value : other._end.value;
    return this._rebased(attribute, start, end);
  }

  intersect(other) {
    if (this.id !== other.id) throw new Error('Not the same dimension');
    if (this.attributes.includes(other.rootAttribute)) return other.diceRange('day', this._start.value, this._end.value);
    if (other.attributes.includes(this.rootAttribute)) return this.diceRange('day', other._start.value, other._end.value);
    throw new Error('The dimensions are not compatible');
  }

  /** Same record as the reference (time.js:39-47). */
  serialize() {
    return toBuffer({ id: this.id, label: this.label, rootAttribute: this.rootAttribute, start: this._start.value, end: this._end.value });
  }

  static deserialize(buffer) {
    const data = fromBuffer(buffer);
    return new TimeDimension(data.id, data.rootAttribute, data.start, data.end, data.label);
  }
}

module.exports = TimeDimension;
