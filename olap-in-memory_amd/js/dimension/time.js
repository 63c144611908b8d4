'use strict';
/*
 * TimeDimension: the days from `start` to `end`, looked at with some periodicity (day, week_*,
 * month_week_*, month, quarter, semester, year).  Nothing is stored but the two bounding days; item
 * lists and the root -> group index maps are computed from calendar arithmetic when first asked for.
 *
 * Public behaviour as /root/reference/src/dimension/time.js (constructor :16-26, getItems :49-66,
 * drillUp / drillDown :75-103, dice :105-139, diceRange :141-180,
 * getGroupIndexFromRootIndexMap :182-197, union / intersect :207-248), built on ../calendar.js
 * because the reference's `timeslot-dag` dependency is not available.
 */
const AbstractDimension = require('./abstract');
const TimeSlot = require('../calendar');
const wire = require('../wire');

/** first (edge = 'firstDate') or last day covered by a slot value, as a 'day' slot */
function boundingDay(slotValue, edge) {
  return TimeSlot.fromDate(TimeSlot.fromValue(slotValue)[edge], 'day');
}

/** every slot of `periodicity` that touches [firstDay, lastDay] */
function slotsCovering(firstDay, lastDay, periodicity) {
  const out = [];
  const stop = lastDay.toParentPeriodicity(periodicity).value;
  let slot = firstDay.toParentPeriodicity(periodicity);
  for (;;) {
    out.push(slot.value);
    if (slot.value >= stop) return out;
    slot = slot.next();
  }
}

class TimeDimension extends AbstractDimension {
  constructor(id, rootAttribute, start, end, label = null) {
    super(id, rootAttribute, label);
    this._start = boundingDay(start, 'firstDate');
    this._end = boundingDay(end, 'lastDate');
    if (this._start.periodicity !== 'day' || this._end.periodicity !== 'day') throw new Error('Start and end must be dates.');
    this._itemCache = Object.create(null);
    this._mapCache = Object.create(null);
  }

  get attributes() {
    return [this._rootAttribute, ...TimeSlot.upperSlots[this._rootAttribute]];
  }

  get _isEmpty() {
    return this._start.value > this._end.value; // an intersection that came out empty
  }

  getItems(attribute = null) {
    if (this._isEmpty) return [];
    const periodicity = attribute || this._rootAttribute;
    if (!this._itemCache[periodicity]) this._itemCache[periodicity] = slotsCovering(this._start, this._end, periodicity);
    return this._itemCache[periodicity];
  }

  getEntries(attribute = null, language = 'en') {
    return this.getItems(attribute).map((value) => [value, TimeSlot.fromValue(value).humanizeValue(language)]);
  }

  /** root index -> index of the enclosing slot of `attribute` (a plain Array, like the reference's) */
  getGroupIndexFromRootIndexMap(attribute) {
    if (this._mapCache[attribute] === undefined) {
      this._checkAttribute(attribute);
      const indexOfGroup = this.getItemsToIdx(attribute);
      this._mapCache[attribute] = this.getItems().map((value) => indexOfGroup[TimeSlot.fromValue(value).toParentPeriodicity(attribute).value]);
    }
    return this._mapCache[attribute];
  }

  getGroupIndexFromRootIndex(attribute, rootIndex) {
    return this.getGroupIndexFromRootIndexMap(attribute)[rootIndex];
  }

  /** same days, another periodicity (or other days) */
  _like(periodicity, firstDay = this._start.value, lastDay = this._end.value) {
    return new TimeDimension(this.id, periodicity, firstDay, lastDay, this.label);
  }

  drillUp(periodicity) {
    // eslint-disable-next-line eqeqeq
    if (periodicity == this.rootAttribute) return this;
    return this._like(periodicity);
  }

  drillDown(periodicity) {
    // eslint-disable-next-line eqeqeq
    if (periodicity == this.rootAttribute) return this;
    const coarserThanTarget = TimeSlot.upperSlots[periodicity].includes(this._rootAttribute);
    if (!coarserThanTarget) throw new Error('Invalid periodicity.');
    return this._like(periodicity);
  }

  /** Keeps the days of [start, end] (slot values of `periodicity`; a missing bound = unbounded). */
  diceRange(periodicity, start, end) {
    if (periodicity === 'all') return this;
    const dayOf = (value, edge) => {
      const slot = TimeSlot.fromValue(value);
      if (slot.periodicity !== periodicity) throw new Error(`${value} is not a valid slot of periodicity ${periodicity}`);
      return TimeSlot.fromDate(slot[edge], 'day').value;
    };
    const mine = [this._start.value, this._end.value];
    const asked = [start ? dayOf(start, 'firstDate') : mine[0], end ? dayOf(end, 'lastDate') : mine[1]];
    if (asked[0] <= mine[0] && mine[1] <= asked[1]) return this; // nothing would be cut
    const first = asked[0] < mine[0] ? mine[0] : asked[0];
    const last = asked[1] < mine[1] ? asked[1] : mine[1];
    return this._like(this._rootAttribute, first, last);
  }

  /** A time dimension stays a range: the items must be consecutive slots of `periodicity`. */
  dice(periodicity, items, reorder = false) {
    if (items.length === 1) return this.diceRange(periodicity, items[0], items[0]);
    const sequence = reorder ? items : [...items].sort();
    let previous = TimeSlot.fromValue(items[0]);
    if (previous.periodicity !== periodicity) throw new Error('Unsupported: wrong periodicity');
    sequence.slice(1).forEach((value) => {
      const slot = TimeSlot.fromValue(value);
      if (slot.periodicity !== periodicity || slot.value !== previous.next().value) throw new Error('Unsupported: follow');
      previous = slot;
    });
    return this.diceRange(periodicity, sequence[0], sequence[sequence.length - 1]);
  }

  _commonAttribute(other) {
    if (this.id !== other.id) throw new Error('Not the same dimension');
    if (this.attributes.includes(other.rootAttribute)) return other.rootAttribute;
    if (other.attributes.includes(this.rootAttribute)) return this.rootAttribute;
    throw new Error('The dimensions are not compatible');
  }

  union(other) {
    const periodicity = this._commonAttribute(other);
    const first = this._start.value < other._start.value ? this._start.value : other._start.value;
    const last = this._end.value > other._end.value ? this._end.value : other._end.value;
    return this._like(periodicity, first, last);
  }

  intersect(other) {
    const periodicity = this._commonAttribute(other);
    // the coarser of the two dimensions is cut to the other's days
    const coarse = periodicity === other.rootAttribute ? other : this;
    const fine = coarse === other ? this : other;
    return coarse.diceRange('day', fine._start.value, fine._end.value);
  }

  /** Same record as the reference (time.js:39-47). */
  serialize() {
    return wire.toBuffer({ id: this.id, label: this.label, rootAttribute: this.rootAttribute, start: this._start.value, end: this._end.value });
  }

  static deserialize(buffer) {
    const record = wire.fromBuffer(buffer);
    return new TimeDimension(record.id, record.rootAttribute, record.start, record.end, record.label);
  }
}

module.exports = TimeDimension;
