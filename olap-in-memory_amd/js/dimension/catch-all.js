'use strict';
/*
 * The one-item placeholder dimension ('_total') used while a dimension is being added:
 * Cube.addDimension puts it where the new dimension will go, so that "add a dimension" becomes a
 * drillDown from this single item to the new dimension's items
 * (role described at /root/reference/src/cube.js:919-927 and src/dimension/catch-all.js).
 */
const AbstractDimension = require('./abstract');

const ONLY_ITEM = '_total';
const unsupported = () => {
  throw new Error('Unsupported');
};

class CatchAllDimension extends AbstractDimension {
  /** @param childDimension the dimension this placeholder stands for (target of drillDown) */
  constructor(id, childDimension = null) {
    super(id, 'all');
    this.childDimension = childDimension;
    this._single = Object.freeze([ONLY_ITEM]);
    this._zeroMap = new Uint32Array(1);
  }

  getItems() {
    return this._single;
  }

  getEntries() {
    return [[ONLY_ITEM, 'Total']];
  }

  /** Every attribute of a one-item dimension maps that item to group 0. */
  getGroupIndexFromRootIndexMap() {
    return this._zeroMap;
  }

  getGroupIndexFromRootIndex() {
    return 0;
  }

  drillUp() {
    return this; // already as coarse as it gets
  }

  drillDown(attribute) {
    if (this.childDimension === null) throw new Error('Must set child dimension.');
    return this.childDimension.drillUp(attribute);
  }

  dice(attribute, items) {
    const keepsTheItem = attribute === this.rootAttribute && items.includes(ONLY_ITEM);
    return keepsTheItem ? this : unsupported();
  }

  union() {
    return this;
  }

  intersect(other) {
    return other;
  }
}

// what a placeholder cannot answer
Object.defineProperty(CatchAllDimension.prototype, 'attributes', { get: unsupported });
CatchAllDimension.prototype.diceRange = unsupported;
CatchAllDimension.prototype.serialize = unsupported;

module.exports = CatchAllDimension;
