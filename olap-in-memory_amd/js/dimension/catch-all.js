'use strict';
/*
 * One-item placeholder ('_total') that Cube.addDimension puts where the new dimension will go, so
 * that adding a dimension is a drillDown from 1 item to the new items
 * (/root/reference/src/cube.js:919-927, src/dimension/catch-all.js).
 */
const AbstractDimension = require('./abstract');

class CatchAllDimension extends AbstractDimension {
  constructor(id, childDimension = null) {
    super(id, 'all');
    this.childDimension = childDimension;
  }

  get attributes() {
    throw new Error('Unsupported');
  }

  getItems(_attribute = null) {
    return ['_total'];
  }

  getEntries(_attribute = null, _language = 'en') {
    return [['_total', 'Total']];
  }

  drillUp(_attribute) {
    return this;
  }

  drillDown(attribute) {
    if (!this.childDimension) throw new Error('Must set child dimension.');
    return this.childDimension.drillUp(attribute);
  }

  dice(attribute, items, _reorder = false) {
    if (attribute === this.rootAttribute && items.includes('_total')) return this;
    throw new Error('Unsupported');
  }

  diceRange() {
    throw new Error('Unsupported');
  }

  getGroupIndexFromRootIndex() {
    return 0;
  }

  getGroupIndexFromRootIndexMap() {
    return new Uint32Array(1);
  }

  intersect(other) {
    return other;
  }

  union() {
    return this;
  }

  serialize() {
    throw new Error('Unsupported');
  }
}

module.exports = CatchAllDimension;
