'use strict';
/*
 * Common part of the dimension classes.  A store operation reads exactly five things from a
 * dimension (in-memory.js:141-147, :196, :216-223, :268-273, :347-352): numItems, getItems(),
 * getItemsToIdx(), rootAttribute and getGroupIndexFromRootIndexMap(attribute); the rest are the
 * item <-> index <-> group conversions of the public API (contract of
 * /root/reference/src/dimension/abstract.js).  Subclasses provide getItems(attribute),
 * attributes, getGroupIndexFromRootIndex(attribute, rootIndex) and the drill/dice operations.
 */
const mustOverride = (name) =>
  function () {
    throw new Error(`Override me (${name})`);
  };

class AbstractDimension {
  constructor(id, rootAttribute, label = null) {
    this.id = id;
    this._rootAttribute = rootAttribute;
    this._label = label;
    this._positions = new Map(); // attribute -> plain object { item: index }, built on demand
  }

  get rootAttribute() {
    return this._rootAttribute;
  }

  get label() {
    return this._label;
  }

  get numItems() {
    return this.getItems().length;
  }

  /** Lookup table item -> index for one attribute (the root attribute by default). */
  getItemsToIdx(attribute = null) {
    const key = attribute || this._rootAttribute;
    if (!this._positions.has(key)) {
      const table = {};
      const items = this.getItems(key);
      // a repeated item keeps its LAST position, as in the reference (src/dimension/abstract.js:66)
      for (let index = 0; index < items.length; ++index) table[items[index]] = index;
      this._positions.set(key, table);
    }
    return this._positions.get(key);
  }

  /** -1 when the item is not a root item */
  getRootIndexFromRootItem(rootItem) {
    const at = this.getItemsToIdx()[rootItem];
    return at === undefined ? -1 : at;
  }

  getGroupIndexFromRootItem(attribute, rootItem) {
    return this.getGroupIndexFromRootIndex(attribute, this.getRootIndexFromRootItem(rootItem));
  }

  getGroupItemFromRootIndex(attribute, rootIndex) {
    return this.getItems(attribute)[this.getGroupIndexFromRootIndex(attribute, rootIndex)];
  }

  getGroupItemFromRootItem(attribute, rootItem) {
    return this.getGroupItemFromRootIndex(attribute, this.getRootIndexFromRootItem(rootItem));
  }

  _forgetPositions(attribute) {
    this._positions.delete(attribute);
  }

  _checkRootIndex(index) {
    const n = this.numItems;
    if (!(index >= 0 && index < n)) throw new Error(`rootIndex ${index} out of bounds [0, ${n}[`);
  }

  _checkAttribute(attribute) {
    if (this.attributes.indexOf(attribute) === -1) throw new Error(`No attribute ${attribute} was found on dimension ${this.id}`);
  }
}

for (const name of ['getItems', 'drillUp', 'dice', 'diceRange', 'getGroupIndexFromRootIndex']) AbstractDimension.prototype[name] = mustOverride(name);
Object.defineProperty(AbstractDimension.prototype, 'attributes', { get: mustOverride('attributes'), configurable: true });

module.exports = AbstractDimension;
