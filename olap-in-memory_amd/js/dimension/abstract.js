'use strict';
/*
 * Base of the dimension objects the store consumes.  The store needs exactly five things from a
 * dimension (in-memory.js:141-147, :196, :216-223, :268-273, :347-352): numItems, getItems(),
 * getItemsToIdx(), rootAttribute and getGroupIndexFromRootIndexMap(attr).
 * Contract mirrored from /root/reference/src/dimension/abstract.js.
 */
class AbstractDimension {
  constructor(id, rootAttribute, label = null) {
    this.id = id;
    this._rootAttribute = rootAttribute;
    this._label = label;
    this._indexOfItem = {}; // attribute -> { item: index }
  }

  get numItems() {
    return this.getItems().length;
  }

  get rootAttribute() {
    return this._rootAttribute;
  }

  get label() {
    return this._label;
  }

  get attributes() {
    throw new Error('Override me');
  }

  getItems(_attribute = null) {
    throw new Error('Override me');
  }

  drillUp(_attribute) {
    throw new Error('Override me');
  }

  dice(_attribute, _items, _reorder = false) {
    throw new Error('Override me');
  }

  diceRange(_attribute, _start, _end) {
    throw new Error('Override me');
  }

  getGroupIndexFromRootIndex(_attribute, _rootIndex) {
    throw new Error('Override me');
  }

  /** { item -> index } of one attribute's items, cached. */
  getItemsToIdx(attribute = null) {
    const attr = attribute || this._rootAttribute;
    let table = this._indexOfItem[attr];
    if (!table) {
      table = {};
      this.getItems(attr).forEach((item, i) => {
        table[item] = i;
      });
      this._indexOfItem[attr] = table;
    }
    return table;
  }

  getRootIndexFromRootItem(rootItem) {
    const index = this.getItemsToIdx()[rootItem];
    return index === undefined ? -1 : index;
  }

  getGroupIndexFromRootItem(attribute, rootItem) {
    return this.getGroupIndexFromRootIndex(attribute, this.getRootIndexFromRootItem(rootItem));
  }

  getGroupItemFromRootIndex(attribute, rootIndex) {
    return this.getItems(attribute)[this.getGroupIndexFromRootIndex(attribute, rootIndex)];
  }

  getGroupItemFromRootItem(attribute, rootItem) {
    return this.getItems(attribute)[this.getGroupIndexFromRootItem(attribute, rootItem)];
  }

  _checkRootIndex(index) {
    if (index < 0 || index >= this.numItems) throw new Error(`rootIndex ${index} out of bounds [0, ${this.numItems}[`);
  }

  _checkAttribute(attribute) {
    if (!this.attributes.includes(attribute)) throw new Error(`No attribute ${attribute} was found on dimension ${this.id}`);
  }
}

module.exports = AbstractDimension;
