'use strict';
/*
 * GenericDimension — explicit item list plus a graph of roll-up attributes.  Owns the
 * root-index -> group-index maps (Uint32Array) that are the drillUp/drillDown kernels' key input.
 * Behaviour follows /root/reference/src/dimension/generic.js (ctor :12-45, addAttribute :83-113,
 * drillUp :154-195, dice :197-241, getGroupIndexFromRootIndexMap :243-247, union :256-318,
 * intersect :320-337); the implementation is independent.
 */
const AbstractDimension = require('./abstract');
const { toBuffer, fromBuffer } = require('../wire');

const resolve = (source, key) => {
  if (!source) return key;
  return typeof source === 'function' ? source(key) : source[key];
};

class GenericDimension extends AbstractDimension {
  constructor(id, rootAttribute, items, label = null, itemToLabel = null) {
    super(id, rootAttribute, label);
    // attribute -> { items: string[], map: Uint32Array(root index -> item index), labels: {item: label} }
    this._attr = {};
    this._attr.all = { items: ['all'], map: new Uint32Array(items.length), labels: { all: 'All' } };
    const labels = {};
    items.forEach((item) => {
      labels[item] = resolve(itemToLabel, item);
    });
    this._attr[rootAttribute] = { items, map: Uint32Array.from(items, (_item, i) => i), labels };
  }

  get attributes() {
    return Object.keys(this._attr);
  }

  getItems(attribute = null) {
    const entry = this._attr[attribute || this._rootAttribute];
    return entry ? entry.items : undefined;
  }

  getEntries(attribute = null) {
    const entry = this._attr[attribute || this._rootAttribute];
    return entry.items.map((item) => [item, entry.labels[item]]);
  }

  /**
   * Adds a parent attribute derived from an existing one.  Groups are numbered in order of first
   * appearance along the root items (which is what makes drillUp outputs "ascending").
   */
  addAttribute(baseAttribute, newAttribute, baseToNew, newToLabel = null) {
    const base = this._attr[baseAttribute];
    const n = this.numItems;
    const items = [];
    const seen = {};
    const labels = {};
    const map = new Uint32Array(n);
    for (let root = 0; root < n; ++root) {
      const group = resolve(baseToNew, base.items[base.map[root]]);
      if (typeof group !== 'string') throw new Error('Mapping result must be a string.');
      if (seen[group] === undefined) {
        seen[group] = items.length;
        items.push(group);
        labels[group] = resolve(newToLabel, group);
      }
      map[root] = seen[group];
    }
    this._attr[newAttribute] = { items, map, labels };
    this._forgetPositions(newAttribute);
  }

  renameItem(oldItem, newItem, newLabel = null) {
    if (this.getItems().includes(newItem)) throw new Error(`Item ${newItem} already exists`);
    for (const attr of this.attributes) {
      const entry = this._attr[attr];
      const at = entry.items.indexOf(oldItem);
      if (at !== -1) entry.items[at] = newItem;
      if (entry.labels[oldItem]) {
        entry.labels[newItem] = newLabel || newItem;
        delete entry.labels[oldItem];
      }
      this._forgetPositions(attr);
    }
  }

  /** New dimension rooted at `attribute`, keeping every attribute that is still a function of it. */
  drillUp(attribute) {
    if (attribute === this._rootAttribute) return this;
    const target = this._attr[attribute];
    if (!target) throw new Error(`No attribute ${attribute} was found on dimension ${this.id}`);
    const result = new GenericDimension(this.id, attribute, target.items, this.label, target.labels);
    const n = this.numItems;
    for (const other of this.attributes) {
      if (other === attribute) continue;
      const candidate = this._attr[other];
      const parentOf = {};
      let functional = true;
      for (let root = 0; root < n && functional; ++root) {
        const from = target.items[target.map[root]];
        const to = candidate.items[candidate.map[root]];
        if (parentOf[from] && parentOf[from] !== to) functional = false; // no clean cut in the graph
        else parentOf[from] = to;
      }
      if (functional) result.addAttribute(attribute, other, parentOf, candidate.labels);
    }
    return result;
  }

  /** Keeps the root items selected directly (root attribute) or through a group attribute. */
  dice(attribute, items, reorder = false) {
    const current = this.getItems();
    let kept;
    if (attribute === this._rootAttribute) {
      kept = reorder ? items.filter((item) => current.includes(item)) : current.filter((item) => items.includes(item));
    } else {
      if (reorder) throw new Error('Reordering is not allowed when using groups');
      kept = current.filter((item) => items.includes(this.getGroupItemFromRootItem(attribute, item)));
    }
    if (kept.length === current.length && kept.every((item, i) => item === current[i])) return this;

    const root = this._attr[this._rootAttribute];
    const result = new GenericDimension(this.id, this._rootAttribute, kept, this.label, root.labels);
    for (const attr of this.attributes) {
      if (attr === this._rootAttribute) continue;
      result.addAttribute(this._rootAttribute, attr, (item) => this.getGroupItemFromRootItem(attr, item), this._attr[attr].labels);
    }
    return result;
  }

  getGroupIndexFromRootIndexMap(attribute) {
    this._checkAttribute(attribute);
    return this._attr[attribute].map;
  }

  getGroupIndexFromRootIndex(attribute, rootIndex) {
    this._checkAttribute(attribute);
    this._checkRootIndex(rootIndex);
    return this._attr[attribute].map[rootIndex];
  }

  union(other) {
    if (this.id !== other.id) throw new Error('not the same dimension');
    let mine = this;
    let theirs = other;
    if (this.attributes.includes(other.rootAttribute)) mine = this.drillUp(other.rootAttribute);
    else if (other.attributes.includes(this.rootAttribute)) theirs = other.drillUp(this.rootAttribute);
    else throw new Error('The dimensions are not compatible');

    const labelOf = (attr, item) => {
      for (const dim of [mine, theirs]) {
        const entry = dim._attr[attr];
        if (entry && entry.labels[item]) return entry.labels[item];
      }
      return item;
    };
    const groupOf = (attr, rootItem) => {
      try {
        return mine.getGroupItemFromRootItem(attr, rootItem);
      } catch (_e) {
        return theirs.getGroupItemFromRootItem(attr, rootItem);
      }
    };
    const own = mine.getItems();
    // NB: the extra items come from the dimension as it was passed in, as in the reference (:289-293)
    const merged = own.concat(other.getItems().filter((item) => !own.includes(item))).sort();
    const result = new GenericDimension(mine.id, mine.rootAttribute, merged, mine.label, (item) => labelOf(mine.rootAttribute, item));
    const groups = new Set();
    for (const dim of [mine, theirs]) for (const attr of dim.attributes) if (attr !== dim.rootAttribute) groups.add(attr);
    for (const attr of groups) {
      try {
        result.addAttribute(mine.rootAttribute, attr, (item) => groupOf(attr, item), (item) => labelOf(attr, item));
      } catch (_e) {
        // an attribute that cannot be rebuilt on the merged items is dropped
      }
    }
    return result;
  }

  intersect(other) {
    if (this.id !== other.id) throw new Error('not the same dimension');
    let attribute;
    if (this.attributes.includes(other.rootAttribute)) attribute = other.rootAttribute;
    else if (other.attributes.includes(this.rootAttribute)) attribute = this.rootAttribute;
    else throw new Error('The dimensions are not compatible');
    const theirs = other.getItems(attribute);
    const common = this.getItems(attribute).filter((item) => theirs.includes(item));
    return this.drillUp(attribute).dice(attribute, common);
  }

  /** Same record as the reference (generic.js:62-72). */
  serialize() {
    const items = {};
    const labels = {};
    const maps = {};
    for (const attr of this.attributes) {
      items[attr] = this._attr[attr].items;
      labels[attr] = this._attr[attr].labels;
      maps[attr] = this._attr[attr].map;
    }
    return toBuffer({
      id: this.id,
      label: this.label,
      rootAttribute: this._rootAttribute,
      rootItems: items[this._rootAttribute],
      attributeItems: items,
      attributeLabels: labels,
      attributeMappings: maps,
    });
  }

  static deserialize(buffer) {
    const data = fromBuffer(buffer);
    const dimension = new GenericDimension(data.id, data.rootAttribute, data.attributeItems[data.rootAttribute], data.label);
    for (const attr of Object.keys(data.attributeItems)) {
      dimension._attr[attr] = {
        items: data.attributeItems[attr],
        map: Uint32Array.from(data.attributeMappings[attr]),
        labels: (data.attributeLabels && data.attributeLabels[attr]) || {},
      };
    }
    return dimension;
  }
}

module.exports = GenericDimension;
