'use strict';
/*
 * flat cell array <-> nested arrays / objects keyed by dimension items
 * (/root/reference/src/formatter/nested-array.js, nested-object.js).  Nesting is done for every
 * dimension; the reference's array form only nests correctly for <= 2 dimensions (its loop
 * re-slices the original array), which is what its tests use.
 */

function toNestedArray(values, dimensions) {
  if (dimensions.length === 0) return values[0];
  let level = values;
  for (let d = dimensions.length - 1; d > 0; --d) {
    const width = dimensions[d].numItems;
    const next = new Array(width ? level.length / width : 0);
    for (let j = 0; j < next.length; ++j) next[j] = level.slice(j * width, (j + 1) * width);
    level = next;
  }
  return level;
}

function fromNestedArray(values, dimensions) {
  let flat = values;
  for (let d = 1; d < dimensions.length; ++d) flat = [].concat(...flat);
  return flat;
}

function toNestedObject(values, dimensions, depth = 0, offset = 0) {
  if (depth >= dimensions.length) return values[offset];
  const items = dimensions[depth].getItems();
  const out = {};
  items.forEach((item, i) => {
    out[item] = toNestedObject(values, dimensions, depth + 1, offset * items.length + i);
  });
  return out;
}

function fromNestedObject(value, dimensions) {
  let level = [value];
  for (const dimension of dimensions) {
    const items = dimension.getItems();
    const next = new Array(level.length * items.length);
    for (let j = 0; j < next.length; ++j) next[j] = level[Math.floor(j / items.length)][items[j % items.length]];
    level = next;
  }
  return level;
}

module.exports = { toNestedArray, fromNestedArray, toNestedObject, fromNestedObject };
