'use strict';
/*
 * Loads the N-API addon (lib/olapgpu.node -> lib/libolapgpu.so).  There is exactly one backend;
 * if the addon is missing or no MI355X is visible the error surfaces — nothing falls back to JS.
 */
const path = require('path');

let addon = null;

function load() {
  if (!addon) {
    const file = path.join(__dirname, '..', 'lib', 'olapgpu.node');
    try {
      addon = require(file);
    } catch (e) {
      throw new Error(`olap-in-memory_amd: cannot load the HIP addon (${file}): ${e.message}. Build it with __graft_entry__.build().`);
    }
  }
  return addon;
}

module.exports = { load };
