'use strict';
/*
 * Loads the N-API addon (lib/olapgpu.node -> lib/libolapgpu.so).  There is exactly one backend;
 * if the addon is missing or no MI355X is visible the error surfaces — nothing falls back to JS.
 *
 * Multi-GPU: setDevices([0, 1, ..., 7]) (or the environment variable OLAP_DEVICES=0,1,...,7 read at
 * load) makes every stored measure created afterwards a ShardedStore, split along the cube's
 * outermost dimension over those devices (include/olap_hip.h "Multi-GPU"); setDevices(null) goes back
 * to one device.  One device repeated (e.g. [0, 0]) is allowed: the shards then exchange by direct
 * reads instead of RCCL, which lets a one-GPU machine run the sharded path.
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
    const env = process.env.OLAP_DEVICES;
    if (env) addon.setDevices(env.split(',').map((d) => parseInt(d, 10)));
  }
  return addon;
}

/** The devices new stored measures are sharded over; fewer than two entries (or null) = one device. */
function setDevices(devices) {
  return load().setDevices(devices || []);
}

/** Number of shards of new stored measures (0 = not sharded). */
function shardWorld() {
  return load().shardWorld();
}

/**
 * How int32 / uint32 measures are held on the device (store/hip.js cellTypeOf): false (default) = Float64
 * cells, the reference's values exactly; true = 4-byte typed cells, coerced after every operation.
 * Applies to measures created afterwards.
 */
let compact = process.env.OLAP_COMPACT_INT === '1';
function setCompactIntegers(on) {
  compact = !!on;
}
function compactIntegers() {
  return compact;
}

module.exports = { load, setDevices, shardWorld, setCompactIntegers, compactIntegers };
