'use strict';
/*
 * Package entry of the Node.js host.  Carries the four names of the reference's entry point
 * (Cube, GenericDimension, TimeDimension, getParser — /root/reference/src/index.js) plus what is
 * specific to this implementation: the device store class, the calendar and the wire codec.
 */
const lazy = (file, pick) => {
  let loaded;
  return () => {
    if (loaded === undefined) loaded = pick ? require(file)[pick] : require(file);
    return loaded;
  };
};

const entries = {
  Cube: lazy('./cube'),
  GenericDimension: lazy('./dimension/generic'),
  TimeDimension: lazy('./dimension/time'),
  getParser: lazy('./formula', 'getParser'),
  HipStore: lazy('./store/hip'),
  TimeSlot: lazy('./calendar'),
  wire: lazy('./wire'),
  setDevices: lazy('./backend', 'setDevices'),
  shardWorld: lazy('./backend', 'shardWorld'),
  setCompactIntegers: lazy('./backend', 'setCompactIntegers'),
  backend: lazy('./backend'),
};

for (const name of Object.keys(entries)) Object.defineProperty(exports, name, { enumerable: true, get: entries[name] });
