'use strict';
/*
 * Entry point of the Node.js host: the reference's public names (src/index.js:1-6) for the
 * aggregation path.  `getParser` belongs to computed measures (expr-eval), which are outside the
 * accelerated path, so it is exported as a function that says so.
 */
const Cube = require('./cube');
const GenericDimension = require('./dimension/generic');
const TimeDimension = require('./dimension/time');
const HipStore = require('./store/hip');
const TimeSlot = require('./calendar');
const wire = require('./wire');

function getParser() {
  throw new Error('getParser (expr-eval computed measures) is outside the accelerated aggregation path of olap-in-memory_amd');
}

module.exports = { Cube, GenericDimension, TimeDimension, getParser, HipStore, TimeSlot, wire };
