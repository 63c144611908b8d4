'use strict';
/*
 * Entry point of the Node.js host: the reference's public names (src/index.js:1-6) for the
 * aggregation path.  `getParser` returns the formula parser of computed measures (./formula.js).
 */
const Cube = require('./cube');
const GenericDimension = require('./dimension/generic');
const TimeDimension = require('./dimension/time');
const HipStore = require('./store/hip');
const TimeSlot = require('./calendar');
const wire = require('./wire');

const { getParser } = require('./formula');

module.exports = { Cube, GenericDimension, TimeDimension, getParser, HipStore, TimeSlot, wire };
