'use strict';
// Prints, for every day of [from, to], the slot it rolls up to in each periodicity (one JSON object), for
// tests/test_calendar_independent.py to compare with Python's datetime.
const TimeSlot = require('../../olap-in-memory_amd/js/calendar');

const [from, to] = process.argv.slice(2);
const out = {};
const periodicities = ['week_sat', 'week_sun', 'week_mon', 'month_week_sat', 'month_week_sun', 'month_week_mon', 'month', 'quarter', 'semester', 'year'];
for (let ms = Date.parse(from + 'T00:00:00Z'); ms <= Date.parse(to + 'T00:00:00Z'); ms += 86400000) {
  const day = TimeSlot.fromDate(new Date(ms), 'day');
  const row = {};
  for (const p of periodicities) row[p] = day.toParentPeriodicity(p).value;
  out[day.value] = row;
}
console.log(JSON.stringify(out));
