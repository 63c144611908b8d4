'use strict';
// Minimal describe/it runner on top of Node's assert (vitest is not available offline).
const assert = require('assert');

const suites = [];
let current = null;

function describe(name, fn) {
  const parent = current;
  current = { name: (parent ? parent.name + ' > ' : '') + name, tests: [], before: parent ? parent.before.slice() : [] };
  suites.push(current);
  fn();
  current = parent;
}
function beforeEach(fn) {
  current.before.push(fn);
}
function it(name, fn) {
  current.tests.push({ name, fn, before: current.before.slice() });
}

// chai-style helpers used by the reference's tests, NaN-aware like chai's deepEqual
const chaiLike = {
  equal: (a, b, m) => assert.strictEqual(a, b, m),
  deepEqual: (a, b, m) => assert.deepStrictEqual(a, b, m),
  throws: (fn, re) => assert.throws(fn, re),
  doesNotThrow: (fn) => assert.doesNotThrow(fn),
  sameMembers: (a, b) => assert.deepStrictEqual(a.slice().sort(), b.slice().sort()),
  ok: (v, m) => assert.ok(v, m),
};

function run() {
  let passed = 0;
  const failures = [];
  for (const suite of suites) {
    for (const t of suite.tests) {
      try {
        for (const b of t.before) b();
        t.fn();
        ++passed;
      } catch (e) {
        failures.push(`${suite.name} > ${t.name}\n    ${String(e && e.stack ? e.stack : e).split('\n').slice(0, 6).join('\n    ')}`);
      }
    }
  }
  console.log(`${passed} passed, ${failures.length} failed`);
  for (const f of failures) console.log('FAIL ' + f);
  process.exit(failures.length ? 1 : 0);
}

module.exports = { describe, it, beforeEach, assert: chaiLike, run };
