// flat_baseline.js — TEST / BENCH INFRASTRUCTURE (bench.py's cpu_baseline_js leg only; never part of
// the product path).
//
// The flat-TypedArray JavaScript form of drillUp(sum) that the reference's README describes
// (README.md:12-14, 105-108) and SURVEY.md section 8(d)(1) asks to be timed under node on the GPU box:
// one Float32Array per measure, [K, inner] -> [1, inner] column sums with float64 accumulators in
// ascending row order (in-memory.js:282-290, 311-318), one thread.  Timing shape: the reference's own
// benchmark, test/cube-benchmark.js:5-18 (process.hrtime around repeated calls, mean reported).
//
//   node flat_baseline.js <K> <inner> <repetitions>   -> one JSON line
'use strict';
const os = require('os');
const K = parseInt(process.argv[2] || '10', 10);
const inner = parseInt(process.argv[3] || '10000000', 10);
const reps = parseInt(process.argv[4] || '5', 10);

// mulberry32, the generator of oracle/gen_golden.js
function mulberry32(a) {
  return function () {
    a |= 0; a = (a + 0x6d2b79f5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

const n = K * inner;
const data = new Float32Array(n);
const rnd = mulberry32(20240807);
for (let i = 0; i < n; ++i) data[i] = 0.5 + rnd();

function drillUpSum(values, K, inner) {
  const acc = new Float64Array(inner);
  for (let k = 0; k < K; ++k) {
    const base = k * inner;
    for (let i = 0; i < inner; ++i) acc[i] += values[base + i];
  }
  return Float32Array.from(acc);
}

let out = drillUpSum(data, K, inner); // warm-up (JIT)
const times = [];
for (let r = 0; r < reps; ++r) {
  const t0 = process.hrtime.bigint();
  out = drillUpSum(data, K, inner);
  times.push(Number(process.hrtime.bigint() - t0) / 1e9);
}
const mean = times.reduce((a, b) => a + b, 0) / times.length;
let check = 0;
for (let i = 0; i < Math.min(1000, inner); ++i) check += out[i];
console.log(JSON.stringify({
  cells: n, K, inner, repetitions: reps, seconds_mean: mean, seconds_min: Math.min.apply(null, times),
  cells_per_s: n / mean, cores_used: 1, host_cores: os.cpus().length, node: process.version, check,
}));
