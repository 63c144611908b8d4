/* flat_baseline.c — TEST / BENCH INFRASTRUCTURE (bench.py's cpu_baseline legs only; never part of the
 * product path).
 *
 * The "flat TypedArray" CPU form of drillUp(sum) that the reference's README describes
 * (README.md:12-14,105-108: one contiguous Float32Array per measure) and SURVEY.md section 8(d)(2) asks
 * to be timed beside the GPU: a dense [K, inner] -> [1, inner] column sum with float64 accumulators
 * in ascending row order (the accumulation order and width of in-memory.js:282-290,311-318), plain
 * loops, no Map.  Built twice from this file: single thread and OpenMP over column blocks.
 *
 * flat_drillup_sum(in, out, K, inner, threads): returns the seconds of ONE pass.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int flat_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

double flat_drillup_sum(const float *in, float *out, uint64_t K, uint64_t inner, int threads) {
  const uint64_t block = 4096; /* columns per task: 16 KiB of accumulators stay in L1 */
  const uint64_t n_blocks = (inner + block - 1) / block;
  const double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
  for (uint64_t b = 0; b < n_blocks; ++b) {
    double acc[4096];
    const uint64_t lo = b * block, hi = lo + block < inner ? lo + block : inner, n = hi - lo;
    for (uint64_t i = 0; i < n; ++i) acc[i] = 0.0;
    for (uint64_t k = 0; k < K; ++k) {
      const float *row = in + k * inner + lo;
      for (uint64_t i = 0; i < n; ++i) acc[i] += (double)row[i];
    }
    for (uint64_t i = 0; i < n; ++i) out[lo + i] = (float)acc[i];
  }
  (void)threads;
  return now_s() - t0;
}
