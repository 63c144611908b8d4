/*
 * olap_oracle.h — CPU ORACLE (test infrastructure, not product code).
 *
 * A plain-C restatement of the reference's per-measure cell store,
 * /root/reference/src/store/in-memory.js (Growblocks/olap-in-memory @ 2024_08_07).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library; the product path (libolapgpu + hosts) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against vectors produced by executing the reference's own JavaScript
 * (oracle/gen_golden.js -> tests/golden/), including the reference tests' literals.
 *
 * The reference keeps cells in a JS Map<flatIndex, number> (in-memory.js:63): values are
 * float64, never coerced to the declared type, and iteration follows insertion order.
 * The oracle reproduces exactly that: double values, a presence flag per index and an
 * append-only insertion log with tombstones (the same structure V8 uses), so that
 * first/last and "delete on default" behave as in the reference.
 */
#ifndef OLAP_ORACLE_H
#define OLAP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_INT32 = 0, ORACLE_UINT32 = 1, ORACLE_FLOAT32 = 2, ORACLE_FLOAT64 = 3 };
enum {
  ORACLE_SUM = 0,
  ORACLE_AVERAGE = 1,
  ORACLE_HIGHEST = 2,
  ORACLE_LOWEST = 3,
  ORACLE_FIRST = 4,
  ORACLE_LAST = 5,
  ORACLE_PRODUCT = 6
};

typedef struct oracle_store oracle_store;

/* in-memory.js:48-64.  default_is_nan: 1 = NaN default, 0 = 0 default. */
oracle_store *oracle_store_new(uint64_t size, int type, int default_is_nan);
void oracle_store_free(oracle_store *s);
oracle_store *oracle_store_clone(const oracle_store *s); /* :66-73 */

uint64_t oracle_size(const oracle_store *s);
int oracle_type(const oracle_store *s);
int oracle_default_is_nan(const oracle_store *s);
uint64_t oracle_num_keys(const oracle_store *s);
/* keys / values in Map insertion order; arrays sized oracle_num_keys() */
void oracle_entries(const oracle_store *s, uint64_t *keys, double *values);
/* `data` getter (:30-37) plus the key-presence bitmap */
void oracle_dense(const oracle_store *s, double *values, uint8_t *present);
double oracle_total(const oracle_store *s); /* :22-28 */

double oracle_get_value(const oracle_store *s, uint64_t index);           /* :118-120 */
void oracle_set_value(oracle_store *s, uint64_t index, double value);     /* :122-133 */
void oracle_unset_value(oracle_store *s, uint64_t index);                 /* setValue(i, null|undefined) */
void oracle_set_data(oracle_store *s, const double *values);              /* `data` setter :39-46 */
void oracle_fill(oracle_store *s, double value);                          /* :135-137 */
/* SURVEY §8(d) generator: per cell i ascending, v = fround(0.5 + u1), keep iff u2 < frac
 * (two mulberry32 draws per cell), same stream as oracle/gen_golden.js configCube(). */
void oracle_fill_seeded(oracle_store *s, uint32_t seed, double frac);

/* maps / sel are concatenated per-dimension tables; dimension d starts at offset
 * sum(len[0..d-1]) of the corresponding length vector. */
/* :265-334 — maps[d][oldRootIdx] -> newIdx, lengths old_len[d].  Returns NULL for an unknown method. */
oracle_store *oracle_drillup(const oracle_store *s, int ndim, const uint32_t *old_len,
                             const uint32_t *new_len, const uint32_t *maps, int method);
/* :336-430 — maps[d][newRootIdx] -> oldIdx, lengths new_len[d].  distributions may be NULL
 * (n_dist entries; NaN entry = "missing").  Returns NULL and sets *missing_index when the
 * reference would throw `distribution missing for index i`. */
oracle_store *oracle_drilldown(const oracle_store *s, int ndim, const uint32_t *old_len,
                               const uint32_t *new_len, const uint32_t *maps, int method,
                               const double *distributions, uint64_t n_dist,
                               int64_t *missing_index);
/* :213-263 — sel[d][newIdx] -> oldIdx or -1 (item unknown to the old dimension) */
oracle_store *oracle_dice(const oracle_store *s, int ndim, const uint32_t *old_len,
                          const uint32_t *new_len, const int32_t *sel);
/* :178-211 — new axis i is old axis perm[i] */
oracle_store *oracle_reorder(const oracle_store *s, int ndim, const uint32_t *old_len,
                             const int32_t *perm);
/* :139-176 — his_to_mine[d][hisIdx] -> myIdx or -1; mutates `mine` */
void oracle_load(oracle_store *mine, const oracle_store *his, int ndim, const uint32_t *my_len,
                 const uint32_t *his_len, const int32_t *his_to_mine);

#ifdef __cplusplus
}
#endif
#endif
