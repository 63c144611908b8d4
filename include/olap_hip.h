/*
 * olap_hip.h — C ABI of libolapgpu, the MI355X (gfx950) implementation of the
 * olap-in-memory cube aggregation path.
 *
 * The reference (Growblocks/olap-in-memory @ 2024_08_07) has no FFI: its seam for this path
 * is the `InMemoryStore` class, src/store/in-memory.js, which `Cube` constructs at
 * src/cube.js:172-176 / :198-202 and calls at :1013 (drillUp), :938 / :979 (drillDown),
 * :628 / :826 / :851 (dice), :778 (reorder), :741 (load).  Every entry point below cites the
 * reference method it replaces.  The Node.js host (olap-in-memory_amd/js) binds these through
 * the N-API addon (olap-in-memory_amd/napi); Python hosts (bench.py, tests) bind them with
 * ctypes.  See INTEGRATION.md for the reference-side binding.
 *
 * Data model (dense restatement of the reference's Map<flatIndex, number>):
 *   values : one element per cell, row-major over the cube's dimensions, last dimension
 *            fastest (src/cube.js:709-728), element type = the measure's declared type;
 *   status : Int32 per cell, bit OLAP_STATUS_SET (0x2) <=> the reference Map holds the key.
 *   A cell is "set" iff (status == NULL || status[i] & 0x2) and values[i] is not the
 *   measure's default (setValue deletes default-valued keys, in-memory.js:122-133; for an
 *   integer type with NaN default no value is the default, so status alone decides).
 *   Outputs always hold the default in unset cells (0 for integer types) and an exact mask.
 *
 * All pointers named `*_values`, `*_status`, `workspace` are DEVICE pointers; everything
 * else is host memory.  Functions return OLAP_OK or a negative olap_error; the message of
 * the last failure on the calling thread is olap_last_error().  Arguments are validated
 * before the device is touched, so argument errors are identical with and without a GPU.
 * Nothing here falls back to the CPU: without a usable device the calls fail with
 * OLAP_ERR_NO_DEVICE.
 */
#ifndef OLAP_HIP_H
#define OLAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLAP_ABI_VERSION 1
#define OLAP_MAX_DIMS 32
#define OLAP_STATUS_SET 0x2 /* README.md:714-716: the only status bit the reference code implements */

/* in-memory.js:59 — ['int32','uint32','float32','float64'] */
typedef enum { OLAP_INT32 = 0, OLAP_UINT32 = 1, OLAP_FLOAT32 = 2, OLAP_FLOAT64 = 3 } olap_dtype;
/* in-memory.js:56-57 — only NaN and 0 are legal defaults */
typedef enum { OLAP_DEFAULT_ZERO = 0, OLAP_DEFAULT_NAN = 1 } olap_default;
/* in-memory.js:282-290 */
typedef enum {
  OLAP_SUM = 0,
  OLAP_AVERAGE = 1,
  OLAP_HIGHEST = 2,
  OLAP_LOWEST = 3,
  OLAP_FIRST = 4,
  OLAP_LAST = 5,
  OLAP_PRODUCT = 6,
  /* Not a reference method.  The shard-local half of `average` for a cube partitioned across
   * GPUs: out_values receives the (undivided) float64 sum of the set cells, out_status receives
   * the NUMBER of contributions (not a mask).  Ranks add both and call olap_average_finish(). */
  OLAP_PARTIAL_AVERAGE = 7
} olap_method;

typedef enum {
  OLAP_OK = 0,
  OLAP_ERR_INVALID_ARGUMENT = -1,
  OLAP_ERR_INVALID_TYPE = -2,        /* 'Invalid type'                                 in-memory.js:60 */
  OLAP_ERR_INVALID_DEFAULT = -3,     /* 'Invalid default value, only NaN and 0 ...'    in-memory.js:57 */
  OLAP_ERR_UNSUPPORTED_METHOD = -4,  /* 'Unsupported aggregation method: <m>'          in-memory.js:295 */
  OLAP_ERR_LENGTH_MISMATCH = -5,     /* 'value length is invalid: a !== b'             in-memory.js:41-43 */
  OLAP_ERR_DISTRIBUTION_MISSING = -6,/* 'distribution missing for index <i>'           in-memory.js:398 */
  OLAP_ERR_NO_DEVICE = -7,
  OLAP_ERR_HIP = -8,
  OLAP_ERR_OUT_OF_MEMORY = -9,
  OLAP_ERR_INDEX_RANGE = -10
} olap_error;

const char *olap_last_error(void);
int olap_abi_version(void);
/* 'sum' | 'average' | 'highest' | 'lowest' | 'first' | 'last' | 'product' -> olap_method, or
 * OLAP_ERR_UNSUPPORTED_METHOD (message = the reference's).  NULL means the default, 'sum'. */
int olap_method_from_name(const char *name);
/* 'int32' | 'uint32' | 'float32' | 'float64' -> olap_dtype, or OLAP_ERR_INVALID_TYPE */
int olap_dtype_from_name(const char *name);
size_t olap_dtype_size(int dtype);

/* Number of HIP devices visible (0 when none / no driver); selects the device used by the
 * calling thread's subsequent calls (hipSetDevice). */
int olap_device_count(void);
int olap_set_device(int device);
int olap_device_synchronize(void);

/* ------------------------------------------------------------------------------------------
 * Plans: an op's small host-side tables (per-dimension index maps) are validated, folded into
 * a launch plan and uploaded once; olap_plan_run() is then pure kernel launches on `stream`
 * (a hipStream_t, NULL = the null stream) and may be captured into a hipGraph.  A plan may be
 * run any number of times on buffers of the planned sizes; it is not thread-safe.
 * ---------------------------------------------------------------------------------------- */
typedef struct olap_plan olap_plan;

/* InMemoryStore.drillUp(oldDimensions, newDimensions, method) — in-memory.js:265-334.
 * maps[d] has old_len[d] entries: old root index -> new index (< new_len[d]); this is
 * oldDimensions[d].getGroupIndexFromRootIndexMap(newDimensions[d].rootAttribute) (:270-274).
 * Semantics kept from the reference: only set cells contribute; float64 accumulation in
 * ascending flat-index order; `average` divides by the number of contributions modulo 65536
 * (Uint16Array, :278/:320) and leaves the sum when that is 0; a cell whose running value equals
 * the default is dropped and restarts (:126-131); Math.max/Math.min NaN propagation. */
int olap_drillup_plan(olap_plan **plan, int dtype, int default_kind, int method, int ndim,
                      const uint32_t *old_len, const uint32_t *new_len,
                      const uint32_t *const *maps);

/* InMemoryStore.drillDown(oldDimensions, newDimensions, method, distributions) — :336-430.
 * maps[d] has new_len[d] entries: new root index -> old index (< old_len[d]) (:349-353).
 * distributions: NULL or n_dist float64 weights (NaN entry = JS null/undefined => the run
 * fails with OLAP_ERR_DISTRIBUTION_MISSING if a contributing cell needs it, :397-398). */
int olap_drilldown_plan(olap_plan **plan, int dtype, int default_kind, int method, int ndim,
                        const uint32_t *old_len, const uint32_t *new_len,
                        const uint32_t *const *maps, const double *distributions,
                        uint64_t n_dist);

/* InMemoryStore.dice(oldDimensions, newDimensions) — :213-263.
 * sel[d] has new_len[d] entries: new index -> old index, or -1 when the new item does not exist
 * in the old dimension (:219-224; such rows stay unset). */
int olap_dice_plan(olap_plan **plan, int dtype, int default_kind, int ndim,
                   const uint32_t *old_len, const uint32_t *new_len, const int32_t *const *sel);

/* Fused dice -> drillUp (the slice / dice / drillUp chains of src/cube.js:799-857 + :995-1023):
 * equivalent to olap_dice_plan(old_len -> mid_len, sel) followed by olap_drillup_plan(mid_len ->
 * new_len, maps, method) with ONE rolled-up dimension, but the selection tables are folded into the
 * reduction's address computation, so only the surviving cells are read, once, and the diced
 * intermediate cube is never written.  More than one non-identity map: OLAP_ERR_INVALID_ARGUMENT
 * (run the two plans instead). */
int olap_dice_drillup_plan(olap_plan **plan, int dtype, int default_kind, int method, int ndim,
                           const uint32_t *old_len, const uint32_t *mid_len, const uint32_t *new_len,
                           const int32_t *const *sel, const uint32_t *const *maps);

/* InMemoryStore.reorder(oldDimensions, newDimensions) — :178-211.  New axis i is old axis perm[i]. */
int olap_reorder_plan(olap_plan **plan, int dtype, int default_kind, int ndim,
                      const uint32_t *old_len, const int32_t *perm);

/* InMemoryStore.load(otherStore, myDimensions, hisDimensions) — :139-176.  his_to_mine[d] has
 * his_len[d] entries: his index -> my index (or -1: skipped).  `in` of olap_plan_run is the
 * other store (his default kind given here), `out` is this store and is updated in place. */
int olap_load_plan(olap_plan **plan, int dtype, int my_default_kind, int his_default_kind,
                   int ndim, const uint32_t *my_len, const uint32_t *his_len,
                   const int32_t *const *his_to_mine);

uint64_t olap_plan_in_cells(const olap_plan *plan);
uint64_t olap_plan_out_cells(const olap_plan *plan);
/* name of the kernel variant the plan selected (diagnostics / profiles) */
const char *olap_plan_kernel_name(const olap_plan *plan);

/* in_status may be NULL (every cell whose value is not the default is set).  out_status may be
 * NULL when the caller does not need the mask.  Buffers must not alias.  Asynchronous: returns
 * once the work is enqueued on `stream`. */
int olap_plan_run(olap_plan *plan, const void *in_values, const int32_t *in_status,
                  void *out_values, int32_t *out_status, void *stream);
/* After the stream has been synchronised: OLAP_OK, or the deferred data-dependent error of the
 * last run (OLAP_ERR_DISTRIBUTION_MISSING, message as in-memory.js:398). */
int olap_plan_status(olap_plan *plan);
void olap_plan_destroy(olap_plan *plan);

/* ------------------------------------------------------------------------------------------
 * Element-wise helpers on raw device buffers.
 * ---------------------------------------------------------------------------------------- */
/* Applies setValue (in-memory.js:122-133) to every cell: status[i] = set ? 0x2 : 0 and unset
 * cells get the canonical default.  If `status_in_place` already holds a mask it is AND-ed in
 * when `use_existing_status` != 0.  This is the `data` setter (:39-46) for a typed array. */
int olap_canonicalize(void *values, int32_t *status, uint64_t n, int dtype, int default_kind,
                      int use_existing_status, void *stream);
/* JS numbers (float64) -> typed cells with TypedArray conversion, then setValue semantics;
 * NaN in `nulls` positions is not needed: a JS null/undefined is passed as the default. */
int olap_convert_from_f64(const double *src_f64, void *values, int32_t *status, uint64_t n,
                          int dtype, int default_kind, void *stream);
int olap_convert_to_f64(const void *values, double *dst_f64, uint64_t n, int dtype, void *stream);
/* SURVEY §8(d) synthetic measure generated on the device: cell i gets fround(0.5 + u(2i)),
 * kept iff u(2i+1) < frac, u = mulberry32 stream of `seed` (identical to the golden
 * generator); `first_cell` lets a shard generate its own slab. */
int olap_fill_seeded(void *values, int32_t *status, uint64_t n, uint64_t first_cell, int dtype,
                     uint32_t seed, double frac, void *stream);
/* Completes a sharded `average` (in-memory.js:323-331): values[i] = counts[i] mod 65536 ?
 * values[i] / (counts[i] mod 65536) : values[i]; out_status (optional) gets the mask. */
int olap_average_finish(void *values, const int32_t *counts, int32_t *out_status, uint64_t n,
                        int dtype, int default_kind, void *stream);
/* Computed measures (src/cube.js:326-363: one formula evaluated per cell over the stored measures'
 * getValue(i)), as an element-wise interpreter on the device.  `code` is a postfix program
 * (opcodes: olap-in-memory_amd/js/formula.js OP / csrc FormulaOp; CONST k, INPUT i and SCALAR j
 * carry one operand word), at most OLAP_FORMULA_MAX_CODE words, OLAP_FORMULA_MAX_CONSTS constants,
 * OLAP_FORMULA_MAX_INPUTS input measures (device buffers of n cells each; status may be NULL per
 * input) and as many scalars (`<measure>__total` parameters).  out_f64: n float64 results. */
#define OLAP_FORMULA_MAX_CODE 96
#define OLAP_FORMULA_MAX_CONSTS 24
#define OLAP_FORMULA_MAX_INPUTS 8
#define OLAP_FORMULA_MAX_STACK 16
int olap_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                      const void *const *in_values, const int32_t *const *in_status, const int *in_dtypes,
                      const int *in_defaults, const double *scalars, int n_scalars, double *out_f64,
                      uint64_t n, void *stream);
/* `total` getter (in-memory.js:22-28): float64 sum of the set cells, and their count.
 * Synchronises `stream`. */
int olap_total(const void *values, const int32_t *status, uint64_t n, int dtype, int default_kind,
               double *total, uint64_t *n_set, void *stream);

/* ------------------------------------------------------------------------------------------
 * Store handles: device-resident cells owned by the library, for hosts that cannot hold device
 * pointers (the Node.js addon).  One handle = one measure's InMemoryStore (in-memory.js:7-64).
 * Bulk operations are enqueued on the library's (null) stream and return at once; they are ordered
 * behind each other, and every call that hands data to the host (get_*, total, to_sparse, ...) is
 * a blocking copy on that stream, so the API behaves synchronously like the reference's.  Plans
 * built for handles are cached (LRU) and device buffers come from a pool.
 * ---------------------------------------------------------------------------------------- */
typedef struct olap_store olap_store;

/* new InMemoryStore(size, type, defaultValue) — :48-64; every cell unset */
int olap_store_create(olap_store **store, uint64_t size, int dtype, int default_kind);
void olap_store_destroy(olap_store *store);
int olap_store_clone(const olap_store *store, olap_store **out); /* :66-73 */
uint64_t olap_store_size(const olap_store *store);
int olap_store_dtype(const olap_store *store);
int olap_store_default(const olap_store *store);
uint64_t olap_store_byte_length(const olap_store *store); /* :8-16 */
void *olap_store_values_ptr(const olap_store *store);       /* device pointer */
int32_t *olap_store_status_ptr(const olap_store *store);    /* device pointer */

/* `data` setter (:39-46) from a host typed array of the store's dtype (n must equal size,
 * else OLAP_ERR_LENGTH_MISMATCH) or from JS numbers */
int olap_store_set_data(olap_store *store, const void *host_values, uint64_t n);
int olap_store_set_data_f64(olap_store *store, const double *host_values, uint64_t n);
/* `data` getter (:30-37) into a host typed array / float64 array; status mask / key list */
int olap_store_get_data(const olap_store *store, void *host_values);
int olap_store_get_data_f64(const olap_store *store, double *host_values);
int olap_store_get_status(const olap_store *store, int32_t *host_status);
int olap_store_count_set(const olap_store *store, uint64_t *n_set);
/* ascending indices of the set cells (the reference's _dataMap.keys() for stores filled in
 * ascending order); `cap` entries available, *n_keys receives the full count */
int olap_store_get_keys(const olap_store *store, uint64_t *host_keys, uint64_t cap, uint64_t *n_keys);
int olap_store_get_value(const olap_store *store, uint64_t index, double *value, int *is_set); /* :118-120 */
int olap_store_set_value(olap_store *store, uint64_t index, double value, int is_null);       /* :122-133 */
int olap_store_fill(olap_store *store, double value);                                         /* :135-137 */
int olap_store_total(const olap_store *store, double *total);                                 /* :22-28 */

/* The sparse form the reference serialises (in-memory.js:94-100: `indexes` = Map keys as Uint32,
 * `dataBuffer` = the values as the store's TypedArray): device-side stream compaction of the set
 * cells in ascending index order, and its inverse.  olap_store_sparse_count() sizes the host
 * arrays; indices are 32-bit like the reference's (stores above 2^32 cells are refused). */
int olap_store_to_sparse(const olap_store *store, uint32_t *host_indexes, void *host_values,
                         uint64_t cap, uint64_t *n_set);
/* new store with the given cells set (deserialize, in-memory.js:103-116).  Values are stored as
 * given (TypedArray of the store's dtype); setValue semantics apply (a default value unsets). */
int olap_store_from_sparse(olap_store **store, uint64_t size, int dtype, int default_kind,
                           const uint32_t *host_indexes, const void *host_values, uint64_t n);

/* olap_eval_formula over stores (all of the same size) into a host float64 array */
int olap_store_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                            const olap_store *const *inputs, const double *scalars, int n_scalars,
                            double *host_out);

/* The five bulk operations; each returns a NEW store (load mutates `store`). */
int olap_store_drillup(const olap_store *store, olap_store **out, int ndim, const uint32_t *old_len,
                       const uint32_t *new_len, const uint32_t *const *maps, int method);
int olap_store_drilldown(const olap_store *store, olap_store **out, int ndim,
                         const uint32_t *old_len, const uint32_t *new_len,
                         const uint32_t *const *maps, int method, const double *distributions,
                         uint64_t n_dist);
int olap_store_dice(const olap_store *store, olap_store **out, int ndim, const uint32_t *old_len,
                    const uint32_t *new_len, const int32_t *const *sel);
int olap_store_dice_drillup(const olap_store *store, olap_store **out, int ndim, const uint32_t *old_len,
                            const uint32_t *mid_len, const uint32_t *new_len, const int32_t *const *sel,
                            const uint32_t *const *maps, int method);
int olap_store_reorder(const olap_store *store, olap_store **out, int ndim,
                       const uint32_t *old_len, const int32_t *perm);
int olap_store_load(olap_store *store, const olap_store *other, int ndim, const uint32_t *my_len,
                    const uint32_t *his_len, const int32_t *const *his_to_mine);

#ifdef __cplusplus
}
#endif
#endif /* OLAP_HIP_H */
