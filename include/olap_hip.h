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

#define OLAP_ABI_VERSION 2
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
  /* Not a reference method.  The shard-local half of `sum` / `average` for a cube partitioned across GPUs
   * (drillUp plans only).  out_values is a FLOAT64 buffer whatever the cell type: it receives the float64
   * accumulator itself — the undivided, unrounded sum of the set cells, 0 where nothing contributed (never
   * NaN) — because the reference adds all contributions of a cell in float64 and rounds never
   * (in-memory.js:282-290); out_status (optional) receives the NUMBER of contributions, not a mask.  Ranks add
   * both; the sums are rounded to the cell type once, after the division for `average`
   * (olap_shard_recipe: OLAP_FINISH_ROUND / OLAP_FINISH_AVERAGE). */
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
 * run any number of times on buffers of the planned sizes.  It belongs to the device that was
 * current when it was built (its tables and scratch live there; running it with another device
 * current fails with OLAP_ERR_INVALID_ARGUMENT), it is not thread-safe, and because the reduce
 * regimes keep their scratch in the plan it must not run on two streams at once.
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
 * fails with OLAP_ERR_DISTRIBUTION_MISSING if a contributing cell needs it, :397-398).
 * method: OLAP_SUM splits the parent between its children, anything else copies it (:421-423).
 * The reference decides "integers: spread the remainder one by one" by the measure's DECLARED type
 * (:343), while its cells are float64 numbers until serialize() (:77-92).  A host that wants exactly
 * that — int32 / uint32 measures whose intermediate values keep fractions and never wrap — holds them
 * in OLAP_FLOAT64 cells and ORs OLAP_DRILLDOWN_INTEGER_MEASURE into `method`; the remainder rule
 * (:403-417) then runs on the float64 cells.  Int32 / Uint32 cells always use it. */
#define OLAP_DRILLDOWN_INTEGER_MEASURE 0x100
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
/* The same plan over n independent (input, output) buffer pairs — the stored measures of a cube that share cell
 * type, default and rule: Cube.drillUp calls the store once per measure (src/cube.js:1012-1020), and on cubes of a
 * few MB a launch costs more than the bytes it moves.  drillUp plans of one dimension outside the few-outputs reduce
 * regime run up to 8 pairs per LAUNCH (the grid's second dimension picks the pair; pairs must all carry masks or
 * none); every other plan runs pair by pair behind this one call.  in_status / out_status may be NULL (no masks) or
 * lists whose entries may be NULL.  Same results as n olap_plan_run calls. */
int olap_plan_run_batch(olap_plan *plan, int n, const void *const *in_values, const int32_t *const *in_status,
                        void *const *out_values, int32_t *const *out_status, void *stream);
/* A drillUp plan over n pairs with a rule EACH (methods[i], one of the seven reference methods; the rule the plan
 * was built with is ignored — a drillUp plan's tables do not depend on it).  ONE launch when the roll-up runs in the
 * row regime with full 16-byte lanes and the pairs all carry masks or none (drillup_rows_mixed_kernel: the rule is a
 * workgroup-uniform switch); pair by pair otherwise.  Same results as n runs of n plans. */
int olap_plan_run_batch_rules(olap_plan *plan, int n, const int *methods, const void *const *in_values,
                              const int32_t *const *in_status, void *const *out_values, int32_t *const *out_status, void *stream);
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
/* Plain copies between host memory and device buffers of the raw-pointer API, for hosts without a
 * HIP binding of their own (blocking). */
int olap_memcpy_to_host(void *host, const void *device, uint64_t bytes);
int olap_memcpy_to_device(void *device, const void *host, uint64_t bytes);
/* Diagnostic: a plain 16-byte streaming read of `bytes` bytes that computes (and drops) one float sum per
 * lane (the achievable read ceiling of the box, measured in the same run as the kernels
 * it is compared with).  `scratch` needs 4 * 2048 bytes.  Asynchronous on `stream`. */
int olap_diag_read_ceiling(const void *device, uint64_t bytes, void *scratch, void *stream);
/* Diagnostic: the write-side counterpart — `bytes` bytes of `device` overwritten with 16-byte streaming stores (the box's
 * write ceiling; HBM is half-duplex, so a kernel that reads R and writes W bytes is bounded by R / read ceiling +
 * W / write ceiling, not by (R + W) / peak).  The buffer's contents are lost.  Asynchronous on `stream`. */
int olap_diag_write_ceiling(void *device, uint64_t bytes, void *stream);
/* Diagnostic, host-only (no device): where the cells of one row of the view [outer, K, inner] go in the LDS tile of the
 * row-tile drillUp regime when groups interleave (DESIGN.md K1', MODE 3) — cell_pos[K * inner] (LDS cell of every cell
 * of a row), group_bounds[2 G] (first and one-past-last member position of every group's run) and *pitch (members
 * between two rows of the tile).  `map` is a drillUp map as in olap_drillup_plan.  For tests of the planning code. */
int olap_diag_tile_placement(int dtype, uint32_t K, uint32_t G, uint32_t inner, const uint32_t *map, uint32_t *cell_pos,
                             uint32_t *group_bounds, uint32_t *pitch);
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

/* Insertion order of the reference's Map (in-memory.js:298 iterates it; `first` / `last`, keys() and serialize()
 * depend on it).  A dense buffer has none: by default cells are ordered by flat index, which is the reference's
 * order for a store filled ascending (`data=`, `fill`) and rolled up while dense.  A TRACKED store keeps the order
 * exactly — through setValue, data=, fill, drillUp (an output cell sits where its first contributing cell sat),
 * dice, reorder, drillDown and load — at the price of one more uint32 per cell once the order stops being the flat
 * index, and of a second pass per operation; results of operations on a tracked store are tracked.
 * olap_store_from_sparse turns tracking on by itself when its index list is not ascending.
 * sum / average / product depend on the order of their contributions (float64 addition is not associative, and a running
 * value that hits the default drops the key, which re-enters at the end): over an order that is not the flat index the
 * set cells are sorted by (output cell, insertion sequence) and every output cell is replayed in that order — the
 * reference's values and key order exactly, at the price of a radix sort; so are `first` / `last` over several
 * rolled-up dimensions at once.
 * Limits: stores below 2^31 cells; olap_store_totals refuses a tracked store ("ordered: ..."): run the chain of drillUps.
 * olap_store_order_tracked: 0 = not tracked, 1 = tracked and still ascending, 2 = tracked with an explicit order. */
int olap_store_track_order(olap_store *store, int on);
int olap_store_order_tracked(const olap_store *store);

/* `data` setter (:39-46) from a host typed array of the store's dtype (n must equal size,
 * else OLAP_ERR_LENGTH_MISMATCH) or from JS numbers */
int olap_store_set_data(olap_store *store, const void *host_values, uint64_t n);
int olap_store_set_data_f64(olap_store *store, const double *host_values, uint64_t n);
/* `data` getter (:30-37) into a host typed array / float64 array; status mask / key list */
int olap_store_get_data(const olap_store *store, void *host_values);
int olap_store_get_data_f64(const olap_store *store, double *host_values);
int olap_store_get_status(const olap_store *store, int32_t *host_status);
int olap_store_count_set(const olap_store *store, uint64_t *n_set);
/* indices of the set cells in the reference's _dataMap.keys() order: ascending, or the tracked insertion order
 * (olap_store_track_order); `cap` entries available, *n_keys receives the full count */
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

/* getNestedObject(measure, withTotals = true) (src/cube.js:421-440): every marginal of the measure in ONE
 * call.  The reference runs, for each of the 2^D subsets of dimensions, the chain drillUp(dim, 'all') over
 * the subset's dimensions in ascending order; all 2^D results are the cells of one extended cube of shape
 * (lens[0] + 1) x ... x (lens[D-1] + 1), row-major, index lens[d] of dimension d meaning 'all'.
 * methods[d] is the measure's rule for dimension d (src/cube.js:1013).  host_values receives that extended
 * cube as float64 (the default in unset cells), host_status (optional) its mask.  Same operations, order
 * and per-stage rounding as the chain of olap_store_drillup calls, hence the same values.
 * When the extended cube fits in LDS (<= 12288 cells) this is one launch that reads the cube from HBM
 * once; otherwise D + 2 launches instead of 2^D - 1.  *launches / *bytes_read (optional) report what ran. */
int olap_store_totals(const olap_store *store, int ndim, const uint32_t *lens, const int *methods, double *host_values,
                      int32_t *host_status, int *launches, uint64_t *bytes_read);

/* olap_eval_formula over stores (all of the same size) into a host float64 array */
int olap_store_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                            const olap_store *const *inputs, const double *scalars, int n_scalars,
                            double *host_out);

/* The five bulk operations; each returns a NEW store (load mutates `store`). */
int olap_store_drillup(const olap_store *store, olap_store **out, int ndim, const uint32_t *old_len,
                       const uint32_t *new_len, const uint32_t *const *maps, int method);
/* drillUp of n stores — the stored measures of one cube — by the same maps and rule: one plan and one launch when
 * the stores share cell type, default, size and device and none tracks its insertion order (olap_plan_run_batch);
 * otherwise the stores are rolled up one by one.  out[i] receives the new store of stores[i]; on an error none is
 * returned.  Replaces the per-measure loop of src/cube.js:1012-1020 for measures with the same rule. */
int olap_store_drillup_batch(int n, const olap_store *const *stores, olap_store **out, int ndim, const uint32_t *old_len,
                             const uint32_t *new_len, const uint32_t *const *maps, int method);
/* The same for stored measures with a rule EACH (methods[i], one of the seven reference methods): what Cube.drillUp does
 * for every stored measure of a cube, `store.drillUp(oldDims, newDims, storedMeasuresRules[id][dimensionId])`
 * (src/cube.js:1012-1020).  Measures that share cell type, default and size leave in ONE launch even when their rules
 * differ, when the roll-up runs in the row regime with full 16-byte lanes (the rule of a measure is a workgroup-uniform
 * switch: config 5's sum / average / first / last); otherwise one launch per rule (olap_store_drillup_batch), and
 * stores that fit no group one by one.  Same results as n olap_store_drillup calls. */
int olap_store_drillup_multi(int n, const olap_store *const *stores, const int *methods, olap_store **out, int ndim,
                             const uint32_t *old_len, const uint32_t *new_len, const uint32_t *const *maps);
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

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: a cube partitioned along its OUTERMOST dimension.  Row-major layout
 * (src/cube.js:709-728) makes a dimension-0 partition a set of contiguous slabs: rank r owns rows
 * [bounds[r], bounds[r+1]).  Every store operation that leaves dimension 0 alone runs per shard with
 * no communication.  drillUp ON dimension 0 (in-memory.js:265-334 with a non-identity map on the
 * sharded axis — the per-measure store call of src/cube.js:1012-1020) reduces each rank's own rows
 * into a partial [G, inner0] cube with the ordinary kernels and a row sub-map, then ONE RCCL
 * collective over xGMI combines the partials (olap_shard_recipe below says which).
 *
 * A communicator names the ranks and where they run:
 *   olap_comm_init_all     one process drives n devices (the Node.js host); ranks 0..n-1 are all local
 *   olap_comm_init_rank    one process per GPU (bench.py under torch.distributed.run); the 128-byte
 *                          unique id of rank 0 (olap_comm_unique_id) reaches the others out of band
 *   olap_comm_init_detached no transport: olap_shard_drillup_exchange() fails, the caller moves the
 *                          payloads itself (rehearsals over gloo; tests)
 * Ranks that share one device (init_all with a repeated device, e.g. {0,0}) exchange by reading
 * each other's buffers directly — RCCL refuses two ranks on one device — so a one-GPU machine can
 * run the sharded store end to end.
 * ---------------------------------------------------------------------------------------- */
typedef struct olap_comm olap_comm;
#define OLAP_UNIQUE_ID_BYTES 128
int olap_comm_unique_id(char id[OLAP_UNIQUE_ID_BYTES]);
int olap_comm_init_rank(olap_comm **comm, const char id[OLAP_UNIQUE_ID_BYTES], int world, int rank, int device);
int olap_comm_init_all(olap_comm **comm, const int *devices, int n);
int olap_comm_init_detached(olap_comm **comm, int world, int rank, int device);
void olap_comm_destroy(olap_comm *comm);
int olap_comm_world(const olap_comm *comm);
int olap_comm_local_count(const olap_comm *comm);            /* ranks driven by this process */
int olap_comm_local_rank(const olap_comm *comm, int local);  /* global rank of local rank `local` */
int olap_comm_local_device(const olap_comm *comm, int local);
/* "rccl" | "direct" (same-device ranks) | "detached" */
const char *olap_comm_transport(const olap_comm *comm);

/* Host-only partition arithmetic (no device needed).
 * olap_shard_bounds: contiguous balanced split of n_rows over `world` ranks, the first
 *   n_rows % world ranks get one more row; bounds has world + 1 entries.
 * olap_shard_dice_bounds: the partition left by a dice of dimension 0 with the strictly ascending
 *   row list `rows` (every rank keeps its own selected rows, no data moves): new_bounds[r] = number
 *   of selected rows below bounds[r].  Fails (OLAP_ERR_INVALID_ARGUMENT) when `rows` is not
 *   strictly ascending inside [0, bounds[world]). */
int olap_shard_bounds(uint32_t n_rows, int world, uint32_t *bounds);
int olap_shard_dice_bounds(const uint32_t *bounds, int world, const int32_t *rows, uint32_t n_rows_selected,
                           uint32_t *new_bounds);

/* What one sharded drillUp of dimension 0 ships and how the partials are combined. */
typedef enum { OLAP_XCHG_SUM = 0, OLAP_XCHG_MAX = 1, OLAP_XCHG_GATHER = 2 } olap_xchg_op;
typedef enum {
  OLAP_FINISH_NONE = 0,     /* sum over a 0 default: the reduced values ARE the result (set <=> != 0) */
  OLAP_FINISH_RESTORE = 1,  /* sum over a NaN default: cells no rank contributed to get the default back */
  OLAP_FINISH_AVERAGE = 2,  /* (sum, contribution count) -> in-memory.js:323-331, count modulo 65536 */
  OLAP_FINISH_COMBINE = 3,  /* highest/lowest/first/last/product: the same drillUp over the rank axis */
  OLAP_FINISH_ROUND = 4     /* sum of Float32 cells: the float64 partial sums, added in float64, are rounded to the cell
                             * type ONCE (as the one-device kernels round their float64 accumulator); a cell is set iff
                             * somebody contributed (NaN default: contribution counts travel too) and the rounded sum
                             * is not the default */
} olap_shard_finish;
typedef enum {
  OLAP_PLACE_SCATTER = 0,   /* additive methods: rank r keeps flat cells [r*per, (r+1)*per), per = ceil(n_out/world) */
  OLAP_PLACE_ALL = 1,       /* every rank holds the whole result */
  OLAP_PLACE_ROOT = 2,      /* rank 0 holds the whole result */
  OLAP_PLACE_SCATTER_ROWS = 3 /* like SCATTER in blocks of whole rows of the new leading dimension: rank r keeps rows
                               * [r*p, (r+1)*p) of it, p = ceil(new_len[0] / world) (what a sharded store needs) */
} olap_shard_placement;
typedef struct {
  int local_method;     /* olap_method run by each rank over its own rows */
  int zero_unset;       /* float64 cells over a NaN default, `sum`: unset partial cells are shipped as 0, never as NaN */
  int n_payloads;       /* 1 or 2 */
  int payload_dtype[2]; /* [0]: the cell type — OLAP_FLOAT64 for `average` and for `sum` of Float32 cells, whose partials
                         * are the float64 accumulators (twice the bytes on the wire); [1]: OLAP_INT32 (mask, or
                         * contribution counts) */
  int payload_op[2];    /* olap_xchg_op; masks are combined with MAX (an OR of 0 / 0x2), never added */
  int finish;           /* olap_shard_finish */
} olap_shard_recipe;
/* host-only; `method` is one of the seven reference methods */
int olap_shard_recipe_get(int dtype, int default_kind, int method, olap_shard_recipe *recipe);

/* Reusable sharded drillUp of dimension 0.  lens[ndim] are the GLOBAL old lengths, bounds[world+1]
 * the row partition, maps[d] as in olap_drillup_plan with maps[0] over the global rows (every rank
 * takes its own slice), new_len the global new lengths.  The object owns, per local rank, the
 * partial / exchange / result buffers (two sets when depth == 2) and one stream for the exchange.
 *
 * olap_shard_drillup_step(): for every local rank i — local reduction of in_values[i] on
 * streams[i], then the exchange and the finishing kernels on the exchange stream, ordered by
 * events only (no host wait).  With depth 2 consecutive steps are independent queries and the
 * exchange of step k overlaps the local reduction of step k+1.  olap_shard_drillup_wait() makes
 * streams[i] wait for everything issued so far.  The three phases can also be called one by one
 * (rehearsals that move the payloads themselves call _local, their own collective on the buffers
 * olap_shard_drillup_payload() names, then _finish).  Not thread-safe. */
typedef struct olap_shard_drillup olap_shard_drillup;
int olap_shard_drillup_create(olap_shard_drillup **op, olap_comm *comm, int dtype, int default_kind, int method,
                              int ndim, const uint32_t *lens, const uint32_t *new_len, const uint32_t *bounds,
                              const uint32_t *const *maps, int placement, int depth);
void olap_shard_drillup_destroy(olap_shard_drillup *op);
uint64_t olap_shard_drillup_out_cells(const olap_shard_drillup *op);     /* cells of the global result */
uint64_t olap_shard_drillup_local_cells(const olap_shard_drillup *op, int local);
const char *olap_shard_drillup_kernel_name(const olap_shard_drillup *op, int local);
int olap_shard_drillup_step(olap_shard_drillup *op, const void *const *in_values, const int32_t *const *in_status,
                            void *const *streams);
int olap_shard_drillup_wait(olap_shard_drillup *op, void *const *streams);
int olap_shard_drillup_local(olap_shard_drillup *op, int local, const void *in_values, const int32_t *in_status, void *stream);
int olap_shard_drillup_exchange(olap_shard_drillup *op, void *const *streams);
int olap_shard_drillup_finish(olap_shard_drillup *op, int local, void *stream);
/* payload p of local rank `local` in the buffer set of the LAST step: what is sent, where the combined
 * data must land (recv holds count cells for SUM / MAX under PLACE_ALL / ROOT, per cells under SCATTER,
 * world * count cells for GATHER), their element type and the combining operation */
int olap_shard_drillup_payload(const olap_shard_drillup *op, int local, int p, void **send, void **recv,
                               uint64_t *count, int *dtype, int *xchg_op);
/* result of the LAST step on local rank `local` (device pointers; *status may come back NULL when the
 * mask is a function of the values): flat range [*first, *first + *count) of the global output */
int olap_shard_drillup_result(const olap_shard_drillup *op, int local, void **values, int32_t **status,
                              uint64_t *first, uint64_t *count);

/* Sharded store handle: one measure whose dimension 0 is split over the ranks of `comm`; each local
 * rank's slab is an ordinary olap_store on its device.  In a one-process communicator the host-facing
 * accessors see the whole measure; with one process per GPU they see this process' rows only
 * (host arrays are still full-size: only the local slabs are read or written). */
typedef struct olap_sharded_store olap_sharded_store;
int olap_sharded_store_create(olap_sharded_store **store, olap_comm *comm, int ndim, const uint32_t *lens,
                              int dtype, int default_kind, const uint32_t *bounds /* NULL: balanced */);
void olap_sharded_store_destroy(olap_sharded_store *store);
uint64_t olap_sharded_store_size(const olap_sharded_store *store);
int olap_sharded_store_ndim(const olap_sharded_store *store);
const uint32_t *olap_sharded_store_lens(const olap_sharded_store *store);
const uint32_t *olap_sharded_store_bounds(const olap_sharded_store *store);  /* world + 1 entries */
/* Re-describes the dimensions without moving a cell (the reference's Cube inserts and drops one-item
 * dimensions around store calls, src/cube.js:919-927, :950-964): lens[0] must stay the sharded extent and
 * the product of the others the row size; otherwise OLAP_ERR_INVALID_ARGUMENT, message "sharded: ...". */
int olap_sharded_store_reshape(olap_sharded_store *store, int ndim, const uint32_t *lens);
olap_comm *olap_sharded_store_comm(const olap_sharded_store *store);
olap_store *olap_sharded_store_shard(const olap_sharded_store *store, int local);  /* borrowed */
int olap_sharded_store_fill_seeded(olap_sharded_store *store, uint32_t seed, double frac);
int olap_sharded_store_set_data_f64(olap_sharded_store *store, const double *host_values, uint64_t n);
int olap_sharded_store_get_data_f64(const olap_sharded_store *store, double *host_values);
int olap_sharded_store_get_status(const olap_sharded_store *store, int32_t *host_status);
int olap_sharded_store_get_value(const olap_sharded_store *store, uint64_t index, double *value, int *is_set);
int olap_sharded_store_set_value(olap_sharded_store *store, uint64_t index, double value, int is_null);
int olap_sharded_store_fill(olap_sharded_store *store, double value);
/* sum of the set cells of the WHOLE measure: with one process per GPU over RCCL every rank must call it (a
 * collective); on a detached communicator it is the sum of this process' slabs only */
int olap_sharded_store_total(const olap_sharded_store *store, double *total);
int olap_sharded_store_clone(const olap_sharded_store *store, olap_sharded_store **out);
/* olap_store_eval_formula over sharded inputs that are partitioned alike: evaluated per shard, every local slab of
 * host_out (full size) is filled */
int olap_sharded_store_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                                    const olap_sharded_store *const *inputs, const double *scalars, int n_scalars,
                                    double *host_out);
/* whole measure on the device of local rank 0 as an ordinary store, and back (one-process
 * communicators: device-to-device copies; one process per GPU: RCCL broadcasts of the slabs) */
int olap_sharded_store_gather(const olap_sharded_store *store, olap_store **out);
int olap_sharded_store_scatter(olap_sharded_store **store, olap_comm *comm, const olap_store *whole, int ndim,
                               const uint32_t *lens);
/* The bulk operations.  Dimension 0 untouched (identity map / selection / perm[0] == 0): per shard,
 * no communication, *out_sharded keeps the partition.  drillUp that changes dimension 0: partial +
 * ONE collective; when the new leading dimension still has a row per rank (day -> month on a sharded time
 * axis) the result stays sharded along it (reduce-scatter of whole rows, rank r keeps rows [r*p, (r+1)*p),
 * p = ceil(G0 / world)), otherwise ('all') the (K0 / G0 times smaller) result arrives as an ordinary store
 * in *out_whole on the device of local rank 0 (every process gets it when there is one process per GPU).  dice of
 * dimension 0 by a strictly ascending list of existing rows: per shard, the partition becomes
 * uneven.  Anything else that touches dimension 0 (reordering it, refining it, selections that
 * repeat, permute or invent rows): OLAP_ERR_INVALID_ARGUMENT with a message starting "sharded:" —
 * gather first.  Exactly one of *out_sharded / *out_whole is set on success. */
int olap_sharded_store_drillup(const olap_sharded_store *store, olap_sharded_store **out_sharded, olap_store **out_whole,
                               const uint32_t *new_len, const uint32_t *const *maps, int method);
int olap_sharded_store_dice(const olap_sharded_store *store, olap_sharded_store **out, const uint32_t *new_len,
                            const int32_t *const *sel);
int olap_sharded_store_drilldown(const olap_sharded_store *store, olap_sharded_store **out, const uint32_t *new_len,
                                 const uint32_t *const *maps, int method, const double *distributions, uint64_t n_dist);
int olap_sharded_store_reorder(const olap_sharded_store *store, olap_sharded_store **out, const int32_t *perm);

#ifdef __cplusplus
}
#endif
#endif /* OLAP_HIP_H */
