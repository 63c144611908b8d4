// olap_internal.hpp — what the translation units of libolapgpu share besides the C ABI: error
// reporting, the device memory pool and the layout of a store handle.  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/olap_hip.h"

#define OLAP_INTERNAL __attribute__((visibility("hidden")))

// sets the calling thread's olap_last_error() and returns `code`
OLAP_INTERNAL int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
OLAP_INTERNAL int hip_fail(hipError_t e, const char *what);

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e__ = (expr);                           \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)

// remembers the calling thread's device and puts it back
struct DeviceGuard {
  int saved = -1;
  DeviceGuard() { (void)hipGetDevice(&saved); }
  ~DeviceGuard() {
    if (saved >= 0) (void)hipSetDevice(saved);
  }
};

// pooled device memory of the CURRENT device (olap_capi.hip: DevicePool)
OLAP_INTERNAL hipError_t dev_alloc(void **out, size_t bytes);
OLAP_INTERNAL void dev_free(void *p);

OLAP_INTERNAL int check_dtype(int dtype);
OLAP_INTERNAL int check_default(int kind);
OLAP_INTERNAL int require_device();

// One measure's cells on one device (in-memory.js:7-64).  values are always resident; the Int32
// mask is materialised lazily except where it is primary (integer cells over a NaN default).
struct olap_store {
  uint64_t size;
  int dtype;
  int default_kind;
  int device;               // the HIP device the buffers live on
  void *values;
  mutable int32_t *status;  // nullptr until needed
  // Insertion order of the reference's Map (in-memory.js:298 iterates it), kept only on request
  // (olap_store_track_order) and only once it differs from ascending flat index: seq[i] > 0 <=> cell i is set,
  // and set cells compare by seq as the reference's Map entries compare by age.  olap_order.hip.
  bool track_order = false;
  mutable uint32_t *seq = nullptr;   // nullptr: the order is the ascending flat index
  mutable uint64_t next_seq = 1;     // next value to hand out
  bool maybe_nonempty = false;       // some cell may be set (false only for a store nothing was written to)
  uint64_t hi_index = 0;             // no set cell lies above this index (valid while seq == nullptr)
};

// The handle layer runs every operation on the device its store lives on, whatever device the
// calling thread had current (one process may drive several GPUs: the sharded stores hand whole
// results back as plain stores on their first device), and puts the caller's device back on exit.
struct OnStoreDevice : DeviceGuard {
  explicit OnStoreDevice(const olap_store *s) {
    if (s && saved >= 0 && s->device != saved) (void)hipSetDevice(s->device);
  }
};

OLAP_INTERNAL bool mask_is_primary(const olap_store *s);
OLAP_INTERNAL const int32_t *mask_needed(const olap_store *s);
OLAP_INTERNAL int store_alloc(olap_store **out, uint64_t size, int dtype, int default_kind);
OLAP_INTERNAL int ensure_status(const olap_store *s);
OLAP_INTERNAL void drop_lazy_status(olap_store *s);

// ---- reorder of 4-byte cells as a two-axis LDS transpose (olap_transpose.hip) ---------------------
constexpr int kTransposeMaxAxis = 4;    // dimensions merged into the X (source-contiguous) or Y (destination-contiguous) axis
constexpr int kTransposeMaxBatch = 12;  // remaining dimensions: one coordinate per workgroup
struct TransposeXY {
  int nx, ny, nb;
  uint32_t len_x[kTransposeMaxAxis];         // X chain, fastest source dimension first
  uint64_t out_stride_x[kTransposeMaxAxis];  // their strides in the destination
  uint32_t len_y[kTransposeMaxAxis];         // Y chain, fastest destination dimension first
  uint64_t in_stride_y[kTransposeMaxAxis];   // their strides in the source
  uint32_t len_b[kTransposeMaxBatch];
  uint64_t in_stride_b[kTransposeMaxBatch], out_stride_b[kTransposeMaxBatch];
  uint64_t lx, ly;                           // merged axis lengths
  uint64_t tiles_x, tiles_y, batch;
  int tx, ty;                                // tile extents (cells)
  int super;                                 // tiles are walked in super x super blocks
  int y_first;                               // tile order: Y fastest (write locality) instead of X fastest (read locality)
  int cached_stores;                         // OLAP_XY_CACHED_STORES=1: plain instead of streaming stores (L2 may merge neighbouring tiles' pieces)
  int vec_in, vec_out;                       // every tile row starts 16-byte aligned on that side
  int default_test;                          // how a generated mask tells the default: 0 int/0, 1 float/0, 2 float/NaN, 3 never
};
OLAP_INTERNAL hipError_t launch_transpose_xy(const TransposeXY &t, int cell_bytes, const void *in, void *out, int32_t *st_out, bool aligned16,
                                              hipStream_t stream);

// ---- insertion order of tracked stores (olap_order.hip) --------------------------------------------
OLAP_INTERNAL void order_free(olap_store *s);
OLAP_INTERNAL int order_clone(const olap_store *from, olap_store *to);
OLAP_INTERNAL int order_before_bulk_write(olap_store *s);
OLAP_INTERNAL int order_after_bulk_write(olap_store *s);
OLAP_INTERNAL int order_before_set_value(olap_store *s, uint64_t index);
OLAP_INTERNAL int order_after_set_value(olap_store *s, uint64_t index);
OLAP_INTERNAL int order_after_from_sparse(olap_store *s, const uint32_t *host_indexes, uint64_t n);
OLAP_INTERNAL int order_drillup(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                                const uint32_t *const *maps, int method);
OLAP_INTERNAL int order_after_dice(const olap_store *s, olap_store *out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                                   const int32_t *const *sel);
OLAP_INTERNAL int order_after_reorder(const olap_store *s, olap_store *out, int ndim, const uint32_t *old_len, const int32_t *perm);
OLAP_INTERNAL int order_after_drilldown(const olap_store *s, olap_store *out);
OLAP_INTERNAL int order_before_load(olap_store *mine);
OLAP_INTERNAL int order_after_load(olap_store *mine, const olap_store *his, int ndim, const uint32_t *my_len, const uint32_t *his_len,
                                   const int32_t *const *his_to_mine);
// host copy of the set cells' indices in insertion order (ascending when the store has no seq)
OLAP_INTERNAL int order_sorted_keys(const olap_store *s, std::vector<uint64_t> &keys);
// the untracked operations of olap_capi.hip (what the public entry points do for a store that is not tracked)
OLAP_INTERNAL int store_drillup_plain(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                                      const uint32_t *const *maps, int method);
