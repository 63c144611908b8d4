// olap_internal.hpp — what the translation units of libolapgpu share besides the C ABI: error
// reporting, the device memory pool and the layout of a store handle.  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
};

OLAP_INTERNAL bool mask_is_primary(const olap_store *s);
OLAP_INTERNAL const int32_t *mask_needed(const olap_store *s);
OLAP_INTERNAL int store_alloc(olap_store **out, uint64_t size, int dtype, int default_kind);
OLAP_INTERNAL int ensure_status(const olap_store *s);
OLAP_INTERNAL void drop_lazy_status(olap_store *s);
