// olap_totals.hip — every marginal of a measure in one call: getNestedObject(measure, withTotals)
// (/root/reference/src/cube.js:421-440).
//
// The reference builds the totals by running, for EACH of the 2^D subsets of dimensions, the chain
// drillUp(dim_i, 'all') over the subset's dimensions in ascending order, each from the full cube
// (2^D chains, D * 2^(D-1) store passes).  The chain of subset s is the chain of (s without its
// highest dimension) plus one more roll-up, so all 2^D results are the cells of ONE extended cube E
// of shape (len_0 + 1) x ... x (len_{D-1} + 1) — index len_d of dimension d meaning 'all' — filled in
// D stages, stage d writing the slab "d = all" from the cells "d = 0 .. len_d - 1" of what the earlier
// stages left (dimensions below d already extended, above d not yet): same operations, same order,
// same per-stage rounding to the cell type as the chain of store calls, hence the same values.
//
//   * E fits in LDS (<= kLdsCells cells: the cubes getNestedObject is actually used on): ONE launch, one
//     workgroup; the cube is read from HBM once, every stage runs out of LDS, E is written once.
//   * larger: E lives in HBM; one scatter of the cube into E, then one launch per dimension
//     (D + 2 launches instead of 2^D - 1), lanes along the innermost extended dimension (coalesced).
// Per-cell semantics are those of the drillUp kernels: Agg<> replays in-memory.js:282-331.
#include <hip/hip_runtime.h>

#include <vector>

#include "olap_device.hpp"
#include "olap_internal.hpp"

using namespace olap;

namespace {

constexpr int kTotalsMaxDims = 16;
constexpr uint32_t kLdsCells = 12288;  // extended cells held in LDS: 12288 * (8 + 1) B = 108 KiB at most

struct TotalsShape {
  int nd;
  uint32_t len[kTotalsMaxDims];     // cube extents
  uint64_t pitch[kTotalsMaxDims];   // extended pitches: prod_{j > d} (len_j + 1)
  int method[kTotalsMaxDims];       // the measure's rule for each dimension
  uint64_t cells;                   // cube cells
  uint64_t ext;                     // extended cells
  int def_nan;
};

// one output cell of stage d: aggregates the K = len_d cells base + k * pitch_d
template <typename T, int METHOD, typename V, typename F>
__device__ __forceinline__ void stage_cell(const V *val, const F *flag, uint64_t base, uint32_t K, uint64_t pitch, bool def_nan, T &ov, int32_t &os) {
  Agg<METHOD> agg;
  agg.init();
  for (uint32_t k = 0; k < K; ++k) {
    const uint64_t at = base + (uint64_t)k * pitch;
    if (flag[at]) agg.add(Cell<T>::to_f64((T)val[at]), def_nan);
  }
  agg.finish(def_nan);
  emit_cell<T>(agg.acc, agg.has, def_nan, ov, os);
}

template <typename T, typename V, typename F>
__device__ __forceinline__ void stage_cell_any(int method, const V *val, const F *flag, uint64_t base, uint32_t K, uint64_t pitch, bool def_nan, T &ov,
                                               int32_t &os) {
  switch (method) {
    case OLAP_SUM: stage_cell<T, OLAP_SUM>(val, flag, base, K, pitch, def_nan, ov, os); break;
    case OLAP_AVERAGE: stage_cell<T, OLAP_AVERAGE>(val, flag, base, K, pitch, def_nan, ov, os); break;
    case OLAP_HIGHEST: stage_cell<T, OLAP_HIGHEST>(val, flag, base, K, pitch, def_nan, ov, os); break;
    case OLAP_LOWEST: stage_cell<T, OLAP_LOWEST>(val, flag, base, K, pitch, def_nan, ov, os); break;
    case OLAP_FIRST: stage_cell<T, OLAP_FIRST>(val, flag, base, K, pitch, def_nan, ov, os); break;
    case OLAP_LAST: stage_cell<T, OLAP_LAST>(val, flag, base, K, pitch, def_nan, ov, os); break;
    default: stage_cell<T, OLAP_PRODUCT>(val, flag, base, K, pitch, def_nan, ov, os); break;
  }
}

// flat cube index -> extended index (same digits, extended pitches)
__device__ __forceinline__ uint64_t cube_to_ext(const TotalsShape &s, uint64_t i) {
  uint64_t e = 0;
  for (int d = s.nd - 1; d >= 0; --d) {
    e += (i % s.len[d]) * s.pitch[d];
    i /= s.len[d];
  }
  return e;
}

// output o of stage d (dimensions below d extended, d itself rolled up, above d not extended) -> extended
// index of its first member (digit 0 of dimension d)
__device__ __forceinline__ uint64_t stage_base(const TotalsShape &s, int d, uint64_t o) {
  uint64_t e = 0;
  for (int j = s.nd - 1; j >= 0; --j) {
    if (j == d) continue;
    const uint64_t radix = j < d ? (uint64_t)s.len[j] + 1 : (uint64_t)s.len[j];
    e += (o % radix) * s.pitch[j];
    o /= radix;
  }
  return e;
}

__host__ __device__ inline uint64_t stage_outputs(const TotalsShape &s, int d) {
  uint64_t n = 1;
  for (int j = 0; j < s.nd; ++j)
    if (j != d) n *= j < d ? (uint64_t)s.len[j] + 1 : (uint64_t)s.len[j];
  return n;
}

// converts one finished extended cell for the host: float64 value (NaN where an integer cell is unset
// under a NaN default, like olap_store_get_data_f64) + mask
template <typename T>
__device__ __forceinline__ void export_cell(T v, bool set, bool def_nan, double &ov, int32_t &os) {
  ov = set ? Cell<T>::to_f64(v) : (def_nan ? __builtin_nan("") : 0.0);
  os = set ? OLAP_STATUS_SET : 0;
}

// ---- E in LDS: one launch, the cube is read once -------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void totals_lds_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in, double *__restrict__ out,
                                                          int32_t *__restrict__ st_out, const TotalsShape s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *val = reinterpret_cast<T *>(lds_raw);
  unsigned char *flag = lds_raw + (size_t)((s.ext * sizeof(T) + 15) & ~(uint64_t)15);
  const bool def_nan = s.def_nan != 0;
  for (uint64_t i = threadIdx.x; i < s.cells; i += blockDim.x) {
    const T x = in[i];
    const bool set = cell_is_set<T>(x, st_in ? st_in[i] : OLAP_STATUS_SET, st_in != nullptr, def_nan);
    const uint64_t e = cube_to_ext(s, i);
    val[e] = set ? x : Cell<T>::default_value(def_nan);
    flag[e] = set ? 1 : 0;
  }
  __syncthreads();
  for (int d = 0; d < s.nd; ++d) {
    const uint64_t n_out = stage_outputs(s, d);
    for (uint64_t o = threadIdx.x; o < n_out; o += blockDim.x) {
      const uint64_t base = stage_base(s, d, o);
      T ov;
      int32_t os;
      stage_cell_any<T>(s.method[d], val, flag, base, s.len[d], s.pitch[d], def_nan, ov, os);
      const uint64_t at = base + (uint64_t)s.len[d] * s.pitch[d];
      val[at] = ov;
      flag[at] = os ? 1 : 0;
    }
    __syncthreads();
  }
  for (uint64_t e = threadIdx.x; e < s.ext; e += blockDim.x) {
    double ov;
    int32_t os;
    export_cell<T>(val[e], flag[e] != 0, def_nan, ov, os);
    out[e] = ov;
    if (st_out) st_out[e] = os;
  }
}

// ---- E in HBM: scatter, one launch per dimension, export -----------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void totals_fill_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in, T *__restrict__ val,
                                                             int32_t *__restrict__ flag, const TotalsShape s) {
  const bool def_nan = s.def_nan != 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < s.cells; i += (uint64_t)gridDim.x * kBlock) {
    const T x = in[i];
    const bool set = cell_is_set<T>(x, st_in ? st_in[i] : OLAP_STATUS_SET, st_in != nullptr, def_nan);
    const uint64_t e = cube_to_ext(s, i);
    val[e] = set ? x : Cell<T>::default_value(def_nan);
    flag[e] = set ? OLAP_STATUS_SET : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void totals_stage_kernel(T *__restrict__ val, int32_t *__restrict__ flag, const TotalsShape s, int d, uint64_t n_out) {
  const bool def_nan = s.def_nan != 0;
  for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < n_out; o += (uint64_t)gridDim.x * kBlock) {
    const uint64_t base = stage_base(s, d, o);
    T ov;
    int32_t os;
    stage_cell_any<T>(s.method[d], val, flag, base, s.len[d], s.pitch[d], def_nan, ov, os);
    const uint64_t at = base + (uint64_t)s.len[d] * s.pitch[d];
    val[at] = ov;
    flag[at] = os;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void totals_export_kernel(const T *__restrict__ val, const int32_t *__restrict__ flag, double *__restrict__ out,
                                                               int32_t *__restrict__ st_out, uint64_t n, int def_nan) {
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (uint64_t)gridDim.x * kBlock) {
    double ov;
    int32_t os;
    export_cell<T>(val[e], flag[e] != 0, def_nan != 0, ov, os);
    out[e] = ov;
    if (st_out) st_out[e] = os;
  }
}

unsigned stride_grid(uint64_t n) {
  const uint64_t want = (n + kBlock - 1) / kBlock;
  return (unsigned)(want < 1 ? 1 : (want < 4096 ? want : 4096));
}

template <typename T>
int totals_typed(const olap_store *st, const TotalsShape &s, double *dev_out, int32_t *dev_status, int *launches, uint64_t *bytes_read) {
  const T *in = (const T *)st->values;
  const int32_t *st_in = mask_needed(st);
  const uint64_t mask_bytes = st_in ? 4 : 0;
  if (s.ext <= kLdsCells) {
    const size_t lds = (size_t)((s.ext * sizeof(T) + 15) & ~(uint64_t)15) + (size_t)s.ext;
    static PerDeviceFlag raised;  // (per cell type, per device)
    if (lds > 48 * 1024 && !raised.test_and_set())
      HIP_TRY(hipFuncSetAttribute((const void *)totals_lds_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsCells * (sizeof(T) + 1) + 16)));
    const unsigned threads = s.ext >= 4096 ? 1024 : 256;
    hipLaunchKernelGGL((totals_lds_kernel<T>), 1, threads, lds, nullptr, in, st_in, dev_out, dev_status, s);
    HIP_TRY(hipGetLastError());
    if (launches) *launches = 1;
    if (bytes_read) *bytes_read = s.cells * (sizeof(T) + mask_bytes);  // the cube, once
    return OLAP_OK;
  }
  T *val = nullptr;
  int32_t *flag = nullptr;
  HIP_TRY(dev_alloc((void **)&val, s.ext * sizeof(T)));
  hipError_t e = dev_alloc((void **)&flag, s.ext * sizeof(int32_t));
  if (e != hipSuccess) {
    dev_free(val);
    return hip_fail(e, "hipMalloc(totals)");
  }
  uint64_t bytes = s.cells * (sizeof(T) + mask_bytes);
  int n_launch = 0;
  hipLaunchKernelGGL((totals_fill_kernel<T>), stride_grid(s.cells), kBlock, 0, nullptr, in, st_in, val, flag, s);
  ++n_launch;
  for (int d = 0; d < s.nd; ++d) {
    const uint64_t n_out = stage_outputs(s, d);
    hipLaunchKernelGGL((totals_stage_kernel<T>), stride_grid(n_out), kBlock, 0, nullptr, val, flag, s, d, n_out);
    ++n_launch;
    bytes += n_out * s.len[d] * (sizeof(T) + 4);
  }
  hipLaunchKernelGGL((totals_export_kernel<T>), stride_grid(s.ext), kBlock, 0, nullptr, val, flag, dev_out, dev_status, s.ext, s.def_nan);
  ++n_launch;
  bytes += s.ext * (sizeof(T) + 4);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // val / flag go back to the pool
  dev_free(val);
  dev_free(flag);
  if (e != hipSuccess) return hip_fail(e, "totals");
  if (launches) *launches = n_launch;
  if (bytes_read) *bytes_read = bytes;
  return OLAP_OK;
}

}  // namespace

extern "C" int olap_store_totals(const olap_store *st, int ndim, const uint32_t *lens, const int *methods, double *host_values,
                                 int32_t *host_status, int *launches, uint64_t *bytes_read) {
  OnStoreDevice on_device__(st);
  if (!st) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (ndim < 0 || ndim > kTotalsMaxDims) return fail(OLAP_ERR_INVALID_ARGUMENT, "totals: at most %d dimensions", kTotalsMaxDims);
  if (ndim > 0 && (!lens || !methods)) return fail(OLAP_ERR_INVALID_ARGUMENT, "lens/methods is NULL");
  if (!host_values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if (st->track_order)  // every intermediate marginal has an order of its own (first hit), which later `first` / `last` stages follow
    return fail(OLAP_ERR_INVALID_ARGUMENT, "ordered: this store tracks its insertion order; run the chain of drillUps instead");
  TotalsShape s{};
  s.nd = ndim;
  s.def_nan = st->default_kind == OLAP_DEFAULT_NAN;
  long double cells = 1, ext = 1;
  for (int d = 0; d < ndim; ++d) {
    if (methods[d] < OLAP_SUM || methods[d] > OLAP_PRODUCT) return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", methods[d]);
    s.len[d] = lens[d];
    s.method[d] = methods[d];
    cells *= lens[d];
    ext *= (long double)lens[d] + 1;
  }
  if (ext > 4.0e9L) return fail(OLAP_ERR_INVALID_ARGUMENT, "totals: the extended cube would hold %.3Lg cells", ext);
  s.cells = (uint64_t)cells;
  s.ext = (uint64_t)ext;
  if (s.cells != st->size)
    return fail(OLAP_ERR_LENGTH_MISMATCH, "store holds %llu cells but the dimensions describe %llu", (unsigned long long)st->size, (unsigned long long)s.cells);
  uint64_t pitch = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    s.pitch[d] = pitch;
    pitch *= (uint64_t)lens[d] + 1;
  }
  int rc = require_device();
  if (rc) return rc;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(st->device));
  double *dev_out = nullptr;
  int32_t *dev_status = nullptr;
  HIP_TRY(dev_alloc((void **)&dev_out, s.ext * sizeof(double)));
  if (host_status) {
    hipError_t e = dev_alloc((void **)&dev_status, s.ext * sizeof(int32_t));
    if (e != hipSuccess) {
      dev_free(dev_out);
      return hip_fail(e, "hipMalloc(totals)");
    }
  }
  switch (st->dtype) {
    case OLAP_INT32: rc = totals_typed<int32_t>(st, s, dev_out, dev_status, launches, bytes_read); break;
    case OLAP_UINT32: rc = totals_typed<uint32_t>(st, s, dev_out, dev_status, launches, bytes_read); break;
    case OLAP_FLOAT32: rc = totals_typed<float>(st, s, dev_out, dev_status, launches, bytes_read); break;
    default: rc = totals_typed<double>(st, s, dev_out, dev_status, launches, bytes_read); break;
  }
  hipError_t e = hipSuccess;
  if (!rc) e = hipMemcpy(host_values, dev_out, s.ext * sizeof(double), hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && host_status) e = hipMemcpy(host_status, dev_status, s.ext * sizeof(int32_t), hipMemcpyDeviceToHost);
  dev_free(dev_out);
  if (dev_status) dev_free(dev_status);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "totals");
  return OLAP_OK;
}
