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
//   * larger: the dimensions are cut into GROUPS of consecutive dimensions and each group's stages run fused out of
//     LDS (totals_group_kernel): a pass reads the cube-so-far once — a compact tensor [P, group dims, Q], P = the
//     earlier dimensions already extended, Q = the later ones not yet — stages a tile of (one p, a run of q) through
//     LDS at its extended positions, runs the group's stages there in the reference's order, and writes the extended
//     tile back, coalesced along q.  The innermost group's tiles are contiguous in E, and that pass writes the
//     float64 export itself.  [10]^6: 2 passes reading 2.2 x the cube (round 2: a scatter + 6 stage launches + an
//     export reading 15 x, in runs of 11 cells).  A dimension too long for a tile takes a register pass (lane per q,
//     loop over its items); only a cube whose INNERMOST dimension alone exceeds a tile keeps the round-2 form.
// Per-cell semantics are those of the drillUp kernels: Agg<> replays in-memory.js:282-331.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
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

// ---- E in HBM, stages fused per group of dimensions through LDS ---------------------------------------------------------
constexpr int kGroupMaxDims = 8;
struct GroupPass {
  int nd;                          // dimensions of the group, outermost first
  uint32_t len[kGroupMaxDims];
  uint32_t epitch[kGroupMaxDims];  // extended pitch inside the group: prod_{j > i} (len_j + 1)
  int method[kGroupMaxDims];
  uint32_t base_cells;             // prod len
  uint32_t ext_cells;              // prod (len + 1)
  uint64_t P, Q;                   // earlier dimensions (extended) / later dimensions (not yet), as flat counts
  uint32_t W;                      // Q > 1: cells of q per tile;  Q == 1: tiles (values of p) per workgroup
  int def_nan;
};

// position inside the group's extended sub-cube of base cell b (digits over len -> pitches over len + 1)
__device__ __forceinline__ uint32_t group_ext_of(const GroupPass &g, uint32_t b) {
  uint32_t e = 0;
#pragma unroll
  for (int i = kGroupMaxDims - 1; i >= 0; --i)
    if (i < g.nd) {
      const uint32_t q = b / g.len[i];
      e += (b - q * g.len[i]) * g.epitch[i];
      b = q;
    }
  return e;
}
// output o of stage i inside the group (dimensions before i extended, i rolled up, after i not): extended position of its first member
__device__ __forceinline__ uint32_t group_stage_base(const GroupPass &g, int i, uint32_t o) {
  uint32_t e = 0;
#pragma unroll
  for (int j = kGroupMaxDims - 1; j >= 0; --j)
    if (j < g.nd && j != i) {
      const uint32_t radix = j < i ? g.len[j] + 1 : g.len[j];
      const uint32_t q = o / radix;
      e += (o - q * radix) * g.epitch[j];
      o = q;
    }
  return e;
}
inline uint32_t group_stage_outputs(const GroupPass &g, int i) {
  uint64_t n = 1;
  for (int j = 0; j < g.nd; ++j)
    if (j != i) n *= j < i ? (uint64_t)g.len[j] + 1 : (uint64_t)g.len[j];
  return (uint32_t)n;
}

// LAST: the innermost group (Q == 1): W tiles per workgroup, each contiguous in the input and in E; writes the export.
// Otherwise: tile = (one p, W consecutive q); LDS cell (e, w) at e * W + w; writes the compact tensor [P, ext, Q].
template <typename T, bool LAST>
__global__ __launch_bounds__(1024) void totals_group_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in, const unsigned char *__restrict__ fl_in,
                                                            T *__restrict__ out, unsigned char *__restrict__ fl_out, double *__restrict__ ex_out,
                                                            int32_t *__restrict__ ex_st, const GroupPass g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t W = g.W;
  const uint32_t cap = g.ext_cells * W;  // LDS cells of this workgroup
  T *val = reinterpret_cast<T *>(lds_raw);
  unsigned char *flag = lds_raw + (size_t)(((size_t)cap * sizeof(T) + 15) & ~(size_t)15);
  const bool def_nan = g.def_nan != 0;
  uint64_t p0, q0;
  uint32_t wn;  // live columns (q) or tiles (p) of this workgroup
  if constexpr (LAST) {
    p0 = (uint64_t)blockIdx.x * W;
    q0 = 0;
    wn = (uint32_t)((g.P - p0) < W ? (g.P - p0) : W);
  } else {
    const uint64_t qb = (g.Q + W - 1) / W;
    p0 = blockIdx.x / qb;
    q0 = (blockIdx.x % qb) * W;
    wn = (uint32_t)((g.Q - q0) < W ? (g.Q - q0) : W);
  }
  // ---- load the base cells to their extended positions
  if constexpr (LAST) {
    const uint64_t first = p0 * g.base_cells;
    const uint32_t n = wn * g.base_cells;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
      const uint32_t r = idx / g.base_cells, b = idx - r * g.base_cells;
      const T x = in[first + idx];
      const bool set = fl_in ? fl_in[first + idx] != 0 : cell_is_set<T>(x, st_in ? st_in[first + idx] : OLAP_STATUS_SET, st_in != nullptr, def_nan);
      const uint32_t at = r * g.ext_cells + group_ext_of(g, b);
      val[at] = set ? x : Cell<T>::default_value(def_nan);
      flag[at] = set ? 1 : 0;
    }
  } else {
    const uint32_t n = g.base_cells * wn;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
      const uint32_t b = idx / wn, w = idx - b * wn;
      const uint64_t src = (p0 * g.base_cells + b) * g.Q + q0 + w;
      const T x = in[src];
      const bool set = fl_in ? fl_in[src] != 0 : cell_is_set<T>(x, st_in ? st_in[src] : OLAP_STATUS_SET, st_in != nullptr, def_nan);
      const uint32_t at = group_ext_of(g, b) * W + w;
      val[at] = set ? x : Cell<T>::default_value(def_nan);
      flag[at] = set ? 1 : 0;
    }
  }
  __syncthreads();
  // ---- the group's stages, in the reference's order, out of LDS
  for (int i = 0; i < g.nd; ++i) {
    uint32_t n_out = 1;
    for (int j = 0; j < g.nd; ++j)
      if (j != i) n_out *= j < i ? g.len[j] + 1 : g.len[j];
    const uint32_t n = n_out * wn;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
      uint32_t o, w, base, pitch;
      if constexpr (LAST) {
        w = idx / n_out;
        o = idx - w * n_out;
        base = w * g.ext_cells + group_stage_base(g, i, o);
        pitch = g.epitch[i];
      } else {
        o = idx / wn;
        w = idx - o * wn;
        base = group_stage_base(g, i, o) * W + w;
        pitch = g.epitch[i] * W;
      }
      T ov;
      int32_t os;
      stage_cell_any<T>(g.method[i], val, flag, base, g.len[i], pitch, def_nan, ov, os);
      const uint32_t at = base + g.len[i] * pitch;
      val[at] = ov;
      flag[at] = os ? 1 : 0;
    }
    __syncthreads();
  }
  // ---- store the extended tile
  if constexpr (LAST) {
    const uint64_t first = p0 * g.ext_cells;
    const uint32_t n = wn * g.ext_cells;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
      double ov;
      int32_t os;
      export_cell<T>(val[idx], flag[idx] != 0, def_nan, ov, os);
      ex_out[first + idx] = ov;
      if (ex_st) ex_st[first + idx] = os;
    }
  } else {
    const uint32_t n = g.ext_cells * wn;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
      const uint32_t e = idx / wn, w = idx - e * wn;
      const uint64_t dst = (p0 * g.ext_cells + e) * g.Q + q0 + w;
      out[dst] = val[e * W + w];
      if (fl_out) fl_out[dst] = flag[e * W + w];
    }
  }
}

// One dimension too long for a tile: compact [P, K, Q] -> [P, K + 1, Q], a lane per (p, q) walks the K items
// (coalesced along q) copying them and writes their aggregate as item K.
template <typename T>
__global__ __launch_bounds__(kBlock) void totals_long_stage_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in, const unsigned char *__restrict__ fl_in,
                                                                   T *__restrict__ out, unsigned char *__restrict__ fl_out, uint64_t P, uint32_t K, uint64_t Q,
                                                                   int method, int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= P * Q) return;
  const uint64_t p = t / Q, q = t - p * Q;
  const T *src = in + p * K * Q + q;
  T *dst = out + p * ((uint64_t)K + 1) * Q + q;
  unsigned char *fdst = fl_out ? fl_out + p * ((uint64_t)K + 1) * Q + q : nullptr;
  auto run = [&](auto tag) {
    constexpr int METHOD = decltype(tag)::value;
    Agg<METHOD> agg;
    agg.init();
    for (uint32_t k = 0; k < K; ++k) {
      const uint64_t at = (uint64_t)k * Q;
      const T x = src[at];
      const bool set = fl_in ? fl_in[p * K * Q + q + at] != 0
                             : cell_is_set<T>(x, st_in ? st_in[p * K * Q + q + at] : OLAP_STATUS_SET, st_in != nullptr, def_nan);
      dst[at] = set ? x : Cell<T>::default_value(def_nan);
      if (fdst) fdst[at] = set ? 1 : 0;
      if (set) agg.add(Cell<T>::to_f64(x), def_nan);
    }
    agg.finish(def_nan);
    T ov;
    int32_t os;
    emit_cell<T>(agg.acc, agg.has, def_nan, ov, os);
    dst[(uint64_t)K * Q] = ov;
    if (fdst) fdst[(uint64_t)K * Q] = os ? 1 : 0;
  };
  switch (method) {
    case OLAP_SUM: run(std::integral_constant<int, OLAP_SUM>{}); break;
    case OLAP_AVERAGE: run(std::integral_constant<int, OLAP_AVERAGE>{}); break;
    case OLAP_HIGHEST: run(std::integral_constant<int, OLAP_HIGHEST>{}); break;
    case OLAP_LOWEST: run(std::integral_constant<int, OLAP_LOWEST>{}); break;
    case OLAP_FIRST: run(std::integral_constant<int, OLAP_FIRST>{}); break;
    case OLAP_LAST: run(std::integral_constant<int, OLAP_LAST>{}); break;
    default: run(std::integral_constant<int, OLAP_PRODUCT>{}); break;
  }
}

// How the dimensions are cut into passes (host).  Groups are chosen from the innermost dimension outwards — the
// innermost group as large as a tile allows (its tiles are contiguous), then the others with at least kMinRun cells of q per
// tile — and executed from the outermost inwards, the order of the reference's chain.
struct PassPlan {
  int first, count;  // dimensions [first, first + count)
  bool lds;          // fused through LDS; false: one long dimension in registers
};
constexpr uint32_t kMinRun = 64;  // cells of q a tile moves per (group cell): 256-byte runs at least
// LDS of one workgroup: 80 KB leaves room for two workgroups per CU (one loads while the other runs its stages); a
// tile of up to 150 KB owns the CU, which pays when it swallows a whole pass over the cube
constexpr size_t kTileBytes2 = 80 * 1024, kTileBytes1 = 150 * 1024;

inline bool plan_passes_for(const TotalsShape &s, size_t cell_bytes, size_t inner_budget, std::vector<PassPlan> *out, double *cost) {
  out->clear();
  const uint64_t cap_inner = inner_budget / cell_bytes, cap = kTileBytes2 / cell_bytes;
  int a = s.nd;
  uint64_t ext = 1;
  while (a > 0 && a > s.nd - kGroupMaxDims && ext * ((uint64_t)s.len[a - 1] + 1) <= cap_inner) ext *= (uint64_t)s.len[--a] + 1;
  if (a == s.nd) return false;  // the innermost dimension alone does not fit a tile
  std::vector<PassPlan> rev{{a, s.nd - a, true}};
  uint64_t Q = 1;
  for (int d = a; d < s.nd; ++d) Q *= s.len[d];
  int e = a;
  while (e > 0) {
    const uint64_t run = Q < kMinRun ? Q : kMinRun;
    int b = e;
    uint64_t gext = 1;
    while (b > 0 && e - b < kGroupMaxDims && gext * ((uint64_t)s.len[b - 1] + 1) * run <= cap) gext *= (uint64_t)s.len[--b] + 1;
    if (b == e) {  // one long dimension
      rev.push_back({e - 1, 1, false});
      b = e - 1;
    } else {
      rev.push_back({b, e - b, true});
    }
    for (int d = b; d < e; ++d) Q *= s.len[d];
    e = b;
  }
  out->assign(rev.rbegin(), rev.rend());
  // cells moved: every pass reads the cube-so-far and writes it extended by its group (the last one writes the export)
  double moved = 0, cells = (double)s.cells;
  for (const PassPlan &pp : *out) {
    double grow = 1;
    for (int d = pp.first; d < pp.first + pp.count; ++d) grow *= ((double)s.len[d] + 1) / (double)s.len[d];
    const bool last = &pp == &out->back();
    moved += (cells + cells * grow * (last ? 3.0 : 1.0)) * (last && ext * cell_bytes > kTileBytes2 ? 1.25 : 1.0);
    cells *= grow;
  }
  *cost = moved;
  return true;
}
// the cheaper of: the innermost group inside 80 KB, or as large as one workgroup per CU can hold
inline bool plan_passes(const TotalsShape &s, size_t cell_bytes, std::vector<PassPlan> *out) {
  std::vector<PassPlan> p2, p1;
  double c2 = 0, c1 = 0;
  const bool ok2 = plan_passes_for(s, cell_bytes, kTileBytes2, &p2, &c2), ok1 = plan_passes_for(s, cell_bytes, kTileBytes1, &p1, &c1);
  if (!ok2 && !ok1) return false;
  *out = (ok1 && (!ok2 || c1 < c2)) ? p1 : p2;
  return true;
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
  std::vector<PassPlan> passes;
  static const bool no_groups = getenv("OLAP_TOTALS_NO_GROUPS") != nullptr;  // A/B: the round-2 form (scatter + one launch per dimension + export)
  if (!no_groups && plan_passes(s, sizeof(T) + 1, &passes)) {
    // compact tensors between the passes: values (+ one flag byte per cell where the mask is primary)
    const bool primary = st_in != nullptr;
    uint64_t biggest = 0;
    {
      uint64_t P = 1, Qall = s.cells;
      for (size_t k = 0; k + 1 < passes.size(); ++k) {
        uint64_t base = 1, gext = 1;
        for (int d = passes[k].first; d < passes[k].first + passes[k].count; ++d) base *= s.len[d], gext *= (uint64_t)s.len[d] + 1;
        Qall /= base;
        P *= gext;
        biggest = std::max(biggest, P * Qall);
      }
    }
    T *buf[2] = {nullptr, nullptr};
    unsigned char *fbuf[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess && biggest && passes.size() > (size_t)(k + 1); ++k) {
      e = dev_alloc((void **)&buf[k], biggest * sizeof(T));
      if (e == hipSuccess && primary) e = dev_alloc((void **)&fbuf[k], biggest);
    }
    uint64_t bytes = 0;
    int n_launch = 0;
    const T *cur = in;
    const int32_t *cur_st = st_in;
    const unsigned char *cur_fl = nullptr;
    uint64_t P = 1, Q = s.cells;
    static PerDeviceFlag raised_a, raised_b;
    for (size_t k = 0; k < passes.size() && e == hipSuccess; ++k) {
      const PassPlan &pp = passes[k];
      const bool last = k + 1 == passes.size();
      uint64_t base = 1, gext = 1;
      for (int d = pp.first; d < pp.first + pp.count; ++d) base *= s.len[d], gext *= (uint64_t)s.len[d] + 1;
      Q /= base;
      T *dst = last ? nullptr : buf[k & 1];
      unsigned char *dfl = last ? nullptr : fbuf[k & 1];
      bytes += P * base * Q * (sizeof(T) + (cur_st ? 4 : 0) + (cur_fl ? 1 : 0));
      if (!pp.lds) {
        const uint64_t lanes = P * Q;
        hipLaunchKernelGGL((totals_long_stage_kernel<T>), (unsigned)((lanes + kBlock - 1) / kBlock), kBlock, 0, nullptr, cur, cur_st, cur_fl, dst, dfl, P,
                           s.len[pp.first], Q, s.method[pp.first], s.def_nan);
      } else {
        GroupPass g{};
        g.nd = pp.count;
        for (int i = 0; i < pp.count; ++i) {
          g.len[i] = s.len[pp.first + i];
          g.method[i] = s.method[pp.first + i];
        }
        uint32_t ep = 1;
        for (int i = pp.count - 1; i >= 0; --i) {
          g.epitch[i] = ep;
          ep *= g.len[i] + 1;
        }
        g.base_cells = (uint32_t)base;
        g.ext_cells = (uint32_t)gext;
        g.P = P;
        g.Q = Q;
        g.def_nan = s.def_nan;
        uint64_t w = (last ? std::max<uint64_t>(kTileBytes2 / (sizeof(T) + 1), gext) : kTileBytes2 / (sizeof(T) + 1)) / gext;
        if (last) {
          w = std::min<uint64_t>(w, P);
          // enough workgroups for the chip before tiles are packed densely
          while (w > 1 && (P + w - 1) / w < 1024) w = (w + 1) / 2;
        } else {
          w = std::min<uint64_t>(w, Q);
          if (w > 256) w = 256;
          if (w >= 64) w &= ~63ull;  // whole wavefronts along q
        }
        g.W = (uint32_t)std::max<uint64_t>(w, 1);
        const size_t lds = (size_t)((((size_t)g.ext_cells * g.W * sizeof(T)) + 15) & ~(size_t)15) + (size_t)g.ext_cells * g.W;
        const uint64_t blocks = last ? (P + g.W - 1) / g.W : P * ((Q + g.W - 1) / g.W);
        if (blocks > 0x7FFFFFFFull) {
          e = hipErrorInvalidValue;
          break;
        }
        const unsigned threads = (uint64_t)g.ext_cells * g.W >= 4096 ? 1024 : 256;
        const int max_lds = (int)(kTileBytes1 + 64);
        if (last) {
          if (lds > 48 * 1024 && !raised_b.test_and_set())
            e = hipFuncSetAttribute((const void *)totals_group_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
          if (e == hipSuccess)
            hipLaunchKernelGGL((totals_group_kernel<T, true>), (unsigned)blocks, threads, lds, nullptr, cur, cur_st, cur_fl, (T *)nullptr, (unsigned char *)nullptr,
                               dev_out, dev_status, g);
        } else {
          if (lds > 48 * 1024 && !raised_a.test_and_set())
            e = hipFuncSetAttribute((const void *)totals_group_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
          if (e == hipSuccess)
            hipLaunchKernelGGL((totals_group_kernel<T, false>), (unsigned)blocks, threads, lds, nullptr, cur, cur_st, cur_fl, dst, dfl, (double *)nullptr,
                               (int32_t *)nullptr, g);
        }
      }
      ++n_launch;
      P *= gext;
      cur = dst;
      cur_st = nullptr;
      cur_fl = dfl;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // the intermediate tensors go back to the pool
    for (int k = 0; k < 2; ++k) {
      if (buf[k]) dev_free(buf[k]);
      if (fbuf[k]) dev_free(fbuf[k]);
    }
    if (e != hipSuccess) return hip_fail(e, "totals");
    if (launches) *launches = n_launch;
    if (bytes_read) *bytes_read = bytes;
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
