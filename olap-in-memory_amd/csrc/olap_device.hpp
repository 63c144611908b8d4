// olap_device.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Cell semantics restate /root/reference/src/store/in-memory.js on dense typed buffers:
//   * a cell is set iff (status & 0x2, when a mask is given) and value != default  (:122-133)
//   * accumulation is float64 whatever the declared type                        (:282-290)
//   * results are stored with ECMAScript TypedArray conversion                  (:77-92)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/olap_hip.h"

// hipFuncSetAttribute is per device: "already raised" flags are kept per device (one process may drive several)
struct PerDeviceFlag {
  bool done[64] = {};
  bool test_and_set() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return false;  // unknown device: raise again (harmless)
    const bool was = done[dev];
    done[dev] = true;
    return was;
  }
};

namespace olap {

constexpr int kBlock = 256;  // 4 wavefronts of 64 lanes

// ---------------------------------------------------------------- typed <-> float64
__device__ __forceinline__ uint32_t js_to_uint32(double d) {
  // ToUint32: NaN / +-Inf -> 0, truncate toward zero, modulo 2^32
  if (!(fabs(d) < 1.7976931348623157e308)) return 0u;  // NaN or Inf
  double t = trunc(d);
  double m = fmod(t, 4294967296.0);
  if (m < 0) m += 4294967296.0;
  return (uint32_t)m;
}

template <typename T> struct Cell;
template <> struct Cell<float> {
  static constexpr int dtype = OLAP_FLOAT32;
  static __device__ __forceinline__ double to_f64(float v) { return (double)v; }
  static __device__ __forceinline__ float from_f64(double d) { return (float)d; }
  static __device__ __forceinline__ bool is_default(float v, bool def_nan) { return def_nan ? (v != v) : (v == 0.0f); }
  static __device__ __forceinline__ float default_value(bool def_nan) { return def_nan ? __builtin_nanf("") : 0.0f; }
};
template <> struct Cell<double> {
  static constexpr int dtype = OLAP_FLOAT64;
  static __device__ __forceinline__ double to_f64(double v) { return v; }
  static __device__ __forceinline__ double from_f64(double d) { return d; }
  static __device__ __forceinline__ bool is_default(double v, bool def_nan) { return def_nan ? (v != v) : (v == 0.0); }
  static __device__ __forceinline__ double default_value(bool def_nan) { return def_nan ? __builtin_nan("") : 0.0; }
};
template <> struct Cell<int32_t> {
  static constexpr int dtype = OLAP_INT32;
  static __device__ __forceinline__ double to_f64(int32_t v) { return (double)v; }
  static __device__ __forceinline__ int32_t from_f64(double d) { return (int32_t)js_to_uint32(d); }
  static __device__ __forceinline__ bool is_default(int32_t v, bool def_nan) { return def_nan ? false : (v == 0); }
  static __device__ __forceinline__ int32_t default_value(bool) { return 0; }
};
template <> struct Cell<uint32_t> {
  static constexpr int dtype = OLAP_UINT32;
  static __device__ __forceinline__ double to_f64(uint32_t v) { return (double)v; }
  static __device__ __forceinline__ uint32_t from_f64(double d) { return js_to_uint32(d); }
  static __device__ __forceinline__ bool is_default(uint32_t v, bool def_nan) { return def_nan ? false : (v == 0u); }
  static __device__ __forceinline__ uint32_t default_value(bool) { return 0u; }
};

__device__ __forceinline__ bool is_default_f64(double r, bool def_nan) { return def_nan ? (r != r) : (r == 0.0); }

// Math.max / Math.min (in-memory.js:285-286): NaN-propagating, +0 > -0
__device__ __forceinline__ double js_max(double a, double b) {
  if (a != a || b != b) return __builtin_nan("");
  if (a == 0.0 && b == 0.0) return __builtin_signbit(a) ? b : a;
  return a > b ? a : b;
}
__device__ __forceinline__ double js_min(double a, double b) {
  if (a != a || b != b) return __builtin_nan("");
  if (a == 0.0 && b == 0.0) return __builtin_signbit(a) ? a : b;
  return a < b ? a : b;
}

// ---------------------------------------------------------------- vector access (16 B per lane where it divides)
template <typename T, int N> struct alignas(sizeof(T) * N) Vec { T v[N]; };

template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_vec(const T *p) {
  return *reinterpret_cast<const Vec<T, N> *>(p);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T *p, const Vec<T, N> &x) {
  *reinterpret_cast<Vec<T, N> *>(p) = x;
}

// Streaming (non-temporal) forms for data that this kernel touches exactly once: on MI355X a
// 400 MB read stream reaches ~7.1 TB/s with `nt` loads against ~6.6 TB/s with plain loads
// (tools/ceilings.hip, profiles/ceilings_r01.txt).
template <int BYTES> struct RawVec;
template <> struct RawVec<4> { typedef uint32_t type; };
template <> struct RawVec<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct RawVec<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };

template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_stream(const T *p) {
  if constexpr (sizeof(T) * N > 16) {  // wider than one 16 B access: two halves
    union { Vec<T, N / 2> h[2]; Vec<T, N> v; } u;
    u.h[0] = load_stream<T, N / 2>(p);
    u.h[1] = load_stream<T, N / 2>(p + N / 2);
    return u.v;
  } else {
    typedef typename RawVec<sizeof(T) * N>::type R;
    union { R r; Vec<T, N> v; } u;
    u.r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
    return u.v;
  }
}
template <typename T, int N>
__device__ __forceinline__ void store_stream(T *p, const Vec<T, N> &x) {
  if constexpr (sizeof(T) * N > 16) {
    union { Vec<T, N / 2> h[2]; Vec<T, N> v; } u;
    u.v = x;
    store_stream<T, N / 2>(p, u.h[0]);
    store_stream<T, N / 2>(p + N / 2, u.h[1]);
  } else {
    typedef typename RawVec<sizeof(T) * N>::type R;
    union { R r; Vec<T, N> v; } u;
    u.v = x;
    __builtin_nontemporal_store(u.r, reinterpret_cast<R *>(p));
  }
}

// The same 16-byte streaming load from an address that is only CELL-aligned.  gfx950 takes global
// dwordx4 accesses at any 4-byte boundary (the compiler keeps one `global_load_dwordx4 … nt` for an
// under-aligned vector type), and tools/unaligned_probe.hip measures no cost on the load side (a
// 10^8-cell copy: 125 us with the source shifted by 1-3 cells against 123 us aligned; shifted
// STORES cost 7 %), so paths over rows that are not whole 16-byte groups need neither 4-byte lanes
// nor an LDS staging pass to realign.
template <int BYTES, int ALIGN> struct RawVecAt;
template <> struct RawVecAt<16, 4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct RawVecAt<16, 8> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(8))); };
template <> struct RawVecAt<8, 4> { typedef uint32_t type __attribute__((ext_vector_type(2), aligned(4))); };  // the status words of two 8-byte cells

template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_stream_cell_aligned(const T *p) {
  typedef typename RawVecAt<sizeof(T) * N, sizeof(T)>::type R;
  union U { typename RawVec<sizeof(T) * N>::type r; Vec<T, N> v; __device__ U() {} } u;
  u.r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
  return u.v;
}

template <typename T, int N>
__device__ __forceinline__ void store_stream_cell_aligned(T *p, const Vec<T, N> &x) {
  if constexpr (sizeof(T) * N > 16) {  // four float64 partials: two 16-byte halves
    union H { Vec<T, N / 2> h[2]; Vec<T, N> v; __device__ H() {} } u;
    u.v = x;
    store_stream_cell_aligned<T, N / 2>(p, u.h[0]);
    store_stream_cell_aligned<T, N / 2>(p + N / 2, u.h[1]);
  } else {
    typedef typename RawVecAt<sizeof(T) * N, sizeof(T)>::type R;
    union U { typename RawVec<sizeof(T) * N>::type r; Vec<T, N> v; __device__ U() {} } u;
    u.v = x;
    __builtin_nontemporal_store(u.r, reinterpret_cast<R *>(p));
  }
}

// ---------------------------------------------------------------- the per-output-cell aggregate
// Restates the body of the drillUp loop, in-memory.js:311-320: the first contribution stores the
// value, later ones store agg(current, value); setValue drops the key when the running value
// equals the default, so the next contribution starts over; contributions are counted apart.
template <int METHOD>
struct Agg {
  double acc;
  uint32_t count;
  bool has;

  __device__ __forceinline__ void init() {
    acc = 0.0;
    count = 0;
    has = false;
  }
  // `v` is the value of a SET input cell
  __device__ __forceinline__ void add(double v, bool def_nan) {
    if (!has) {
      acc = v;  // a set cell never holds the default, so the key now exists
      has = true;
    } else {
      double r;
      if constexpr (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE) r = acc + v;
      else if constexpr (METHOD == OLAP_HIGHEST) r = js_max(acc, v);
      else if constexpr (METHOD == OLAP_LOWEST) r = js_min(acc, v);
      else if constexpr (METHOD == OLAP_FIRST) r = acc;
      else if constexpr (METHOD == OLAP_LAST) r = v;
      else r = acc * v;
      if (is_default_f64(r, def_nan)) has = false;  // setValue(idx, default) deletes the key
      else acc = r;
    }
    ++count;
  }
  // in-memory.js:323-331
  __device__ __forceinline__ void finish(bool def_nan) {
    if constexpr (METHOD == OLAP_AVERAGE) {
      const uint32_t c16 = count & 0xFFFFu;  // Uint16Array counter
      if (c16) {
        const double cur = has ? acc : (def_nan ? __builtin_nan("") : 0.0);
        const double r = cur / (double)c16;
        has = !is_default_f64(r, def_nan);
        acc = r;
      }
    }
  }
};

// highest / lowest / first / last never leave the cell type: the result is one of the inputs
// (or NaN, Math.max/min propagate it), so no float64 round trip, no restart (a set cell never holds
// the default, so the running value never equals it) and no typed re-conversion are needed.
template <typename T> __device__ __forceinline__ T select_max(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T select_min(T a, T b) { return a < b ? a : b; }
template <> __device__ __forceinline__ float select_max<float>(float a, float b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == b) return __builtin_signbit(a) ? b : a;  // max(-0, +0) = +0
  return a > b ? a : b;
}
template <> __device__ __forceinline__ float select_min<float>(float a, float b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == b) return __builtin_signbit(a) ? a : b;
  return a < b ? a : b;
}
template <> __device__ __forceinline__ double select_max<double>(double a, double b) { return js_max(a, b); }
template <> __device__ __forceinline__ double select_min<double>(double a, double b) { return js_min(a, b); }

// Math.max / Math.min over float cells with the hardware's v_max / v_min: on gfx950 they order -0 below +0 (what
// Math.max / Math.min want) and DROP a NaN operand (tools/… probe: max(-0,+0) = max(+0,-0) = +0, min = -0, max(NaN,1) = 1),
// so the running extreme never holds a NaN and a separate flag remembers that one was seen — Math.max / Math.min
// propagate it.  Inline asm: the builtin would quiet both operands first (two more v_max per call).
__device__ __forceinline__ float hw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double hw_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double hw_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <typename T> struct IsFloatCell { static constexpr bool value = false; };
template <> struct IsFloatCell<float> { static constexpr bool value = true; };
template <> struct IsFloatCell<double> { static constexpr bool value = true; };

template <typename T, int METHOD>
struct Pick {
  // highest / lowest of float cells: `cur` is the running extreme of the non-NaN members (the identity until the first one)
  static constexpr bool kHw = IsFloatCell<T>::value && (METHOD == OLAP_HIGHEST || METHOD == OLAP_LOWEST);
  T cur;
  bool has;
  bool nan;  // (kHw) a set member was NaN
  __device__ __forceinline__ void init() {
    if constexpr (kHw) cur = METHOD == OLAP_HIGHEST ? -__builtin_huge_val() : __builtin_huge_val();
    else cur = T(0);
    has = false;
    nan = false;
  }
  __device__ __forceinline__ void add(T v) { add_if(true, v); }
  // the same as `if (set) add(v)` written as selects: no divergent branch per cell in the streaming kernels
  __device__ __forceinline__ void add_if(bool set, T v) {
    if constexpr (kHw) {
      const T ve = set ? v : cur;  // (cur is never NaN: an unset cell changes nothing)
      nan = nan || (ve != ve);
      if constexpr (METHOD == OLAP_HIGHEST) cur = hw_max(cur, ve);
      else cur = hw_min(cur, ve);
      has = has || set;
    } else {
      T next;
      if constexpr (METHOD == OLAP_HIGHEST) next = select_max<T>(cur, v);
      else if constexpr (METHOD == OLAP_LOWEST) next = select_min<T>(cur, v);
      else if constexpr (METHOD == OLAP_LAST) next = v;
      else next = cur;
      cur = set ? (has ? next : v) : cur;
      has = has || set;
    }
  }
  // the pick (meaningful when `has`)
  __device__ __forceinline__ T value() const {
    if constexpr (kHw) return nan ? (T)__builtin_nan("") : cur;
    else return cur;
  }
  // folds in the pick of ANOTHER lane over members that come AFTER this lane's (highest / lowest: order-free; first keeps
  // its own when it has one, last takes the other's when that has one)
  __device__ __forceinline__ void merge(const Pick &q) {
    if constexpr (kHw) {
      cur = METHOD == OLAP_HIGHEST ? hw_max(cur, q.cur) : hw_min(cur, q.cur);  // (the identity on a side without members)
      nan = nan || q.nan;
    } else {
      T both;
      if constexpr (METHOD == OLAP_HIGHEST) both = select_max<T>(cur, q.cur);
      else if constexpr (METHOD == OLAP_LOWEST) both = select_min<T>(cur, q.cur);
      else if constexpr (METHOD == OLAP_LAST) both = q.cur;
      else both = cur;
      cur = has ? (q.has ? both : cur) : q.cur;
    }
    has = has || q.has;
  }
  // this lane's pick as held `delta` lanes further on (wave shuffle)
  __device__ __forceinline__ Pick shfl_down(uint32_t delta) const {
    Pick q;
    q.cur = __shfl_down(cur, delta, 64);
    q.has = __shfl_down((int)has, delta, 64) != 0;
    q.nan = __shfl_down((int)nan, delta, 64) != 0;
    return q;
  }
};

template <int METHOD> struct IsPick { static constexpr bool value = (METHOD == OLAP_HIGHEST || METHOD == OLAP_LOWEST || METHOD == OLAP_FIRST || METHOD == OLAP_LAST); };

// Writes one output cell: typed conversion, then the store invariant (set => value != default).
template <typename T>
__device__ __forceinline__ void emit_cell(double acc, bool has, bool def_nan, T &value, int32_t &status) {
  T tv = has ? Cell<T>::from_f64(acc) : Cell<T>::default_value(def_nan);
  bool set = has && !Cell<T>::is_default(tv, def_nan);
  value = set ? tv : Cell<T>::default_value(def_nan);
  status = set ? OLAP_STATUS_SET : 0;
}

// OLAP_PARTIAL_AVERAGE (the shard-local half of a sharded sum / average, olap_sharded.hip) hands out the float64
// ACCUMULATOR itself, whatever the cell type, and the number of contributions in the status slot: the reference adds
// every contribution of an output cell in float64 and never rounds in between (in-memory.js:282-290, :311-318), so a
// partial rounded to Float32 before the ranks are added loses that (2^24 + 1 - 2^24 = 0 instead of 1).  A cell nobody
// contributed to ships 0, the neutral element — never the NaN default.
template <int METHOD> struct IsPartial { static constexpr bool value = (METHOD == OLAP_PARTIAL_AVERAGE); };
template <typename T, int METHOD> struct OutCell { typedef T type; };
template <typename T> struct OutCell<T, OLAP_PARTIAL_AVERAGE> { typedef double type; };

// One output cell of a kernel instantiated for METHOD: the typed cell and its mask, or the float64 partial and its count.
template <typename T, int METHOD>
__device__ __forceinline__ void emit_out(double acc, bool has, uint32_t count, bool def_nan, typename OutCell<T, METHOD>::type &value,
                                         int32_t &status) {
  if constexpr (IsPartial<METHOD>::value) {
    value = has ? acc : 0.0;
    status = (int32_t)count;
  } else {
    emit_cell<T>(acc, has, def_nan, value, status);
  }
}

template <typename T>
__device__ __forceinline__ bool cell_is_set(T v, int32_t st, bool has_status, bool def_nan) {
  return (!has_status || (st & OLAP_STATUS_SET)) && !Cell<T>::is_default(v, def_nan);
}

// One output cell from partial payloads added up (over the ranks of a sharded cube, olap_sharded.hip; over the segments
// of a long group, segments_combine_kernel).  FINISH_ROUND / _AVERAGE: float64 partial sums (+ contribution
// counts) -> typed cell, rounded ONCE — what the one-device kernels do with their accumulator: Agg::finish
// (in-memory.js:323-331: divide by the Uint16 counter unless it wrapped to 0), then emit_cell.  Without counts (a sum
// over a 0 default) "somebody contributed" does not matter: set <=> the rounded sum is not 0.
template <typename T, typename P>
__device__ __forceinline__ void finish_cell(int finish, P a, uint32_t b, bool has_b, bool def_nan, T &ov, int32_t &os) {
  if (finish == OLAP_FINISH_ROUND || finish == OLAP_FINISH_AVERAGE) {
    double r = (double)a;
    const uint32_t c = has_b ? b : 1u;
    bool has = c != 0 && !is_default_f64(r, def_nan);
    if (finish == OLAP_FINISH_AVERAGE) {
      const uint32_t c16 = c & 0xFFFFu;  // Uint16Array counter
      if (c16) {
        r = (has ? r : (def_nan ? __builtin_nan("") : 0.0)) / (double)c16;
        has = !is_default_f64(r, def_nan);
      }
    }
    emit_cell<T>(r, has, def_nan, ov, os);
  } else {  // FINISH_NONE / FINISH_RESTORE: the payload is the typed cell (b: the OR of the masks)
    const T v = (T)a;
    const bool set = (!has_b || (b & OLAP_STATUS_SET) != 0) && !Cell<T>::is_default(v, def_nan);
    ov = set ? v : Cell<T>::default_value(def_nan);
    os = set ? OLAP_STATUS_SET : 0;
  }
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and each XCD has its own
// L2.  Mapping b -> logical id so that every XCD owns one CONTIGUOUS range of logical ids keeps
// neighbouring tiles (which share the 128-byte lines at their edges whenever a row is not line-aligned)
// inside one L2, where partial-line writes merge instead of reaching HBM twice.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
  constexpr uint32_t kXcd = 8;
  const uint32_t q = n / kXcd, r = n % kXcd;
  const uint32_t x = b % kXcd, i = b / kXcd;
  return x * q + (x < r ? x : r) + i;
}


// mulberry32 at stream position n (1-based draw index): state = seed + n * 0x6D2B79F5
__device__ __host__ __forceinline__ double mulberry32_at(uint32_t seed, uint64_t n) {
  uint32_t a = seed + (uint32_t)(n * 0x6D2B79F5ull);
  uint32_t t = (a ^ (a >> 15)) * (1u | a);
  t = (t + ((t ^ (t >> 7)) * (61u | t))) ^ t;
  return (double)(t ^ (t >> 14)) / 4294967296.0;
}

}  // namespace olap
