// olap_kernels.hpp — gfx950 kernels of the cube aggregation path and their launchers.
//
// Every kernel is bound by HBM bandwidth (one add per 4 B read), so the rules that matter are
// coalesced 16 B/lane accesses, enough loads in flight per lane, >> 256 workgroups per launch,
// and reading each input cell exactly once.  No MFMA: there is no contraction here.
//
// Instantiated once per cell type in olap_kernels_{f32,f64,i32,u32}.hip.
#pragma once

#include <cstdlib>

#include <type_traits>
#include <vector>

#include "olap_device.hpp"

namespace olap {

constexpr int kMaxDims = 12;  // collapsed dimensions a plan may keep (see plan.cpp)

// ======================================================================= K1: drillUp, one axis
// View [outer, K, inner] -> [outer, G, inner] (in-memory.js:265-334 with one non-identity map,
// which is all Cube.drillUp ever asks for, src/cube.js:999-1000).  The K->G map arrives as a
// CSR: gstart[g]..gstart[g+1] indexes `order`, the old indices of group g in ascending order
// (order == nullptr: groups are the contiguous runs themselves, e.g. calendars and 'all').
struct DrillUpAxis {
  uint64_t outer, K, inner, G;
  uint64_t n_vec;         // inner / VEC
  uint64_t total;         // outer * G * n_vec  (threads needed, flat regime)
  uint64_t blocks_per_row;// ceil(n_vec / kBlock)     (row regime)
  const uint32_t *order;  // device
  const uint32_t *gstart; // device
  int def_nan;
  int aligned16;          // every buffer of this launch is 16 B aligned
  int xcd_order;          // row regime: walk workgroups in XCD-contiguous order
  const uint32_t *gtile;  // device, or nullptr: group-tile regime, first group of every tile [n_gtile + 1]
  uint32_t n_gtile;
  const uint32_t *perm_cell;  // device, or nullptr: row-tile regime with interleaved groups (TilePerm), [K * inner + 3]
  const uint32_t *perm_grp;   // device, [2 G]
  uint32_t perm_pitch;        // members between two rows of the permuted tile
  uint32_t min_group;         // members of the smallest group (the cooperative forms re-associate sums only for groups of >= 256)
  uint32_t lanes;             // row regime: lanes per workgroup of this launch (256, 128 or 64; 0 = 256).  In the kernel
                              // arguments because blockDim.x is a VECTOR load from the dispatch's implicit arguments that
                              // every workgroup would wait for before it can compute its first address
  uint32_t grid;              // row regime: workgroups of this launch along x (0 = read gridDim.x): same reason
  uint32_t depth;             // row regime: 0 = the launcher decides; 4 = four rows in flight per lane whatever the row width
                              // (few, long workgroups: the segmented form)
};

// Several measures of a cube in ONE launch: the same drillUp (cell type, default, rule, shape, map) over up to
// kMaxBatch independent (input, output) buffer pairs; blockIdx.y picks the pair.  Cube.drillUp calls the store once
// per measure (src/cube.js:1012-1020); on cubes of a few MB a launch costs more than the bytes it moves, so the
// measures that share a rule share a launch (olap_plan_run_batch).  A single measure is a batch of one.
constexpr int kMaxBatch = 8;
template <typename T>
struct Batch {
  const T *in[kMaxBatch];
  const int32_t *st_in[kMaxBatch];
  T *out[kMaxBatch];
  int32_t *st_out[kMaxBatch];
  int32_t method[kMaxBatch];  // only read by the mixed-rule kernel (drillup_rows_mixed_kernel)
  static Batch one(const T *in, const int32_t *st_in, T *out, int32_t *st_out) {
    Batch b{};
    b.in[0] = in;
    b.st_in[0] = st_in;
    b.out[0] = out;
    b.st_out[0] = st_out;
    return b;
  }
};

// Per-lane accumulator of VEC adjacent output cells.  Additive/product methods run Agg<> in
// float64 (FAST = plain running sum, see below); highest/lowest/first/last run Pick<> in the cell
// type.
template <typename T, int METHOD, bool HAS_STATUS, int VEC, bool FAST>
struct Lane {
  static constexpr bool kPick = IsPick<METHOD>::value;
  Agg<METHOD> agg[kPick ? 1 : VEC];
  Pick<T, METHOD> pick[kPick ? VEC : 1];

  __device__ __forceinline__ void init() {
    if constexpr (kPick) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) pick[e].init();
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) agg[e].init();
    }
  }

  __device__ __forceinline__ void add_row(const Vec<T, VEC> &v, const Vec<int32_t, VEC> &s, bool def_nan) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T x = v.v[e];
      if constexpr (FAST) {
        // sum/average over a zero default without a mask: an unset cell holds 0 and adding it
        // changes nothing; the reference's "restart when the running sum hits 0" is invisible
        // for addition, so the plain float64 running sum is exact.
        agg[e].acc += Cell<T>::to_f64(x);
        if constexpr (METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE) agg[e].count += Cell<T>::is_default(x, false) ? 0u : 1u;
      } else {
        const int32_t sx = HAS_STATUS ? s.v[e] : OLAP_STATUS_SET;
        if constexpr (kPick) {
          pick[e].add_if(cell_is_set<T>(x, sx, HAS_STATUS, def_nan), x);
        } else if (cell_is_set<T>(x, sx, HAS_STATUS, def_nan)) {
          agg[e].add(Cell<T>::to_f64(x), def_nan);
        }
      }
    }
  }

  // what this METHOD writes: the typed cell, or the float64 partial of OLAP_PARTIAL_AVERAGE (olap_device.hpp: OutCell)
  typedef typename OutCell<T, METHOD>::type O;

  __device__ __forceinline__ void finish(bool def_nan, Vec<O, VEC> &ov, Vec<int32_t, VEC> &os) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if constexpr (kPick) {
        ov.v[e] = pick[e].has ? pick[e].value() : Cell<T>::default_value(def_nan);
        os.v[e] = pick[e].has ? OLAP_STATUS_SET : 0;
      } else {
        if constexpr (FAST) {
          // `sum`: the output is set iff the sum is not the default (0), which also covers
          // "nothing contributed"; `average` additionally needs one contribution
          agg[e].has = agg[e].acc != 0.0;
          if constexpr (METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE) agg[e].has = agg[e].has && agg[e].count != 0;
        }
        agg[e].finish(def_nan);
        emit_out<T, METHOD>(agg[e].acc, agg[e].has, agg[e].count, def_nan, ov.v[e], os.v[e]);
      }
    }
  }

  // `out` is the launch's value buffer: cells of T, or float64 partials under OLAP_PARTIAL_AVERAGE
  template <bool NT>
  __device__ __forceinline__ void finish_and_store(bool def_nan, T *out, int32_t *st_out, uint64_t oidx) {
    Vec<O, VEC> ov;
    Vec<int32_t, VEC> os;
    finish(def_nan, ov, os);
    O *dst = reinterpret_cast<O *>(out) + oidx;
    if constexpr (NT) {
      store_stream<O, VEC>(dst, ov);
      if (st_out) store_stream<int32_t, VEC>(st_out + oidx, os);
    } else {
      store_vec<O, VEC>(dst, ov);
      if (st_out) store_vec<int32_t, VEC>(st_out + oidx, os);
    }
  }
};

// Row regime (inner large): a workgroup owns kBlock adjacent VEC-wide slots of ONE (outer, group)
// pair, so the group bounds, the member list and every loop condition are wave-uniform (scalar
// loads and branches) and each wave-instruction reads VEC*sizeof(T)*64 contiguous bytes.  A lane
// walks its group's rows in ascending order — the reference's accumulation order, so the float64
// running sums are bit-identical — with U independent row loads in flight.
// RAGGED (rows that are not whole 16-byte groups — odd extents): slots are still VEC = 16 bytes wide, counted from
// the row's first cell; a slot is loaded and stored with ONE access at a cell-aligned address (free on the load
// side, +7 % on stores: tools/unaligned_probe.hip) and only the row's last, partial slot goes cell by cell —
// instead of 4-byte lanes (four times the load instructions and waves for the same bytes).
template <typename T, int METHOD, bool HAS_STATUS, int VEC, int U, bool CONTIG, bool FAST, bool NT = true, bool RAGGED = false>
__device__ __forceinline__ void drillup_rows_body(const T *__restrict__ in, const int32_t *__restrict__ st_in, T *__restrict__ out,
                                                  int32_t *__restrict__ st_out, const DrillUpAxis &a) {
  // blocks_per_row = ceil(n_vec / kBlock); blockIdx.x = og * blocks_per_row + chunk  (uniform math)
  const uint32_t bpr = (uint32_t)a.blocks_per_row;
  const uint32_t bid = a.xcd_order ? xcd_contiguous(blockIdx.x, a.grid ? a.grid : gridDim.x) : blockIdx.x;
  const uint64_t og = bid / bpr;
  const uint32_t chunk = bid - (uint32_t)og * bpr;
  const uint64_t g = og % a.G;
  const uint64_t o = og / a.G;
  const uint64_t iv = (uint64_t)chunk * (a.lanes ? a.lanes : kBlock) + threadIdx.x;  // workgroups of 256, 128 or 64 lanes (launcher)
  if (iv >= a.n_vec) return;
  const uint64_t i0 = iv * VEC;
  const bool def_nan = a.def_nan != 0;

  const T *base = in + (o * a.K) * a.inner + i0;
  const int32_t *sbase = HAS_STATUS ? st_in + (o * a.K) * a.inner + i0 : nullptr;
  // ('-> all' over a contiguous member list: the bounds are 0 and K — no table to wait for before the first row's address)
  const bool whole = CONTIG && a.G == 1;
  uint32_t j = 0u, jend = (uint32_t)a.K;
  if constexpr (CONTIG) {
    if (!whole) {
      // (workgroup-uniform; written as the scalar load it is — behind the test above the compiler issues two VECTOR
      // loads here, several times the latency, on every workgroup's way to its first row)
      const uint32_t *gp = a.gstart + __builtin_amdgcn_readfirstlane((uint32_t)g);
      uint64_t bounds;
      asm("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bounds) : "s"(gp));
      j = (uint32_t)bounds;
      jend = (uint32_t)(bounds >> 32);
    }
  } else {
    // (interleaved groups keep the compiler's own scalar loads: behind an asm statement it no longer trusts that nothing
    // was written and reads the member list in the loop with VECTOR loads — a dependent round trip in front of every
    // batch of rows, 6-8 % on [3652,100,274] location -> 10 interleaved groups)
    j = a.gstart[g];
    jend = a.gstart[g + 1];
  }

  Lane<T, METHOD, HAS_STATUS, VEC, FAST> lane;
  lane.init();

  // RAGGED: cells of this slot inside the row.  A row's last slot may be partial; it is loaded as the row's LAST
  // whole group (`shift` cells earlier) and rotated into place, so that no lane reads past its row and the loop
  // has no divergent branch around its loads (a divergent cell-by-cell tail made the compiler wait for every
  // row's load before issuing the next: 145 us against 95 us for the 4-byte lanes).  Only wavefronts that hold
  // such a slot run the rotating form of the loop.
  const uint32_t valid = RAGGED && i0 + VEC > a.inner ? (uint32_t)(a.inner - i0) : (uint32_t)VEC;
  const uint32_t shift = (uint32_t)VEC - valid;
  Vec<T, VEC> v[U];
  Vec<int32_t, VEC> s[U];
  auto accumulate = [&](auto rotating) {
    constexpr bool ROT = decltype(rotating)::value;
    auto fetch = [&](uint64_t k, Vec<T, VEC> &vv, Vec<int32_t, VEC> &ss) {
      const T *p = base + k * a.inner;
      const int32_t *sp = HAS_STATUS ? sbase + k * a.inner : nullptr;
      if constexpr (!RAGGED) {
        vv = NT ? load_stream<T, VEC>(p) : load_vec<T, VEC>(p);
        if constexpr (HAS_STATUS) ss = NT ? load_stream<int32_t, VEC>(sp) : load_vec<int32_t, VEC>(sp);
      } else if constexpr (!ROT) {
        vv = load_stream_cell_aligned<T, VEC>(p);
        if constexpr (HAS_STATUS) ss = load_stream_cell_aligned<int32_t, VEC>(sp);
      } else {
        const Vec<T, VEC> lv = load_stream_cell_aligned<T, VEC>(p - shift);
        Vec<int32_t, VEC> ls;
        if constexpr (HAS_STATUS) ls = load_stream_cell_aligned<int32_t, VEC>(sp - shift);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          T x = Cell<T>::default_value(def_nan);
          int32_t sx = 0;
#pragma unroll
          for (int sh = 0; sh + e < VEC; ++sh) {
            x = shift == (uint32_t)sh ? lv.v[e + sh] : x;
            if constexpr (HAS_STATUS) sx = shift == (uint32_t)sh ? ls.v[e + sh] : sx;
          }
          vv.v[e] = x;
          if constexpr (HAS_STATUS) ss.v[e] = sx;
        }
      }
    };
    if constexpr (METHOD == OLAP_FIRST || METHOD == OLAP_LAST) {
      // `first` / `last` need the first / last SET member only: walk towards it — ascending for first, descending for
      // last — and stop as soon as every cell of the wavefront has found one (a wave-uniform test; the workgroup's
      // other wavefronts go on by themselves).  On a dense cube that is ONE row of the group instead of all K: the
      // roll-up reads 1/K of the cube.  Rows are requested one ahead of the test, then four at a time, so that a sparse
      // group does not pay one memory round trip per row.
      constexpr bool DESC = METHOD == OLAP_LAST;
      const uint32_t n = jend - j;
      auto member = [&](uint32_t t) -> uint64_t {
        const uint32_t jj = DESC ? jend - 1 - t : j + t;
        return CONTIG ? (uint64_t)jj : (uint64_t)a.order[jj];
      };
      auto take = [&](const Vec<T, VEC> &vv, const Vec<int32_t, VEC> &ss) {
        bool all = true;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const bool set = cell_is_set<T>(vv.v[e], HAS_STATUS ? ss.v[e] : OLAP_STATUS_SET, HAS_STATUS, def_nan);
          lane.pick[e].cur = (!lane.pick[e].has && set) ? vv.v[e] : lane.pick[e].cur;  // the first one found on the way stays
          lane.pick[e].has = lane.pick[e].has || set;
          all = all && (lane.pick[e].has || (RAGGED && (uint32_t)e >= valid));
        }
        return all;
      };
      uint32_t t = 0;
      bool done = n == 0;
      if (!done) {
        fetch(member(0), v[0], s[0]);
        done = __all(take(v[0], s[0]));
        t = 1;
      }
      while (!done && t < n) {
        constexpr int B = 4;
        Vec<T, VEC> bv[B];
        Vec<int32_t, VEC> bs[B];
        const uint32_t m = n - t < (uint32_t)B ? n - t : (uint32_t)B;  // wave-uniform
#pragma unroll
        for (int u = 0; u < B; ++u)
          if ((uint32_t)u < m) fetch(member(t + u), bv[u], bs[u]);
        bool all = false;
#pragma unroll
        for (int u = 0; u < B; ++u)
          if ((uint32_t)u < m) all = take(bv[u], bs[u]);
        done = __all(all);
        t += m;
      }
      j = jend;
      return;
    }
    // (not for values + mask — two streams per row already: 149 -> 151 us with it — and not for the plain sums, whose
    // fold is eight instructions: no difference beyond the run-to-run spread in a same-box A/B)
    constexpr bool kPrefetch = U == 1 && !FAST && !HAS_STATUS;
    if constexpr (kPrefetch) {
      // one row in flight, but the next row is requested BEFORE the current one is folded in: the exact state
      // machine / the NaN-propagating picks are a dozen VALU instructions per cell, which otherwise sit between
      // one row's arrival and the next row's request on every lane's critical path
      if (j < jend) {
        fetch(CONTIG ? (uint64_t)j : (uint64_t)a.order[j], v[0], s[0]);
        for (++j; j < jend; ++j) {
          Vec<T, VEC> nv;
          Vec<int32_t, VEC> ns;
          fetch(CONTIG ? (uint64_t)j : (uint64_t)a.order[j], nv, ns);
          lane.add_row(v[0], s[0], def_nan);
          v[0] = nv;
          s[0] = ns;
        }
        lane.add_row(v[0], s[0], def_nan);
      }
    }
    for (; j + U <= jend; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) fetch(CONTIG ? (uint64_t)(j + u) : (uint64_t)a.order[j + u], v[u], s[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) lane.add_row(v[u], s[u], def_nan);
    }
    const uint32_t rest = jend - j;  // < U, wave-uniform
    if (rest) {
#pragma unroll
      for (int u = 0; u < U - 1; ++u)
        if ((uint32_t)u < rest) fetch(CONTIG ? (uint64_t)(j + u) : (uint64_t)a.order[j + u], v[u], s[u]);
#pragma unroll
      for (int u = 0; u < U - 1; ++u)
        if ((uint32_t)u < rest) lane.add_row(v[u], s[u], def_nan);
    }
  };
  if constexpr (RAGGED) {
    if (__any(shift != 0)) accumulate(std::true_type{});  // wave-uniform
    else accumulate(std::false_type{});
  } else {
    accumulate(std::false_type{});
  }
  const uint64_t oidx = (o * a.G + g) * a.inner + i0;
  if constexpr (!RAGGED) {
    lane.template finish_and_store<NT>(def_nan, out, st_out, oidx);
  } else {
    typedef typename OutCell<T, METHOD>::type O;
    Vec<O, VEC> ov;
    Vec<int32_t, VEC> os;
    lane.finish(def_nan, ov, os);
    O *dst = reinterpret_cast<O *>(out) + oidx;
    if (valid == (uint32_t)VEC) {
      store_stream_cell_aligned<O, VEC>(dst, ov);
      if (st_out) store_stream_cell_aligned<int32_t, VEC>(st_out + oidx, os);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if ((uint32_t)e < valid) {
          dst[e] = ov.v[e];
          if (st_out) st_out[oidx + e] = os.v[e];
        }
    }
  }
}

template <typename T, int METHOD, bool HAS_STATUS, int VEC, int U, bool CONTIG, bool FAST, bool NT = true, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void drillup_rows_kernel(const Batch<T> b, const DrillUpAxis a) {
  // Everything that comes from the kernel arguments is requested HERE, in one batch of scalar loads: left to itself the
  // compiler requests each pointer where it is first used — behind the early exit, the table look-up, the loop — and
  // a workgroup then pays five dependent scalar round trips before its first row is requested (ISA of round 3's
  // build; a stripped loop with one round trip was 3 % faster on the headline, tools/headline_limit.hip).
  // (asm statements WITHOUT side effects and without a memory clobber: a volatile one counts as a possible write, after
  // which the compiler reads the bounds and the member list with vector loads — a dependent round trip per batch of
  // rows.  And the pointers themselves must not pass through one: what comes out is a generic pointer — flat loads.
  // Their bits are tied into a value the first address needs instead: `pin >> 63` is 0, which the compiler cannot know.)
  const T *in = b.in[blockIdx.y];
  const int32_t *st_in = b.st_in[blockIdx.y];
  T *out = b.out[blockIdx.y];
  int32_t *st_out = b.st_out[blockIdx.y];
  DrillUpAxis al = a;
  uint64_t pin = (uint64_t)in | (uint64_t)st_in | (uint64_t)out | (uint64_t)st_out | (uint64_t)al.gstart | (uint64_t)al.order;
  asm("" : "+s"(pin), "+s"(al.K), "+s"(al.inner), "+s"(al.G), "+s"(al.n_vec), "+s"(al.blocks_per_row), "+s"(al.grid), "+s"(al.lanes),
      "+s"(al.xcd_order), "+s"(al.def_nan));
  al.blocks_per_row += pin >> 63;
  drillup_rows_body<T, METHOD, HAS_STATUS, VEC, U, CONTIG, FAST, NT, RAGGED>(in, st_in, out, st_out, al);
}

// Measures with DIFFERENT rules in one launch (config 5: sum / average / first / last over the same roll-up): the rule
// of pair blockIdx.y is a workgroup-uniform switch in front of the same bodies.  The kernel carries every rule's code
// and the registers of the greediest, so it only serves batches whose rules differ; one rule -> drillup_rows_kernel.
template <typename T, bool HAS_STATUS, int VEC, int U, bool CONTIG>
__global__ __launch_bounds__(kBlock) void drillup_rows_mixed_kernel(const Batch<T> b, const DrillUpAxis a_) {
  // (kernel arguments in one batch of scalar loads: see drillup_rows_kernel)
  const T *in = b.in[blockIdx.y];
  const int32_t *st_in = b.st_in[blockIdx.y];
  T *out = b.out[blockIdx.y];
  int32_t *st_out = b.st_out[blockIdx.y];
  int32_t rule = b.method[blockIdx.y];
  DrillUpAxis a = a_;
  uint64_t pin = (uint64_t)in | (uint64_t)st_in | (uint64_t)out | (uint64_t)st_out | (uint64_t)a.gstart | (uint64_t)a.order;
  asm("" : "+s"(pin), "+s"(rule), "+s"(a.K), "+s"(a.inner), "+s"(a.G), "+s"(a.n_vec), "+s"(a.blocks_per_row), "+s"(a.grid), "+s"(a.lanes),
      "+s"(a.xcd_order), "+s"(a.def_nan));
  a.blocks_per_row += pin >> 63;
  const bool fast = !HAS_STATUS && !a.def_nan;  // additive rules over a 0 default without a mask: plain running sums
  switch (rule) {
    case OLAP_SUM:
      if (fast) drillup_rows_body<T, OLAP_SUM, HAS_STATUS, VEC, U, CONTIG, !HAS_STATUS>(in, st_in, out, st_out, a);
      else drillup_rows_body<T, OLAP_SUM, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a);
      break;
    case OLAP_AVERAGE:
      if (fast) drillup_rows_body<T, OLAP_AVERAGE, HAS_STATUS, VEC, U, CONTIG, !HAS_STATUS>(in, st_in, out, st_out, a);
      else drillup_rows_body<T, OLAP_AVERAGE, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a);
      break;
    case OLAP_HIGHEST: drillup_rows_body<T, OLAP_HIGHEST, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a); break;
    case OLAP_LOWEST: drillup_rows_body<T, OLAP_LOWEST, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a); break;
    case OLAP_FIRST: drillup_rows_body<T, OLAP_FIRST, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a); break;
    case OLAP_LAST: drillup_rows_body<T, OLAP_LAST, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a); break;
    default: drillup_rows_body<T, OLAP_PRODUCT, HAS_STATUS, VEC, U, CONTIG, false>(in, st_in, out, st_out, a); break;
  }
}


// Flat regime (inner small): one lane per VEC output cells, (outer, group) decoded per lane.
// IDX: the lane index is decoded in 32-bit arithmetic when the launch has < 2^32 lanes (a 64-bit
// division is a long software sequence on CDNA).
template <typename T, int METHOD, bool HAS_STATUS, int VEC, bool FAST, typename IDX>
__global__ __launch_bounds__(kBlock) void drillup_flat_kernel(const Batch<T> b, const DrillUpAxis a) {
  const T *__restrict__ in = b.in[blockIdx.y];
  const int32_t *__restrict__ st_in = b.st_in[blockIdx.y];
  T *__restrict__ out = b.out[blockIdx.y];
  int32_t *__restrict__ st_out = b.st_out[blockIdx.y];
  const IDX t = (IDX)blockIdx.x * kBlock + threadIdx.x;
  if ((uint64_t)t >= a.total) return;
  const IDX nv = (IDX)a.n_vec, ng = (IDX)a.G;
  const uint64_t iv = t % nv;
  const IDX og = t / nv;
  const uint64_t g = og % ng;
  const uint64_t o = og / ng;
  const uint64_t i0 = iv * VEC;
  const bool def_nan = a.def_nan != 0;

  const T *base = in + (o * a.K) * a.inner + i0;
  const int32_t *sbase = HAS_STATUS ? st_in + (o * a.K) * a.inner + i0 : nullptr;
  uint32_t j = a.gstart[g];
  const uint32_t jend = a.gstart[g + 1];

  Lane<T, METHOD, HAS_STATUS, VEC, FAST> lane;
  lane.init();
  constexpr int U = 4;
  // software pipeline: a batch of U members (their list entries, then their row pieces: two dependent loads) is
  // requested while the previous batch is folded in — lanes of a wavefront walk different groups, so nothing else
  // overlaps one lane's load latency with its own arithmetic
  auto fetch = [&](uint32_t j0, Vec<T, VEC> *v, Vec<int32_t, VEC> *s) {
    const uint32_t n = (jend - j0) < (uint32_t)U ? (jend - j0) : (uint32_t)U;
    uint64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t jj = (uint32_t)u < n ? j0 + u : j0;  // clamp: re-reads a valid row, result unused
      k[u] = a.order ? (uint64_t)a.order[jj] : (uint64_t)jj;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // (cached loads: with short row pieces neighbouring outputs share lines — streaming loads
      // took [3001,3333,10] from 89 to 159 us.  A 32-bit `member * inner` where rows are short enough, instead of
      // the 64-bit vector multiply, was measured and is slower here: 75.5 -> 81 us, 85.5 -> 94.4 us.)
      v[u] = load_vec<T, VEC>(base + k[u] * a.inner);
      if constexpr (HAS_STATUS) s[u] = load_vec<int32_t, VEC>(sbase + k[u] * a.inner);
    }
    return n;
  };
  if (j < jend) {
    Vec<T, VEC> v[U], w[U];
    Vec<int32_t, VEC> s[U], ws[U];
    uint32_t n = fetch(j, v, s);
    for (j += U; j < jend; j += U) {
      const uint32_t nn = fetch(j, w, ws);
#pragma unroll
      for (int u = 0; u < U; ++u) lane.add_row(v[u], s[u], def_nan);  // (only a group's last batch is partial)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = w[u];
        s[u] = ws[u];
      }
      n = nn;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((uint32_t)u < n) lane.add_row(v[u], s[u], def_nan);
  }
  lane.template finish_and_store<false>(def_nan, out, st_out, (o * a.G + g) * a.inner + i0);
}

struct SmallDiv {
  uint32_t d, magic;
  int how;  // 0: mul-hi by magic; 1: x < 2 d, one compare; 2: d == 1
};
__device__ __forceinline__ uint32_t small_div(uint32_t x, const SmallDiv &s) {
  return s.how == 0 ? __umulhi(x, s.magic) : s.how == 1 ? (x >= s.d ? 1u : 0u) : x;
}
// exact for every x <= x_max, or false (m = floor(2^32 / d) + 1 is exact while x d < 2^32)
inline bool small_div_for(uint64_t d, uint64_t x_max, SmallDiv *out) {
  if (d == 0 || d > 0xFFFFFFFFull || x_max > 0xFFFFFFFFull) return false;
  out->d = (uint32_t)d;
  out->magic = 0;
  if (d == 1) {
    out->how = 2;
    return x_max <= 0xFFFFFFFFull;
  }
  if (x_max < 2 * d) {
    out->how = 1;
    return true;
  }
  if (x_max * d >= (1ull << 32)) return false;
  out->how = 0;
  out->magic = (uint32_t)((1ull << 32) / d + 1);
  return true;
}

// Tile regime (inner small, K*inner fits LDS): the LDS-staged segmented reduction.  A row of the
// view is K*inner CONTIGUOUS cells, so a workgroup stages R whole rows with coalesced 16 B streaming
// loads, then every lane reduces output cells (row, group, i) out of LDS walking the group's
// members in ascending order (same float64 order as the other regimes) and the R*G*inner results
// leave as one contiguous, coalesced store.  HBM sees each input cell once, in full lines.
struct DrillUpTile {
  uint32_t rows_per_tile;   // R (multiple of 4 so that every tile starts 16 B aligned)
  uint32_t row_elems;       // K * inner
  uint32_t out_row;         // G * inner
  uint32_t inner;
  // MODE 3 (cells permuted into group order on their way into LDS; tables built by the plan, tile_perm_build)
  SmallDiv by_row;              // row of a staged cell without a hardware division
  const uint32_t *perm_cell;    // device, [row_elems + V]: LDS position of every cell of a row (+ the next row's first cells)
  const uint32_t *perm_grp;     // device, [2 G]: first and one-past-last position (in members) of every group's run
  uint32_t pitch_cells;         // LDS cells between two rows of the tile (>= row_elems: runs start on chosen banks)
};

// Host side of MODE 3: where the members of a row go in the permuted tile.  Group g's members become the run
// [start[g], start[g] + size[g]) (in members; a member is `inner` cells), rows are `pitch` members apart.  The
// reduction's lanes are (row, group, i) in that order and read member j of their runs in one instruction, so the
// runs' first cells should fall on distinct LDS banks (32 banks of 4 bytes): start[g] = c g and pitch = c G modulo
// 32 / gcd(inner, 32) puts lane x of 32 consecutive lanes on bank c x (inner = 1, c odd: a permutation of the banks)
// or on bank x (inner > 1, c = 1).  c is the odd multiplier that pads least — equal runs of odd length need no
// padding at all, of even length one cell each.  When the padding would cost more than a quarter of the rows a tile
// holds, runs are packed (start = gstart) and the bank conflicts are accepted.
struct TilePerm {
  std::vector<uint32_t> cell;  // [K * inner + 3]: LDS cell of every cell of a row; entries past the row run into the next row(s)
  std::vector<uint32_t> grp;   // [2 G]: start[g], start[g] + size[g]
  uint32_t pitch = 0;
};
inline uint64_t tile_rows_for(uint64_t row_elems, uint64_t pitch_cells, uint64_t budget_cells, uint64_t V) {
  uint64_t R = pitch_cells ? budget_cells / pitch_cells : 0;
  if (R >= 4) R &= ~3ull;
  while (R > 0 && (R * row_elems) % V != 0) --R;  // every tile must start 16 B aligned
  return R;
}
inline void tile_perm_build(const uint32_t *gstart, const uint32_t *order, uint32_t K, uint32_t G, uint32_t inner,
                            uint64_t budget_cells, uint64_t V, TilePerm *out) {
  uint32_t gcd = inner, b = 32;
  while (b) {
    const uint32_t t = gcd % b;
    gcd = b;
    b = t;
  }
  const uint32_t M = 32 / gcd;
  auto layout = [&](uint32_t c, std::vector<uint32_t> *start) {
    uint64_t s = 0;
    for (uint32_t g = 0; g < G; ++g) {
      s += ((uint64_t)c * g % M + M - s % M) % M;
      if (start) (*start)[g] = (uint32_t)s;
      s += gstart[g + 1] - gstart[g];
    }
    s += ((uint64_t)c * G % M + M - s % M) % M;
    return s;
  };
  uint32_t best_c = 0;
  uint64_t best = ~0ull;
  for (uint32_t c = 1; c < 32; c += 2) {
    const uint64_t pitch = layout(c, nullptr);
    if (pitch < best) {
      best = pitch;
      best_c = c;
    }
    if (inner != 1) break;
  }
  const uint64_t row_elems = (uint64_t)K * inner;
  std::vector<uint32_t> start(G);
  const uint64_t rows_packed = tile_rows_for(row_elems, row_elems, budget_cells, V);
  if (M > 1 && tile_rows_for(row_elems, best * inner, budget_cells, V) * 4 >= rows_packed * 3 && best * inner <= budget_cells) {
    out->pitch = (uint32_t)layout(best_c, &start);
  } else {
    for (uint32_t g = 0; g < G; ++g) start[g] = gstart[g];
    out->pitch = K;
  }
  out->grp.resize(2 * (size_t)G);
  out->cell.assign(row_elems + 3, 0);
  for (uint32_t g = 0; g < G; ++g) {
    out->grp[2 * g] = start[g];
    out->grp[2 * g + 1] = start[g] + (gstart[g + 1] - gstart[g]);
    for (uint32_t j = gstart[g]; j < gstart[g + 1]; ++j)
      for (uint32_t i = 0; i < inner; ++i) out->cell[(uint64_t)order[j] * inner + i] = (start[g] + (j - gstart[g])) * inner + i;
  }
  for (uint64_t idx = row_elems; idx < row_elems + 3; ++idx)
    out->cell[idx] = out->cell[idx % row_elems] + (uint32_t)(idx / row_elems) * out->pitch * inner;
}

// tools/tile_probe.hip builds this header with OLAP_TILE_PROBE: lane 0 of every workgroup of the tile kernel then
// records the 100 MHz wall clock at its phase boundaries (not compiled into the product)
#ifdef OLAP_TILE_PROBE
__device__ unsigned long long g_tile_probe[1 << 20];
#define OLAP_PROBE(i)                                                                         \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < (1u << 17)) g_tile_probe[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
__device__ __forceinline__ void g_probe3(uint32_t b) {
  if (b < (1u << 17)) g_tile_probe[b * 8 + 3] = wall_clock64();
}
#else
#define OLAP_PROBE(i) do { } while (0)
__device__ __forceinline__ void g_probe3(uint32_t) {}
#endif

constexpr uint32_t kTileBytes = 16 * 1024;  // cells staged per workgroup: 8 workgroups (32 waves) per CU

// Reduce regime (few output cells, long groups — [10^6,100] -> [1,100], [27400,3652] -> [27400,1],
// [10^8] -> [1]): the natural one-lane-per-output mappings above would leave the chip idle, so the
// member list of every group is cut into S segments and reduced cooperatively; partial states are
// merged afterwards.  Sums are float64 but RE-ASSOCIATED here (tree order instead of the
// reference's sequential order: ~1e-16 relative before the final rounding); highest / lowest are
// order-free; first / last carry the member position and merge by it, so they stay exact.
struct Partial {
  double acc;
  uint32_t meta;  // bit 31: set, bits 0..30: contributions
  uint32_t pos;   // member position (first / last)
};

template <int METHOD>
__device__ __forceinline__ Partial partial_identity() {
  Partial p;
  p.acc = 0.0;
  p.meta = 0;
  p.pos = 0;
  return p;
}

template <int METHOD>
__device__ __forceinline__ void partial_add(Partial &p, double v, uint32_t pos, bool def_nan) {
  const uint32_t cnt = (p.meta & 0x7FFFFFFFu) + 1;
  bool has = (p.meta & 0x80000000u) != 0;
  if (!has) {
    p.acc = v;
    p.pos = pos;
    has = true;
  } else {
    if constexpr (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE) p.acc += v;
    else if constexpr (METHOD == OLAP_PRODUCT) p.acc *= v;
    else if constexpr (METHOD == OLAP_HIGHEST) p.acc = js_max(p.acc, v);
    else if constexpr (METHOD == OLAP_LOWEST) p.acc = js_min(p.acc, v);
    else if constexpr (METHOD == OLAP_FIRST) { if (pos < p.pos) { p.acc = v; p.pos = pos; } }
    else { if (pos >= p.pos) { p.acc = v; p.pos = pos; } }
    if constexpr (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE || METHOD == OLAP_PRODUCT)
      if (is_default_f64(p.acc, def_nan)) has = false;  // the key is dropped (in-memory.js:126-131)
  }
  p.meta = (has ? 0x80000000u : 0u) | (cnt & 0x7FFFFFFFu);
}

template <int METHOD>
__device__ __forceinline__ void partial_merge(Partial &a, const Partial &b, bool def_nan) {
  const uint32_t cnt = (a.meta & 0x7FFFFFFFu) + (b.meta & 0x7FFFFFFFu);
  const bool ha = (a.meta & 0x80000000u) != 0, hb = (b.meta & 0x80000000u) != 0;
  bool has = ha || hb;
  if (ha && hb) {
    if constexpr (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE) a.acc += b.acc;
    else if constexpr (METHOD == OLAP_PRODUCT) a.acc *= b.acc;
    else if constexpr (METHOD == OLAP_HIGHEST) a.acc = js_max(a.acc, b.acc);
    else if constexpr (METHOD == OLAP_LOWEST) a.acc = js_min(a.acc, b.acc);
    else if constexpr (METHOD == OLAP_FIRST) { if (b.pos < a.pos) { a.acc = b.acc; a.pos = b.pos; } }
    else { if (b.pos >= a.pos) { a.acc = b.acc; a.pos = b.pos; } }
    if constexpr (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE || METHOD == OLAP_PRODUCT)
      if (is_default_f64(a.acc, def_nan)) has = false;
  } else if (hb) {
    a.acc = b.acc;
    a.pos = b.pos;
  }
  a.meta = (has ? 0x80000000u : 0u) | (cnt & 0x7FFFFFFFu);
}

struct DrillUpReduce {
  uint32_t S;         // segments per group
  uint32_t seg_len;   // members per segment
  uint32_t rows;      // member rows a unit reads per step: power of two, rows * inner <= unit
  uint32_t unit;      // lanes cooperating on one (outer, group, segment): 64 (a wavefront) or 256
  uint32_t vec4;      // 1: the 16 B form (drillup_reduce4_kernel) applies; `rows` is then sized for unit*4 cells
  uint32_t edge;      // 16 B form with inner == 1 and rows at ANY cell offset (K % 4 != 0): aligned groups, masked ends
  Partial *part;      // [outer*G*inner * S]
};

// Partial state -> output cell (what drillup_merge_kernel does after merging the segments).
template <typename T, int METHOD>
__device__ __forceinline__ void partial_finish(const Partial &p, bool def_nan, typename OutCell<T, METHOD>::type &ov, int32_t &os) {
  Agg<METHOD> agg;
  agg.acc = p.acc;
  agg.has = (p.meta & 0x80000000u) != 0;
  agg.count = p.meta & 0x7FFFFFFFu;
  agg.finish(def_nan);
  emit_out<T, METHOD>(agg.acc, agg.has, agg.count, def_nan, ov, os);
}

// Lane-to-lane merge step of the reduce regime.  FAST (sum / average over a 0 default, no mask): the
// state is (running sum, count of non-zero contributions) and merging is two adds — the `set` bit
// is derived once at the end (partial_seal).  A wave64 VALU instruction costs 4 cycles and a unit
// runs 6-8 of these levels per element, so the generic partial_merge (~40 instructions) is kept
// for the methods that need it.
template <int METHOD, bool FAST>
__device__ __forceinline__ void partial_merge_lane(Partial &a, uint32_t delta, bool def_nan) {
  Partial q;
  q.acc = __shfl_down(a.acc, delta, 64);
  q.meta = __shfl_down(a.meta, delta, 64);
  if constexpr (METHOD == OLAP_FIRST || METHOD == OLAP_LAST) q.pos = __shfl_down(a.pos, delta, 64);
  else q.pos = 0;
  if constexpr (FAST) {
    a.acc += q.acc;
    a.meta += q.meta;
  } else {
    partial_merge<METHOD>(a, q, def_nan);
  }
}

template <int METHOD, bool FAST>
__device__ __forceinline__ void partial_merge_fast(Partial &a, const Partial &q, bool def_nan) {
  if constexpr (FAST) {
    a.acc += q.acc;
    a.meta += q.meta;
  } else {
    partial_merge<METHOD>(a, q, def_nan);
  }
}

// FAST accumulators count contributions without the `set` bit; this adds it.
__device__ __forceinline__ void partial_seal(Partial &p) {
  p.meta = (p.meta & 0x7FFFFFFFu) | ((p.meta != 0 && p.acc != 0.0) ? 0x80000000u : 0u);
}

__device__ __forceinline__ Partial partial_shfl_down(const Partial &p, uint32_t delta) {
  Partial q;
  q.acc = __shfl_down(p.acc, delta, 64);
  q.meta = __shfl_down(p.meta, delta, 64);
  q.pos = __shfl_down(p.pos, delta, 64);
  return q;
}


// (Measured and dropped twice: a PERSISTENT form — one workgroup per resident slot of the chip walking a contiguous run
// of tiles, the next tile's cells requested into registers before the current tile is reduced, tables to LDS once per
// workgroup: [10]^8 axes 5-7 70 -> 84 us, [27400,3652] day -> month 61 -> 75 us.)
// MODE 1 (ALL): one group holding every member in order (the '-> all' roll-ups of slice /
// removeDimension / collapse): no table reads at all.  MODE 2: groups are contiguous member runs
// (calendars, attribute roll-ups of sorted items): only gstart[G+1] goes to LDS.  MODE 0: gstart and
// the member list are copied to LDS once per workgroup.  MODE 3 (interleaved groups, the default for them):
// the cells are PERMUTED into group order on their way into LDS — every group's members become one contiguous
// run — so that the reduction walks runs as MODE 2 does.  The reduction of MODE 0 reads a member index and then the
// cell, two dependent LDS round trips per member on the critical path of a lane that walks ~K/G members, and copies
// the member list into LDS behind the tile's loads; with few outputs per tile (4 rows x 10 groups of 100 members)
// a workgroup lived 6.6 us of which 3 us reducing (tools/tile_probe.hip).  The runs' first cells are also placed
// on distinct LDS banks (TilePerm below): lanes (row, group) read member j of their runs in the same instruction,
// and with runs 100 cells apart 40 lanes shared 8 banks.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST, int MODE>
__device__ __forceinline__ void drillup_tile_body(const T *__restrict__ in, const int32_t *__restrict__ st_in, T *__restrict__ out,
                                                  int32_t *__restrict__ st_out, const DrillUpAxis &a, const DrillUpTile &tl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int V = 16 / sizeof(T);                       // cells per 16 B access
  constexpr uint32_t kCells = kTileBytes / sizeof(T);     // capacity of the staged tile
  constexpr int NL = kCells / V / kBlock;                 // 16 B loads per lane (4)
  T *tile = reinterpret_cast<T *>(lds_raw);
  int32_t *stile = reinterpret_cast<int32_t *>(lds_raw + kTileBytes);
  uint32_t *csr = reinterpret_cast<uint32_t *>(lds_raw + kTileBytes + (HAS_STATUS ? kCells * 4 : 0));
  constexpr bool ALL = MODE == 1;
  constexpr bool PERMUTE = MODE == 3;
  uint32_t *l_gstart = csr;                 // G + 1 entries
  uint32_t *l_order = csr + a.G + 1;        // K entries (MODE 0: member list; MODE 3: rank of every member)

  const uint64_t row0 = (uint64_t)(a.xcd_order ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x) * tl.rows_per_tile;
  const uint32_t rows = (uint32_t)((a.outer - row0) < tl.rows_per_tile ? (a.outer - row0) : tl.rows_per_tile);
  const uint32_t n_in = rows * tl.row_elems;
  const T *src = in + row0 * tl.row_elems;
  const int32_t *ssrc = HAS_STATUS ? st_in + row0 * tl.row_elems : nullptr;
  const bool def_nan = a.def_nan != 0;

  OLAP_PROBE(0);
  // stage: all global loads of the lane first, then the LDS writes
  const uint32_t n_vec = n_in / V;
  Vec<T, V> v[NL];
  Vec<int32_t, V> sv[NL];
  // MODE 3: where the lane's cells go (tl.perm_cell, built by the plan: the LDS position of every cell of a row, plus V
  // entries that run into the next row) and the bounds of one group, requested BEFORE the cells — loads complete in
  // order, so a table load issued behind the tile's would be waited for with the whole tile in front of it.  The
  // table is a few KB read by every workgroup (L2 / L1 hits) with one 16-byte load at a 4-byte-aligned address per
  // 16 bytes of cells.
  typedef uint32_t PosVec __attribute__((ext_vector_type(V), aligned(4)));
  PosVec pos[NL];
  uint32_t grp0 = 0;
  if constexpr (PERMUTE) {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const uint32_t iv = threadIdx.x + u * kBlock;
      const uint32_t c0 = iv < n_vec ? iv * V : 0u;
      const uint32_t r = small_div(c0, tl.by_row);
      pos[u] = *reinterpret_cast<const PosVec *>(tl.perm_cell + (c0 - r * tl.row_elems)) + r * tl.pitch_cells;
    }
    grp0 = threadIdx.x < 2 * a.G ? tl.perm_grp[threadIdx.x] : 0u;
  }
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t i = threadIdx.x + u * kBlock;
    if (i < n_vec) {
      v[u] = load_stream<T, V>(src + (uint64_t)i * V);
      if constexpr (HAS_STATUS) sv[u] = load_stream<int32_t, V>(ssrc + (uint64_t)i * V);
    }
  }
  if constexpr (PERMUTE) {
    if (threadIdx.x < 2 * a.G) l_gstart[threadIdx.x] = grp0;
    for (uint32_t i = threadIdx.x + kBlock; i < 2 * a.G; i += kBlock) l_gstart[i] = tl.perm_grp[i];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const uint32_t iv = threadIdx.x + u * kBlock;
      if (iv < n_vec) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          tile[pos[u][e]] = v[u].v[e];
          if constexpr (HAS_STATUS) stile[pos[u][e]] = sv[u].v[e];
        }
      }
    }
    for (uint32_t c = n_vec * V + threadIdx.x; c < n_in; c += kBlock) {  // < V leftover cells
      const uint32_t r = small_div(c, tl.by_row);
      const uint32_t at = tl.perm_cell[c - r * tl.row_elems] + r * tl.pitch_cells;
      tile[at] = src[c];
      if constexpr (HAS_STATUS) stile[at] = ssrc[c];
    }
  } else {
    if constexpr (!ALL) {
      for (uint32_t i = threadIdx.x; i <= a.G; i += kBlock) l_gstart[i] = a.gstart[i];
      if constexpr (MODE == 0)
        for (uint32_t i = threadIdx.x; i < a.K; i += kBlock) l_order[i] = a.order[i];
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const uint32_t i = threadIdx.x + u * kBlock;
      if (i < n_vec) {
        *reinterpret_cast<Vec<T, V> *>(tile + i * V) = v[u];
        if constexpr (HAS_STATUS) *reinterpret_cast<Vec<int32_t, V> *>(stile + i * V) = sv[u];
      }
    }
    for (uint32_t i = n_vec * V + threadIdx.x; i < n_in; i += kBlock) {  // < V leftover cells
      tile[i] = src[i];
      if constexpr (HAS_STATUS) stile[i] = ssrc[i];
    }
  }
  __syncthreads();
  OLAP_PROBE(2);

  const uint32_t n_out = rows * tl.out_row;
  // (starting the reducing lanes at a wavefront that differs from workgroup to workgroup — with fewer outputs than
  // lanes only the first wavefront reduces — was measured and is slower: 69 -> 77 us on 4 rows x 10 runs of 100)
  const uint32_t tid = threadIdx.x;
  typedef typename OutCell<T, METHOD>::type O;  // float64 partials under OLAP_PARTIAL_AVERAGE
  O *dst = reinterpret_cast<O *>(out) + row0 * tl.out_row;
  int32_t *sdst = st_out ? st_out + row0 * tl.out_row : nullptr;
  if constexpr (ALL && FAST) {  // (FAST: sum / average over a 0 default without a mask)
    // one long row per output (rolling up a last dimension of 256..4096 items): a tile holds a dozen
    // rows, so a lane per row would leave the workgroup idle — 16 lanes share a row, each sums a
    // contiguous sixteenth, and the partial sums are added up by a 16-lane shuffle.  Float64 and
    // re-associated, like the reduce regime and under the same condition (groups of >= 256 members).
    if (tl.inner == 1 && a.K >= 256) {
      constexpr uint32_t L = 16;
      for (uint32_t slot = tid; slot < rows * L; slot += kBlock) {
        const uint32_t r = slot / L, part = slot % L;
        const uint32_t kb = (uint32_t)((uint64_t)a.K * part / L), ke = (uint32_t)((uint64_t)a.K * (part + 1) / L);
        const T *cells = tile + r * tl.row_elems;
        double acc = 0.0;
        uint32_t cnt = 0;  // contributions (cells that are set, i.e. non-zero): `average` divides by it
        uint32_t k = kb;
        for (; k + 4 <= ke; k += 4) {
          const T x0 = cells[k], x1 = cells[k + 1], x2 = cells[k + 2], x3 = cells[k + 3];
          acc += Cell<T>::to_f64(x0);
          acc += Cell<T>::to_f64(x1);
          acc += Cell<T>::to_f64(x2);
          acc += Cell<T>::to_f64(x3);
          if constexpr (METHOD != OLAP_SUM)
            cnt += (Cell<T>::is_default(x0, false) ? 0u : 1u) + (Cell<T>::is_default(x1, false) ? 0u : 1u) +
                   (Cell<T>::is_default(x2, false) ? 0u : 1u) + (Cell<T>::is_default(x3, false) ? 0u : 1u);
        }
        for (; k < ke; ++k) {
          acc += Cell<T>::to_f64(cells[k]);
          if constexpr (METHOD != OLAP_SUM) cnt += Cell<T>::is_default(cells[k], false) ? 0u : 1u;
        }
#pragma unroll
        for (uint32_t d = L / 2; d > 0; d >>= 1) {
          acc += __shfl_down(acc, d, L);
          if constexpr (METHOD != OLAP_SUM) cnt += __shfl_down(cnt, d, L);
        }
        if (part == 0) {
          Agg<METHOD> agg;
          agg.acc = acc;
          agg.count = cnt;
          agg.has = acc != 0.0 && (METHOD == OLAP_SUM || cnt != 0);
          agg.finish(def_nan);
          O ov;
          int32_t os;
          emit_out<T, METHOD>(agg.acc, agg.has, agg.count, def_nan, ov, os);
          dst[r] = ov;
          if (sdst) sdst[r] = os;
        }
      }
      return;
    }
  }
  if constexpr (ALL && !FAST && METHOD != OLAP_FIRST && METHOD != OLAP_LAST) {
    // the same sharing of a long row by 16 lanes for highest / lowest / product (and for masked / NaN-default sums): each
    // lane folds its contiguous part of the row into a Partial (value, set bit, contribution count) and the parts are
    // merged lane to lane — a lane per row left 15 of 256 lanes walking 271 members each ([3653,101,271] product -> all:
    // `highest` 232 -> 95 us, `product` 481 -> 90 us, against 59 us for `sum`).  first / last keep the lane per row: it
    // stops at the first / last set member, one LDS read on a dense cube.
    if (tl.inner == 1 && a.K >= 256) {
      constexpr uint32_t L = 16;
      for (uint32_t slot = tid; slot < rows * L; slot += kBlock) {
        const uint32_t r = slot / L, part = slot % L;
        const uint32_t kb = (uint32_t)((uint64_t)a.K * part / L), ke = (uint32_t)((uint64_t)a.K * (part + 1) / L);
        const T *cells = tile + r * tl.row_elems;
        if constexpr (METHOD == OLAP_HIGHEST || METHOD == OLAP_LOWEST) {
          // the running extreme in the cell's own type (hardware max / min for float cells), merged lane to lane
          Pick<T, METHOD> pk;
          pk.init();
          uint32_t k = kb;
          for (; k + 4 <= ke; k += 4) {  // four LDS reads in flight, then the four folds
            T x[4];
            int32_t sx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              x[u] = cells[k + u];
              sx[u] = HAS_STATUS ? stile[r * tl.row_elems + k + u] : OLAP_STATUS_SET;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) pk.add_if(cell_is_set<T>(x[u], sx[u], HAS_STATUS, def_nan), x[u]);
          }
          for (; k < ke; ++k) {
            const T x = cells[k];
            const int32_t sx = HAS_STATUS ? stile[r * tl.row_elems + k] : OLAP_STATUS_SET;
            pk.add_if(cell_is_set<T>(x, sx, HAS_STATUS, def_nan), x);
          }
#pragma unroll
          for (uint32_t d = L / 2; d > 0; d >>= 1) pk.merge(pk.shfl_down(d));  // (lanes past a row's 16 read a neighbour's state, which nobody consumes)
          if (part == 0) {
            dst[r] = pk.has ? pk.value() : Cell<T>::default_value(def_nan);
            if (sdst) sdst[r] = pk.has ? OLAP_STATUS_SET : 0;
          }
        } else {
          Partial p = partial_identity<METHOD>();
          for (uint32_t k = kb; k < ke; ++k) {
            const T x = cells[k];
            const int32_t sx = HAS_STATUS ? stile[r * tl.row_elems + k] : OLAP_STATUS_SET;
            if (cell_is_set<T>(x, sx, HAS_STATUS, def_nan)) partial_add<METHOD>(p, Cell<T>::to_f64(x), k, def_nan);
          }
#pragma unroll
          for (uint32_t d = L / 2; d > 0; d >>= 1) {
            const Partial q = partial_shfl_down(p, d);  // (lanes past a row's 16 read a neighbour's state, which nobody consumes)
            partial_merge<METHOD>(p, q, def_nan);
          }
          if (part == 0) {
            O ov;
            int32_t os;
            partial_finish<T, METHOD>(p, def_nan, ov, os);
            dst[r] = ov;
            if (sdst) sdst[r] = os;
          }
        }
      }
      return;
    }
  }
  // (Measured and dropped: tiles with few output cells over long groups — ten interleaved groups of a hundred members, four
  // rows per tile — sharing a cell among four lanes for highest / lowest: 90 -> 84 us there, but the mere presence of that
  // block in the kernel cost the ordinary path 25 % — day -> month with the day innermost, `highest`: 76 -> 96 us.)
  for (uint32_t idx = tid; idx < n_out; idx += kBlock) {
    const uint32_t r = idx / tl.out_row;
    const uint32_t rem = idx - r * tl.out_row;
    const uint32_t g = rem / tl.inner;
    const uint32_t i = rem - g * tl.inner;
    const uint32_t base = (PERMUTE ? r * tl.pitch_cells : r * tl.row_elems) + i;
    Lane<T, METHOD, HAS_STATUS, 1, FAST> lane;
    lane.init();
    uint32_t j = ALL ? 0u : PERMUTE ? l_gstart[2 * g] : l_gstart[g];
    const uint32_t jend = ALL ? (uint32_t)a.K : PERMUTE ? l_gstart[2 * g + 1] : l_gstart[g + 1];
    if constexpr (METHOD == OLAP_FIRST || METHOD == OLAP_LAST) {
      // the first / last SET member is the answer: walk towards it and stop there
      for (uint32_t t = 0; t < jend - j && !lane.pick[0].has; ++t) {
        const uint32_t jj = METHOD == OLAP_FIRST ? j + t : jend - 1 - t;
        const uint32_t k = MODE == 0 ? l_order[jj] : jj;
        const T xv = tile[base + k * tl.inner];
        const int32_t sv = HAS_STATUS ? stile[base + k * tl.inner] : OLAP_STATUS_SET;
        lane.pick[0].has = cell_is_set<T>(xv, sv, HAS_STATUS, def_nan);
        lane.pick[0].cur = xv;
      }
      j = jend;
    }
    if constexpr (METHOD == OLAP_PRODUCT) {
      // The plain chain first: prod *= (set ? v : 1).  It IS the reference's result unless a running product hit the
      // default on the way (then the key drops and the product restarts, in-memory.js:311-318) — and in that case the
      // plain product ends as 0 (default 0: a zero stays zero) or NaN (NaN default), which is what is tested; only then
      // is the group walked again through the exact state machine below.  Three instructions a member instead of a dozen.
      double prod = 1.0;
      bool any = false;
      for (uint32_t jj = j; jj < jend; ++jj) {
        const uint32_t k = MODE == 0 ? l_order[jj] : jj;
        const T xv = tile[base + k * tl.inner];
        const int32_t sv = HAS_STATUS ? stile[base + k * tl.inner] : OLAP_STATUS_SET;
        const bool set = cell_is_set<T>(xv, sv, HAS_STATUS, def_nan);
        prod *= set ? Cell<T>::to_f64(xv) : 1.0;
        any = any || set;
      }
      if (!any || (prod != 0.0 && prod == prod)) {
        T ov;
        int32_t os;
        emit_cell<T>(prod, any, def_nan, ov, os);
        reinterpret_cast<T *>(dst)[idx] = ov;
        if (sdst) sdst[idx] = os;
        continue;
      }
    }
    constexpr int UJ = (MODE == 0 || MODE == 3) ? 8 : 4;  // independent LDS reads in flight (MODE 0: member index, then cell — two dependent reads; MODE 3: long runs, few lanes)
    Vec<T, 1> x[UJ], y[UJ];
    Vec<int32_t, 1> sx[UJ], sy[UJ];
    auto fetch = [&](uint32_t jj, Vec<T, 1> *xx, Vec<int32_t, 1> *ss) {
      uint32_t kk[UJ];
#pragma unroll
      for (int u = 0; u < UJ; ++u) kk[u] = MODE == 0 ? l_order[jj + u] : (jj + u);
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        const uint32_t k = kk[u];
        xx[u].v[0] = tile[base + k * tl.inner];
        ss[u].v[0] = HAS_STATUS ? stile[base + k * tl.inner] : OLAP_STATUS_SET;
      }
    };
    // software pipeline: the next UJ members are requested before the current ones are folded in (a lane walks its
    // members one after the other — the reference's order — so its float64 adds are one dependent chain; the LDS
    // round trips must not sit in that chain as well)
    if (j + UJ <= jend) {
      fetch(j, x, sx);
      for (j += UJ; j + UJ <= jend; j += UJ) {
        fetch(j, y, sy);
#pragma unroll
        for (int u = 0; u < UJ; ++u) lane.add_row(x[u], sx[u], def_nan);
#pragma unroll
        for (int u = 0; u < UJ; ++u) {
          x[u] = y[u];
          sx[u] = sy[u];
        }
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u) lane.add_row(x[u], sx[u], def_nan);
    }
    for (; j < jend; ++j) {
      const uint32_t k = MODE == 0 ? l_order[j] : j;
      x[0].v[0] = tile[base + k * tl.inner];
      sx[0].v[0] = HAS_STATUS ? stile[base + k * tl.inner] : OLAP_STATUS_SET;
      lane.add_row(x[0], sx[0], def_nan);
    }
    lane.template finish_and_store<false>(def_nan, reinterpret_cast<T *>(dst), sdst, idx);
    if (idx == 0) g_probe3(blockIdx.x);
  }
}

template <typename T, int METHOD, bool HAS_STATUS, bool FAST, int MODE>
__global__ __launch_bounds__(kBlock) void drillup_tile_kernel(const Batch<T> b, const DrillUpAxis a, const DrillUpTile tl) {
  drillup_tile_body<T, METHOD, HAS_STATUS, FAST, MODE>(b.in[blockIdx.y], b.st_in[blockIdx.y], b.out[blockIdx.y], b.st_out[blockIdx.y], a, tl);
}

// Measures with DIFFERENT rules in one launch, row-tile regime (see drillup_rows_mixed_kernel): small cubes rolled up
// along an inner dimension, where a launch per rule is what costs.
template <typename T, bool HAS_STATUS, int MODE>
__global__ __launch_bounds__(kBlock) void drillup_tile_mixed_kernel(const Batch<T> b, const DrillUpAxis a, const DrillUpTile tl) {
  const T *in = b.in[blockIdx.y];
  const int32_t *st_in = b.st_in[blockIdx.y];
  T *out = b.out[blockIdx.y];
  int32_t *st_out = b.st_out[blockIdx.y];
  const bool fast = !HAS_STATUS && !a.def_nan;
  switch (b.method[blockIdx.y]) {
    case OLAP_SUM:
      if (fast) drillup_tile_body<T, OLAP_SUM, HAS_STATUS, !HAS_STATUS, MODE>(in, st_in, out, st_out, a, tl);
      else drillup_tile_body<T, OLAP_SUM, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl);
      break;
    case OLAP_AVERAGE:
      if (fast) drillup_tile_body<T, OLAP_AVERAGE, HAS_STATUS, !HAS_STATUS, MODE>(in, st_in, out, st_out, a, tl);
      else drillup_tile_body<T, OLAP_AVERAGE, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl);
      break;
    case OLAP_HIGHEST: drillup_tile_body<T, OLAP_HIGHEST, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl); break;
    case OLAP_LOWEST: drillup_tile_body<T, OLAP_LOWEST, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl); break;
    case OLAP_FIRST: drillup_tile_body<T, OLAP_FIRST, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl); break;
    case OLAP_LAST: drillup_tile_body<T, OLAP_LAST, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl); break;
    default: drillup_tile_body<T, OLAP_PRODUCT, HAS_STATUS, false, MODE>(in, st_in, out, st_out, a, tl); break;
  }
}


constexpr uint32_t kTotalBlocks = 4096;  // workgroups (and partial slots) of the store total
constexpr uint32_t kGroupTileMaxGroups = 1024;  // groups per tile of the group-tile regime (LDS holds their bounds)

// (Measured and dropped: the tile's bounds as ONE scalar load instead of two dependent ones, and the row-tile kernel's
// software-pipelined reduction loop — 62.4 -> 64 us on [900,3652,30] day -> month; skipping the gstart loads of a
// '-> all' roll-up in the row regime changed nothing.)
// Group-tile regime: contiguous groups (calendars) whose rows K*inner do not fit LDS — e.g. day ->
// month on [100, 3652, 30].  The members of consecutive groups are consecutive memory, so a tile is a
// run of whole GROUPS of one outer row (the plan cuts the group list into tiles of <= kTileBytes of
// cells, table `gtile`), staged with 16 B loads from the aligned address below its first cell; the
// reduction and the store are those of the row-tile kernel.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST>
__global__ __launch_bounds__(kBlock) void drillup_gtile_kernel(const Batch<T> b, const DrillUpAxis a, const uint64_t n_cells) {
  const T *__restrict__ in = b.in[blockIdx.y];
  const int32_t *__restrict__ st_in = b.st_in[blockIdx.y];
  T *__restrict__ out = b.out[blockIdx.y];
  int32_t *__restrict__ st_out = b.st_out[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int V = 16 / sizeof(T);
  constexpr uint32_t kCells = kTileBytes / sizeof(T);
  constexpr int NL = kCells / V / kBlock;
  T *tile = reinterpret_cast<T *>(lds_raw);
  int32_t *stile = reinterpret_cast<int32_t *>(lds_raw + kTileBytes);
  uint32_t *l_gstart = reinterpret_cast<uint32_t *>(lds_raw + kTileBytes + (HAS_STATUS ? kCells * 4 : 0));  // kGroupTileMaxGroups + 1 entries

  const uint32_t bid = a.xcd_order ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint32_t t = bid % a.n_gtile;
  const uint64_t o = bid / a.n_gtile;
  const uint32_t g0 = a.gtile[t], g1 = a.gtile[t + 1];
  const uint32_t ng = g1 - g0;
  const uint32_t k0 = a.gstart[g0], k1 = a.gstart[g1];
  const uint32_t inner = (uint32_t)a.inner;
  const uint64_t first = (o * a.K + k0) * a.inner;        // first cell of the tile
  const uint32_t n_in = (k1 - k0) * inner;                 // <= kCells - V
  const uint64_t base = first & ~(uint64_t)(V - 1);        // 16 B aligned cell below it
  const uint32_t shift = (uint32_t)(first - base);
  const uint32_t n_vec = (shift + n_in + V - 1) / V;       // <= kCells / V
  const bool def_nan = a.def_nan != 0;

  Vec<T, V> v[NL];
  Vec<int32_t, V> sv[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t i = threadIdx.x + u * kBlock;
    if (i < n_vec) {
      const uint64_t at = base + (uint64_t)i * V;
      if (at + V <= n_cells) {
        v[u] = load_stream<T, V>(in + at);
        if constexpr (HAS_STATUS) sv[u] = load_stream<int32_t, V>(st_in + at);
      } else {  // the very end of the buffer: cell by cell
#pragma unroll
        for (int e = 0; e < V; ++e) {
          v[u].v[e] = at + e < n_cells ? in[at + e] : T(0);
          if constexpr (HAS_STATUS) sv[u].v[e] = at + e < n_cells ? st_in[at + e] : 0;
        }
      }
    }
  }
  for (uint32_t i = threadIdx.x; i <= ng; i += kBlock) l_gstart[i] = a.gstart[g0 + i] - k0;
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t i = threadIdx.x + u * kBlock;
    if (i < n_vec) {
      *reinterpret_cast<Vec<T, V> *>(tile + i * V) = v[u];
      if constexpr (HAS_STATUS) *reinterpret_cast<Vec<int32_t, V> *>(stile + i * V) = sv[u];
    }
  }
  __syncthreads();

  const uint32_t n_out = ng * inner;
  if constexpr (!FAST) {
    // the same sharing of an output cell by L lanes for every other rule (and for masked / NaN-default sums): each lane
    // folds its contiguous part of the members into a Partial (the reduce regime's state: value, set bit, contribution
    // count, member position for first / last) and the parts are merged lane to lane in member order
    if (a.min_group >= 256 && n_out * 2 <= kBlock) {
      typedef typename OutCell<T, METHOD>::type O;
      uint32_t L = 2;
      while (L < 64 && L * 2 * n_out <= kBlock) L *= 2;
      const uint32_t cell = threadIdx.x / L, part = threadIdx.x % L;
      const bool live = cell < n_out;
      const uint32_t g = live ? cell / inner : 0u, i = live ? cell - g * inner : 0u;
      const uint32_t j0 = l_gstart[g], len = l_gstart[g + 1] - j0;
      uint32_t k = live ? j0 + (uint32_t)((uint64_t)len * part / L) : 0u;
      const uint32_t ke = live ? j0 + (uint32_t)((uint64_t)len * (part + 1) / L) : 0u;
      if constexpr (METHOD == OLAP_HIGHEST || METHOD == OLAP_LOWEST) {
        // the running extreme in the cell's own type (hardware max / min for float cells), merged lane to lane
        Pick<T, METHOD> pk;
        pk.init();
        for (; k + 4 <= ke; k += 4) {  // four LDS reads in flight, then the four folds
          T x[4];
          int32_t sx[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            x[u] = tile[shift + i + (k + u) * inner];
            sx[u] = HAS_STATUS ? stile[shift + i + (k + u) * inner] : OLAP_STATUS_SET;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) pk.add_if(cell_is_set<T>(x[u], sx[u], HAS_STATUS, def_nan), x[u]);
        }
        for (; k < ke; ++k) {
          const T x = tile[shift + i + k * inner];
          const int32_t sx = HAS_STATUS ? stile[shift + i + k * inner] : OLAP_STATUS_SET;
          pk.add_if(cell_is_set<T>(x, sx, HAS_STATUS, def_nan), x);
        }
        for (uint32_t d = L / 2; d > 0; d >>= 1) pk.merge(pk.shfl_down(d));
        if (live && part == 0) {
          const uint64_t at = (o * a.G + g0) * a.inner + cell;
          out[at] = pk.has ? pk.value() : Cell<T>::default_value(def_nan);
          if (st_out) st_out[at] = pk.has ? OLAP_STATUS_SET : 0;
        }
        return;
      }
      Partial p = partial_identity<METHOD>();
      for (; k < ke; ++k) {
        const T x = tile[shift + i + k * inner];
        const int32_t sx = HAS_STATUS ? stile[shift + i + k * inner] : OLAP_STATUS_SET;
        if (cell_is_set<T>(x, sx, HAS_STATUS, def_nan)) partial_add<METHOD>(p, Cell<T>::to_f64(x), k, def_nan);
      }
      for (uint32_t d = L / 2; d > 0; d >>= 1) {
        const Partial q = partial_shfl_down(p, d);  // (lanes past a cell's L read a neighbour's state, which nobody consumes)
        partial_merge<METHOD>(p, q, def_nan);
      }
      if (live && part == 0) {
        O ov;
        int32_t os;
        partial_finish<T, METHOD>(p, def_nan, ov, os);
        const uint64_t at = (o * a.G + g0) * a.inner + cell;
        reinterpret_cast<O *>(out)[at] = ov;
        if (st_out) st_out[at] = os;
      }
      return;
    }
  }
  if constexpr (FAST) {
    // Long groups, few output cells per tile (day -> year over [stores, 3652 days, 10 metrics]: a tile is ONE year of one
    // store, ten output cells of 365 members each): a lane per output cell leaves 246 lanes idle behind one wavefront's
    // 365-deep dependent chain.  L lanes share an output cell instead, each adds a contiguous L-th of the group's members,
    // and an L-lane shuffle adds the partial sums — float64, re-associated like the reduce regime and under the same
    // condition (every group of the roll-up has >= 256 members; sum / average over a 0 default without a mask).
    if (a.min_group >= 256 && n_out * 2 <= kBlock) {
      typedef typename OutCell<T, METHOD>::type O;
      uint32_t L = 2;
      while (L < 64 && L * 2 * n_out <= kBlock) L *= 2;
      const uint32_t cell = threadIdx.x / L, part = threadIdx.x % L;
      const bool live = cell < n_out;
      const uint32_t g = live ? cell / inner : 0u, i = live ? cell - g * inner : 0u;
      const uint32_t j0 = l_gstart[g], len = l_gstart[g + 1] - j0;
      uint32_t k = live ? j0 + (uint32_t)((uint64_t)len * part / L) : 0u;
      const uint32_t ke = live ? j0 + (uint32_t)((uint64_t)len * (part + 1) / L) : 0u;
      const T *cells = tile + shift + i;
      double acc = 0.0;
      uint32_t cnt = 0;  // contributions (cells that are set, i.e. non-zero): `average` divides by it
      for (; k + 4 <= ke; k += 4) {
        const T x0 = cells[k * inner], x1 = cells[(k + 1) * inner], x2 = cells[(k + 2) * inner], x3 = cells[(k + 3) * inner];
        acc += Cell<T>::to_f64(x0);
        acc += Cell<T>::to_f64(x1);
        acc += Cell<T>::to_f64(x2);
        acc += Cell<T>::to_f64(x3);
        if constexpr (METHOD != OLAP_SUM)
          cnt += (Cell<T>::is_default(x0, false) ? 0u : 1u) + (Cell<T>::is_default(x1, false) ? 0u : 1u) +
                 (Cell<T>::is_default(x2, false) ? 0u : 1u) + (Cell<T>::is_default(x3, false) ? 0u : 1u);
      }
      for (; k < ke; ++k) {
        acc += Cell<T>::to_f64(cells[k * inner]);
        if constexpr (METHOD != OLAP_SUM) cnt += Cell<T>::is_default(cells[k * inner], false) ? 0u : 1u;
      }
      for (uint32_t d = L / 2; d > 0; d >>= 1) {
        acc += __shfl_down(acc, d, L);
        if constexpr (METHOD != OLAP_SUM) cnt += __shfl_down(cnt, d, L);
      }
      if (live && part == 0) {
        Agg<METHOD> agg;
        agg.acc = acc;
        agg.count = cnt;
        agg.has = acc != 0.0 && (METHOD == OLAP_SUM || cnt != 0);
        agg.finish(def_nan);
        O ov;
        int32_t os;
        emit_out<T, METHOD>(agg.acc, agg.has, agg.count, def_nan, ov, os);
        const uint64_t at = (o * a.G + g0) * a.inner + cell;
        reinterpret_cast<O *>(out)[at] = ov;
        if (st_out) st_out[at] = os;
      }
      return;
    }
  }
  // (finish_and_store indexes the buffer in units of what METHOD writes: float64 partials under OLAP_PARTIAL_AVERAGE)
  T *dst = reinterpret_cast<T *>(reinterpret_cast<typename OutCell<T, METHOD>::type *>(out) + (o * a.G + g0) * a.inner);
  int32_t *sdst = st_out ? st_out + (o * a.G + g0) * a.inner : nullptr;
  for (uint32_t idx = threadIdx.x; idx < n_out; idx += kBlock) {
    const uint32_t g = idx / inner;
    const uint32_t i = idx - g * inner;
    const uint32_t at0 = shift + i;
    Lane<T, METHOD, HAS_STATUS, 1, FAST> lane;
    lane.init();
    uint32_t j = l_gstart[g];
    const uint32_t jend = l_gstart[g + 1];
    if constexpr (METHOD == OLAP_FIRST || METHOD == OLAP_LAST) {
      // the first / last SET member is the answer: walk towards it and stop there (dense cubes: one LDS read, not a group's worth)
      for (uint32_t t = 0; t < jend - j && !lane.pick[0].has; ++t) {
        const uint32_t k = METHOD == OLAP_FIRST ? j + t : jend - 1 - t;
        const T xv = tile[at0 + k * inner];
        const int32_t sv = HAS_STATUS ? stile[at0 + k * inner] : OLAP_STATUS_SET;
        lane.pick[0].has = cell_is_set<T>(xv, sv, HAS_STATUS, def_nan);
        lane.pick[0].cur = xv;
      }
      j = jend;
    }
    if constexpr (METHOD == OLAP_PRODUCT) {
      // (the plain chain first, the exact state machine only when it ends as 0 or NaN: see drillup_tile_body)
      double prod = 1.0;
      bool any = false;
      for (uint32_t jj = j; jj < jend; ++jj) {
        const T xv = tile[at0 + jj * inner];
        const int32_t sv = HAS_STATUS ? stile[at0 + jj * inner] : OLAP_STATUS_SET;
        const bool set = cell_is_set<T>(xv, sv, HAS_STATUS, def_nan);
        prod *= set ? Cell<T>::to_f64(xv) : 1.0;
        any = any || set;
      }
      if (!any || (prod != 0.0 && prod == prod)) {
        T ov;
        int32_t os;
        emit_cell<T>(prod, any, def_nan, ov, os);
        dst[idx] = ov;
        if (sdst) sdst[idx] = os;
        continue;
      }
    }
    constexpr int UJ = 4;
    Vec<T, 1> x[UJ];
    Vec<int32_t, 1> sx[UJ];
    for (; j + UJ <= jend; j += UJ) {
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        x[u].v[0] = tile[at0 + (j + u) * inner];
        sx[u].v[0] = HAS_STATUS ? stile[at0 + (j + u) * inner] : OLAP_STATUS_SET;
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u) lane.add_row(x[u], sx[u], def_nan);
    }
    for (; j < jend; ++j) {
      x[0].v[0] = tile[at0 + j * inner];
      sx[0].v[0] = HAS_STATUS ? stile[at0 + j * inner] : OLAP_STATUS_SET;
      lane.add_row(x[0], sx[0], def_nan);
    }
    lane.template finish_and_store<false>(def_nan, dst, sdst, idx);
  }
}

// One UNIT (a wavefront for short segments, a whole workgroup for long ones) per (outer, group,
// segment); its lanes cover `rows` consecutive members x `inner` cells per step, i.e. consecutive
// memory for contiguous groups: coalesced however small `inner` is.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST>
__global__ __launch_bounds__(kBlock) void drillup_reduce_kernel(const T *__restrict__ in,
                                                                const int32_t *__restrict__ st_in,
                                                                const DrillUpAxis a, const DrillUpReduce rd) {
  __shared__ Partial lds[kBlock];
  const uint32_t inner = (uint32_t)a.inner;
  const uint32_t units_per_block = kBlock / rd.unit;
  const uint64_t unit_id = (uint64_t)blockIdx.x * units_per_block + threadIdx.x / rd.unit;
  const uint64_t n_units = a.outer * a.G * rd.S;
  const bool live = unit_id < n_units;
  const uint32_t seg = (uint32_t)(unit_id % rd.S);
  const uint64_t og = live ? unit_id / rd.S : 0;
  const uint64_t g = og % a.G, o = og / a.G;
  const uint32_t lane = threadIdx.x % rd.unit;
  const uint32_t lds_base = threadIdx.x - lane;
  const bool active = live && lane < rd.rows * inner;
  const uint32_t i = lane % inner, r = lane / inner;
  const bool def_nan = a.def_nan != 0;
  const uint32_t gend = a.gstart[g + 1];
  const uint64_t s0 = (uint64_t)a.gstart[g] + (uint64_t)seg * rd.seg_len;
  const uint32_t jbeg = s0 < gend ? (uint32_t)s0 : gend;
  const uint32_t jend = (uint64_t)jbeg + rd.seg_len < gend ? jbeg + rd.seg_len : gend;
  const T *base = in + (o * a.K) * a.inner + i;
  const int32_t *sbase = HAS_STATUS ? st_in + (o * a.K) * a.inner + i : nullptr;
  Partial p = partial_identity<METHOD>();
  if (active) {
    constexpr int U = 8;
    for (uint32_t j = jbeg + r; j < jend; j += rd.rows * U) {
      T x[U];
      int32_t sx[U];
      uint32_t jj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        jj[u] = j + u * rd.rows;
        const uint32_t jc = jj[u] < jend ? jj[u] : j;
        const uint64_t k = a.order ? (uint64_t)a.order[jc] : (uint64_t)jc;
        x[u] = base[k * a.inner];
        sx[u] = HAS_STATUS ? sbase[k * a.inner] : OLAP_STATUS_SET;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (FAST) {
          // additive method, zero default, no mask: unset cells hold 0 (see Lane::add_row)
          const double v = jj[u] < jend ? Cell<T>::to_f64(x[u]) : 0.0;
          p.acc += v;
          p.meta += (v != 0.0) ? 1u : 0u;
        } else {
          if (jj[u] < jend && cell_is_set<T>(x[u], sx[u], HAS_STATUS, def_nan)) partial_add<METHOD>(p, Cell<T>::to_f64(x[u]), jj[u], def_nan);
        }
      }
    }
  }
  if constexpr (FAST) p.meta = (p.meta & 0x7FFFFFFFu) | ((p.meta != 0 && p.acc != 0.0) ? 0x80000000u : 0u);
  lds[lds_base + lane] = p;
  __syncthreads();
  for (uint32_t s = rd.rows >> 1; s > 0; s >>= 1) {
    if (active && r < s) {
      Partial q = lds[lds_base + lane + s * inner];
      partial_merge<METHOD>(p, q, def_nan);
      lds[lds_base + lane] = p;
    }
    __syncthreads();
  }
  if (active && r == 0) rd.part[((o * a.G + g) * a.inner + i) * rd.S + seg] = p;
}

// 16 B form of the cooperative reduction for '-> all' roll-ups of contiguous rows (G == 1, no member
// table, K*inner and the segment length multiples of V cells, V = 16 / sizeof(T): 4 cells of 4 bytes, 2 of 8): the unit
// streams `rows*inner` consecutive cells per step as one 16-byte group per lane; lane element e always lands on output
// cell (lane*V + e) % inner because the step is a whole number of rows.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST>
__global__ __launch_bounds__(kBlock) void drillup_reduce4_kernel(const T *__restrict__ in,
                                                                 const int32_t *__restrict__ st_in,
                                                                 T *__restrict__ out, int32_t *__restrict__ st_out,
                                                                 const DrillUpAxis a, const DrillUpReduce rd) {
  constexpr int V = 16 / sizeof(T);
  __shared__ Partial lds[kBlock * V];
  typedef typename OutCell<T, METHOD>::type O;  // float64 partials under OLAP_PARTIAL_AVERAGE
  O *outp = reinterpret_cast<O *>(out);
  const uint32_t inner = (uint32_t)a.inner;
  const uint32_t units_per_block = kBlock / rd.unit;
  const uint64_t unit_id = (uint64_t)blockIdx.x * units_per_block + threadIdx.x / rd.unit;
  const uint64_t n_units = a.outer * rd.S;  // G == 1
  const bool live = unit_id < n_units;
  const uint32_t seg = (uint32_t)(unit_id % rd.S);
  const uint64_t o = live ? unit_id / rd.S : 0;
  const uint32_t lane = threadIdx.x % rd.unit;
  const uint32_t lds_base = (threadIdx.x - lane) * V;
  const uint32_t step_cells = rd.rows * inner;          // multiple of V, <= unit * V
  const bool active = live && lane * V < step_cells;
  const bool def_nan = a.def_nan != 0;
  const uint64_t jbeg = (uint64_t)seg * rd.seg_len < a.K ? (uint64_t)seg * rd.seg_len : a.K;
  const uint64_t jend = jbeg + rd.seg_len < a.K ? jbeg + rd.seg_len : a.K;
  const uint64_t cell_beg = jbeg * inner, cell_end = jend * inner;  // within the row `o`
  const T *row = in + o * a.K * a.inner;
  const int32_t *srow = HAS_STATUS ? st_in + o * a.K * a.inner : nullptr;
  Partial p[V];
#pragma unroll
  for (int e = 0; e < V; ++e) p[e] = partial_identity<METHOD>();
  // highest / lowest / first / last: the running pick in the cell's own type (hardware max / min for float cells, selects —
  // no float64 round trip, no per-cell branch), turned into a Partial once, after the sweep.  A lane meets its cells in
  // ascending member order, so `first` keeps the first set one it meets and `last` the latest.
  constexpr bool kPick = IsPick<METHOD>::value && !FAST;
  Pick<T, METHOD> pk[V];
  uint32_t ppos[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    pk[e].init();
    ppos[e] = 0u;
  }
  // sum / average / product through the exact state machine (a mask, a NaN default, product): the running value, the
  // key's presence and the contribution count in registers of their own (Agg), packed into a Partial after the sweep
  constexpr bool kAgg = !IsPick<METHOD>::value && !FAST;
  Agg<METHOD> ag[V];
#pragma unroll
  for (int e = 0; e < V; ++e) ag[e].init();
  auto pick_cell = [&](int e, bool set, T x, uint32_t pos) {
    if constexpr (METHOD == OLAP_FIRST) ppos[e] = (set && !pk[e].has) ? pos : ppos[e];
    if constexpr (METHOD == OLAP_LAST) ppos[e] = set ? pos : ppos[e];
    pk[e].add_if(set, x);
  };
  if (active && rd.edge) {
    // inner == 1, the row starts anywhere inside a 16-byte group: sweep the ALIGNED groups that cover
    // [gbeg, gend) of the buffer and let cells outside the row contribute nothing
    constexpr int U = 4;
    const int64_t gbeg = (int64_t)(o * a.K) + (int64_t)cell_beg, gend = (int64_t)(o * a.K) + (int64_t)cell_end;
    const int64_t n_cells = (int64_t)(a.outer * a.K);
    for (int64_t gi = (gbeg & ~(int64_t)(V - 1)) + (int64_t)lane * V; gi < gend; gi += (int64_t)step_cells * U) {
      Vec<T, V> x[U];
      Vec<int32_t, V> sx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t g4 = gi + (int64_t)u * step_cells;
        if (g4 < gend && g4 + V <= n_cells) {
          x[u] = load_stream<T, V>(in + g4);
          if constexpr (HAS_STATUS) sx[u] = load_stream<int32_t, V>(st_in + g4);
        } else if (g4 < gend) {  // the last group of the buffer, cell by cell
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const bool ok = g4 + e < n_cells;
            x[u].v[e] = ok ? in[g4 + e] : T(0);
            if constexpr (HAS_STATUS) sx[u].v[e] = ok ? st_in[g4 + e] : 0;
          }
        } else {  // past the segment: nothing to read (the masks below drop these cells)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            x[u].v[e] = T(0);
            if constexpr (HAS_STATUS) sx[u].v[e] = 0;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t g4 = gi + (int64_t)u * step_cells;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const bool valid = g4 + e >= gbeg && g4 + e < gend;
          if constexpr (FAST) {
            const double v = valid ? Cell<T>::to_f64(x[u].v[e]) : 0.0;
            p[e].acc += v;
            p[e].meta += (v != 0.0) ? 1u : 0u;
          } else {
            const int32_t st = HAS_STATUS ? sx[u].v[e] : OLAP_STATUS_SET;
            const bool set = valid && cell_is_set<T>(x[u].v[e], st, HAS_STATUS, def_nan);
            if constexpr (kPick) pick_cell(e, set, x[u].v[e], (uint32_t)(g4 + e - (int64_t)(o * a.K)));
            else if (set) ag[e].add(Cell<T>::to_f64(x[u].v[e]), def_nan);
          }
        }
      }
    }
  } else if (active) {
    constexpr int U = 4;
    // member position of the lane's cells (first / last merge by it): a step is a whole number of rows, so it advances by
    // rd.rows per step — ONE division per cell of the lane here instead of a 64-bit division per cell of the cube
    // ([10^8] -> [1] highest / first / last: 150 us, of which the divisions were most)
    uint32_t pos0[V];
#pragma unroll
    for (int e = 0; e < V; ++e) pos0[e] = (uint32_t)((cell_beg + lane * V + e) / inner);
    uint32_t step = 0;
    for (uint64_t c = cell_beg + lane * V; c < cell_end; c += (uint64_t)step_cells * U, step += U) {
      Vec<T, V> x[U];
      Vec<int32_t, V> sx[U];
      uint64_t cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cc[u] = c + (uint64_t)u * step_cells;
        const uint64_t at = cc[u] < cell_end ? cc[u] : c;  // cell_end - cell_beg is a multiple of V
        x[u] = load_stream<T, V>(row + at);
        if constexpr (HAS_STATUS) sx[u] = load_stream<int32_t, V>(srow + at);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in_range = cc[u] < cell_end;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          if constexpr (FAST) {
            const double v = in_range ? Cell<T>::to_f64(x[u].v[e]) : 0.0;
            p[e].acc += v;
            p[e].meta += (v != 0.0) ? 1u : 0u;
          } else {
            const int32_t st = HAS_STATUS ? sx[u].v[e] : OLAP_STATUS_SET;
            const bool set = in_range && cell_is_set<T>(x[u].v[e], st, HAS_STATUS, def_nan);
            if constexpr (kPick) pick_cell(e, set, x[u].v[e], pos0[e] + (step + (uint32_t)u) * rd.rows);
            else if (set) ag[e].add(Cell<T>::to_f64(x[u].v[e]), def_nan);
          }
        }
      }
    }
  }
  const bool by_shuffle = inner * 2 <= (uint32_t)V || (inner % V == 0 && (rd.rows & (rd.rows - 1)) == 0);
  bool merged = false;  // the rows of a step are already folded (typed, below)
  if constexpr (kPick) {
    // A wavefront-sized unit (short segments: the last dimension of [10^5,1000] rolled up) merges its rows by wave
    // shuffles only — done here on the typed picks (a hardware max, or a compare of member positions, per level) before
    // they become Partials: the generic Partial merge is ~40 instructions a level, nine levels a segment, more than the
    // sweep of a 1 000-cell segment itself (highest / first / last 108-117 us against 62 us for sum)
    if (by_shuffle && rd.unit == 64) {
      auto fold = [&](Pick<T, METHOD> &x, uint32_t &xpos, const Pick<T, METHOD> &y, uint32_t ypos) {
        if constexpr (METHOD == OLAP_FIRST || METHOD == OLAP_LAST) {
          const bool take = y.has && (!x.has || (METHOD == OLAP_FIRST ? ypos < xpos : ypos >= xpos));
          x.cur = take ? y.cur : x.cur;
          xpos = take ? ypos : xpos;
          x.has = x.has || y.has;
        } else {
          x.merge(y);
        }
      };
      uint32_t m, top, ne = V;
      if (inner * 2 <= (uint32_t)V) {
        if constexpr (V == 4) {
          if (inner == 1) {
            fold(pk[0], ppos[0], pk[1], ppos[1]);
            fold(pk[2], ppos[2], pk[3], ppos[3]);
            fold(pk[0], ppos[0], pk[2], ppos[2]);
          } else {
            fold(pk[0], ppos[0], pk[2], ppos[2]);
            fold(pk[1], ppos[1], pk[3], ppos[3]);
          }
        } else {
          fold(pk[0], ppos[0], pk[1], ppos[1]);
        }
        m = 1;
        top = rd.unit;
        ne = inner;
      } else {
        m = inner / V;
        top = rd.rows;
      }
      for (uint32_t sft = top >> 1; sft > 0; sft >>= 1) {
        const uint32_t d = sft * m;  // <= 32: a unit is one wavefront
#pragma unroll
        for (int e = 0; e < V; ++e)
          if ((uint32_t)e < ne) {
            const Pick<T, METHOD> q = pk[e].shfl_down(d);
            const uint32_t qpos = __shfl_down(ppos[e], d, 64);
            fold(pk[e], ppos[e], q, qpos);
          }
      }
      merged = true;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      p[e].acc = pk[e].has ? Cell<T>::to_f64(pk[e].value()) : 0.0;
      p[e].meta = pk[e].has ? 0x80000001u : 0u;
      p[e].pos = ppos[e];
    }
  }
  if constexpr (kAgg) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      p[e].acc = ag[e].has ? ag[e].acc : 0.0;
      p[e].meta = (ag[e].has ? 0x80000000u : 0u) | (ag[e].count & 0x7FFFFFFFu);
    }
  }
  // When rows line up with lanes (inner a multiple of V, or a divisor of it: 1 / 2) the rows of a step are merged
  // lane to lane: row r + s of a step sits s*inner/4 lanes further on.  Distances up to 32 lanes stay
  // inside the first wavefront and are wave shuffles; a workgroup-wide unit folds the longer ones
  // through LDS first (three levels at most).
  // Idle lanes hold the identity.  With one segment per group the result is final and leaves from here.
  // (rows per step need not be a power of two — 10 rows of 100 cells fill 250 of a unit's 256 lanes where 8 fill 200 —
  // but the lane-to-lane merge needs one: other row counts take the LDS tree below)
  if (by_shuffle) {
    uint32_t m, top;  // lanes per row, rows spread over the unit's lanes
    uint32_t ne = V;  // accumulators per lane still in play
    if (merged) {
      m = 1;
      top = 1;  // nothing left to fold
      ne = inner * 2 <= (uint32_t)V ? inner : (uint32_t)V;
    } else if (inner * 2 <= (uint32_t)V) {  // a lane's 16 bytes hold several rows: fold them first
      if constexpr (V == 4) {
        if (inner == 1) {
          partial_merge_fast<METHOD, FAST>(p[0], p[1], def_nan);
          partial_merge_fast<METHOD, FAST>(p[2], p[3], def_nan);
          partial_merge_fast<METHOD, FAST>(p[0], p[2], def_nan);
        } else {
          partial_merge_fast<METHOD, FAST>(p[0], p[2], def_nan);
          partial_merge_fast<METHOD, FAST>(p[1], p[3], def_nan);
        }
      } else {
        partial_merge_fast<METHOD, FAST>(p[0], p[1], def_nan);  // (8-byte cells: inner == 1)
      }
      m = 1;
      top = rd.unit;
      ne = inner;
    } else {
      m = inner / V;
      top = rd.rows;
    }
    for (uint32_t s = top >> 1; s > 0; s >>= 1) {
      const uint32_t d = s * m;
      if (d > 32) {  // partner lane l + d may sit in another wavefront (only for rd.unit == kBlock): fold through LDS
#pragma unroll
        for (int e = 0; e < V; ++e)
          if ((uint32_t)e < ne) lds[e * kBlock + threadIdx.x] = p[e];  // 16 B per lane, lanes adjacent: no bank conflicts
        __syncthreads();
        if (threadIdx.x < d) {
#pragma unroll
          for (int e = 0; e < V; ++e)
            if ((uint32_t)e < ne) partial_merge_fast<METHOD, FAST>(p[e], lds[e * kBlock + threadIdx.x + d], def_nan);
        }
        __syncthreads();
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if ((uint32_t)e < ne) partial_merge_lane<METHOD, FAST>(p[e], d, def_nan);
      }
    }
    if constexpr (FAST) {
#pragma unroll
      for (int e = 0; e < V; ++e) partial_seal(p[e]);
    }
    if (live && lane * V < inner) {
      if (rd.S == 1) {
        if (inner % V == 0) {
          Vec<O, V> ov;
          Vec<int32_t, V> os;
#pragma unroll
          for (int e = 0; e < V; ++e) partial_finish<T, METHOD>(p[e], def_nan, ov.v[e], os.v[e]);
          store_vec<O, V>(outp + o * a.inner + lane * V, ov);
          if (st_out) store_vec<int32_t, V>(st_out + o * a.inner + lane * V, os);
        } else {
#pragma unroll
          for (int e = 0; e < V / 2; ++e) {  // constant indices: p[] must stay in registers
            if ((uint32_t)e < inner) {
              O ov;
              int32_t os;
              partial_finish<T, METHOD>(p[e], def_nan, ov, os);
              outp[o * a.inner + e] = ov;
              if (st_out) st_out[o * a.inner + e] = os;
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (lane * V + e < inner) rd.part[(o * a.inner + lane * V + e) * rd.S + seg] = p[e];
      }
    }
    return;  // the whole workgroup takes this path
  }
#pragma unroll
  for (int e = 0; e < V; ++e) lds[lds_base + lane * V + e] = p[e];
  __syncthreads();
  // tree over the rows of a step: flat slot f = r*inner + i.  Rows beyond the largest power of two are folded onto
  // the first rows before the tree starts.
  uint32_t pow2 = 1;
  while (pow2 * 2 <= rd.rows) pow2 *= 2;
  if (pow2 < rd.rows) {
    if (live) {
      for (uint32_t f = lane; f < (rd.rows - pow2) * inner; f += rd.unit) {
        Partial x = lds[lds_base + f];
        partial_merge_fast<METHOD, FAST>(x, lds[lds_base + f + pow2 * inner], def_nan);
        lds[lds_base + f] = x;
      }
    }
    __syncthreads();
  }
  for (uint32_t s = pow2 >> 1; s > 0; s >>= 1) {
    if (live) {
      for (uint32_t f = lane; f < s * inner; f += rd.unit) {
        Partial x = lds[lds_base + f];
        partial_merge_fast<METHOD, FAST>(x, lds[lds_base + f + s * inner], def_nan);
        lds[lds_base + f] = x;
      }
    }
    __syncthreads();
  }
  if (live) {
    for (uint32_t i = lane; i < inner; i += rd.unit) {
      Partial x = lds[lds_base + i];
      if constexpr (FAST) partial_seal(x);
      if (rd.S == 1) {
        O ov;
        int32_t os;
        partial_finish<T, METHOD>(x, def_nan, ov, os);
        outp[o * a.inner + i] = ov;
        if (st_out) st_out[o * a.inner + i] = os;
      } else {
        rd.part[(o * a.inner + i) * rd.S + seg] = x;
      }
    }
  }
}

// Lane-per-(cell, segment) form for wide `inner` (> 128): lanes along `inner` are already coalesced.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST>
__global__ __launch_bounds__(kBlock) void drillup_split_kernel(const T *__restrict__ in,
                                                               const int32_t *__restrict__ st_in,
                                                               const DrillUpAxis a, const DrillUpReduce rd) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t cells = a.outer * a.G * a.inner;
  if (t >= cells * rd.S) return;
  const uint64_t i = t % a.inner;
  const uint32_t seg = (uint32_t)((t / a.inner) % rd.S);
  const uint64_t og = t / (a.inner * rd.S);
  const uint64_t g = og % a.G, o = og / a.G;
  const bool def_nan = a.def_nan != 0;
  const T *base = in + (o * a.K) * a.inner + i;
  const int32_t *sbase = HAS_STATUS ? st_in + (o * a.K) * a.inner + i : nullptr;
  const uint32_t gend = a.gstart[g + 1];
  const uint64_t s0 = (uint64_t)a.gstart[g] + (uint64_t)seg * rd.seg_len;
  uint32_t j = s0 < gend ? (uint32_t)s0 : gend;
  const uint32_t jend = (uint64_t)j + rd.seg_len < gend ? j + rd.seg_len : gend;
  Partial p = partial_identity<METHOD>();
  constexpr int U = 8;
  const bool contig = a.order == nullptr;
  uint64_t at = (uint64_t)j * a.inner;  // (no 64-bit vector multiply per load: see drillup_split4_kernel)
  for (; j < jend; j += U) {
    const uint32_t n = (jend - j) < (uint32_t)U ? (jend - j) : (uint32_t)U;
    T x[U];
    int32_t sx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t jj = (uint32_t)u < n ? j + u : j;
      const uint64_t off = contig ? at + ((uint32_t)u < n ? (uint64_t)u * a.inner : 0ull) : (uint64_t)a.order[jj] * a.inner;
      x[u] = base[off];
      sx[u] = HAS_STATUS ? sbase[off] : OLAP_STATUS_SET;
    }
    at += (uint64_t)U * a.inner;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (FAST) {
        const double v = (uint32_t)u < n ? Cell<T>::to_f64(x[u]) : 0.0;
        p.acc += v;
        p.meta += (v != 0.0) ? 1u : 0u;
      } else {
        if ((uint32_t)u < n && cell_is_set<T>(x[u], sx[u], HAS_STATUS, def_nan)) partial_add<METHOD>(p, Cell<T>::to_f64(x[u]), j + u, def_nan);
      }
    }
  }
  if constexpr (FAST) p.meta = (p.meta & 0x7FFFFFFFu) | ((p.meta != 0 && p.acc != 0.0) ? 0x80000000u : 0u);
  rd.part[((o * a.G + g) * a.inner + i) * rd.S + seg] = p;
}

// One wavefront per output cell merges its S partial states (segments are ascending member ranges).
// The same with 16-byte lanes (4-byte cells, inner a multiple of 4, aligned buffers): a lane owns FOUR adjacent output
// cells of one (outer, group, segment) and streams its segment's rows with non-temporal 16-byte loads, four in flight.
// The scalar form above moves 4 bytes per lane with cached loads: [1e4,1e4] -> [1,1e4] 185 us (0.27), [1e5,1000] ->
// [1,1000] 128 us (0.39); this one: see DESIGN.md K1r.
template <typename T, int METHOD, bool HAS_STATUS, bool FAST>
__global__ __launch_bounds__(kBlock) void drillup_split4_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in,
                                                                const DrillUpAxis a, const DrillUpReduce rd) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t n_vec = a.inner / 4;
  const uint64_t groups = a.outer * a.G;
  if (t >= groups * n_vec * rd.S) return;
  const uint64_t iv = t % n_vec;
  const uint32_t seg = (uint32_t)((t / n_vec) % rd.S);
  const uint64_t og = t / (n_vec * rd.S);
  const uint64_t g = og % a.G, o = og / a.G;
  const bool def_nan = a.def_nan != 0;
  const T *base = in + (o * a.K) * a.inner + iv * 4;
  const int32_t *sbase = HAS_STATUS ? st_in + (o * a.K) * a.inner + iv * 4 : nullptr;
  const uint32_t gend = a.gstart[g + 1];
  const uint64_t s0 = (uint64_t)a.gstart[g] + (uint64_t)seg * rd.seg_len;
  uint32_t j = s0 < gend ? (uint32_t)s0 : gend;
  const uint32_t jend = (uint64_t)j + rd.seg_len < gend ? j + rd.seg_len : gend;
  Partial p[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e] = partial_identity<METHOD>();
  constexpr int U = 4;
  // (a lane's segment is its own, so its row index lives in a vector register: `row * inner` per load is a 64-bit
  // VECTOR multiply — quarter-rate instructions on every load's critical path with a handful of waves per SIMD.  With
  // contiguous groups the rows of a segment are consecutive: one multiply per lane, then additions of u * inner, which
  // is wave-uniform.)
  const bool contig = a.order == nullptr;
  uint64_t at = (uint64_t)j * a.inner;  // contiguous groups: first cell of row j
  for (; j < jend; j += U) {
    const uint32_t n = (jend - j) < (uint32_t)U ? (jend - j) : (uint32_t)U;
    Vec<T, 4> x[U];
    Vec<int32_t, 4> sx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t jj = (uint32_t)u < n ? j + u : j;  // clamp: re-reads a valid row, result unused
      const uint64_t off = contig ? at + ((uint32_t)u < n ? (uint64_t)u * a.inner : 0ull) : (uint64_t)a.order[jj] * a.inner;
      x[u] = load_stream<T, 4>(base + off);
      if constexpr (HAS_STATUS) sx[u] = load_stream<int32_t, 4>(sbase + off);
    }
    at += (uint64_t)U * a.inner;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (FAST) {
          const double v = (uint32_t)u < n ? Cell<T>::to_f64(x[u].v[e]) : 0.0;
          p[e].acc += v;
          p[e].meta += (v != 0.0) ? 1u : 0u;
        } else {
          const int32_t st = HAS_STATUS ? sx[u].v[e] : OLAP_STATUS_SET;
          if ((uint32_t)u < n && cell_is_set<T>(x[u].v[e], st, HAS_STATUS, def_nan)) partial_add<METHOD>(p[e], Cell<T>::to_f64(x[u].v[e]), j + u, def_nan);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (FAST) p[e].meta = (p[e].meta & 0x7FFFFFFFu) | ((p[e].meta != 0 && p[e].acc != 0.0) ? 0x80000000u : 0u);
    rd.part[((o * a.G + g) * a.inner + iv * 4 + e) * rd.S + seg] = p[e];
  }
}

template <typename T, int METHOD>
__global__ __launch_bounds__(kBlock) void drillup_merge_kernel(T *__restrict__ out, int32_t *__restrict__ st_out,
                                                               const DrillUpAxis a, const DrillUpReduce rd) {
  const uint64_t cell = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (cell >= a.outer * a.G * a.inner) return;  // whole waves leave together
  const bool def_nan = a.def_nan != 0;
  Partial p = partial_identity<METHOD>();
  for (uint32_t s = lane; s < rd.S; s += 64) {
    const Partial q = rd.part[cell * rd.S + s];
    partial_merge<METHOD>(p, q, def_nan);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Partial q;
    q.acc = __shfl_down(p.acc, off, 64);
    q.meta = __shfl_down(p.meta, off, 64);
    q.pos = __shfl_down(p.pos, off, 64);
    partial_merge<METHOD>(p, q, def_nan);
  }
  if (lane == 0) {
    typename OutCell<T, METHOD>::type ov;
    int32_t os;
    partial_finish<T, METHOD>(p, def_nan, ov, os);
    reinterpret_cast<typename OutCell<T, METHOD>::type *>(out)[cell] = ov;
    if (st_out) st_out[cell] = os;
  }
}

// Many segments (S >= 512, i.e. very few output cells): one WORKGROUP per output cell, 4 partials in
// flight per lane — a single wavefront walking 4096 partials 64 at a time is a 30 us latency chain.
template <typename T, int METHOD>
__global__ __launch_bounds__(kBlock) void drillup_merge_block_kernel(T *__restrict__ out, int32_t *__restrict__ st_out,
                                                                     const DrillUpAxis a, const DrillUpReduce rd) {
  __shared__ Partial wave_part[kBlock / 64];
  const uint64_t cell = blockIdx.x;
  const bool def_nan = a.def_nan != 0;
  const Partial *src = rd.part + cell * rd.S;
  Partial p = partial_identity<METHOD>();
  constexpr uint32_t U = 4;
  for (uint32_t s0 = threadIdx.x; s0 < rd.S; s0 += kBlock * U) {
    Partial q[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t si = s0 + u * kBlock;
      q[u] = si < rd.S ? src[si] : partial_identity<METHOD>();
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) partial_merge<METHOD>(p, q[u], def_nan);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const Partial q = partial_shfl_down(p, off);
    partial_merge<METHOD>(p, q, def_nan);
  }
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (uint32_t w = 1; w < kBlock / 64; ++w) partial_merge<METHOD>(p, wave_part[w], def_nan);
    typename OutCell<T, METHOD>::type ov;
    int32_t os;
    partial_finish<T, METHOD>(p, def_nan, ov, os);
    reinterpret_cast<typename OutCell<T, METHOD>::type *>(out)[cell] = ov;
    if (st_out) st_out[cell] = os;
  }
}

// Few segments (S <= 16): one LANE per output cell merges them in order; stores are coalesced.
template <typename T, int METHOD>
__global__ __launch_bounds__(kBlock) void drillup_merge_few_kernel(T *__restrict__ out, int32_t *__restrict__ st_out,
                                                                   const DrillUpAxis a, const DrillUpReduce rd) {
  const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (cell >= a.outer * a.G * a.inner) return;
  const bool def_nan = a.def_nan != 0;
  Partial p = rd.part[cell * rd.S];
  for (uint32_t s = 1; s < rd.S; ++s) partial_merge<METHOD>(p, rd.part[cell * rd.S + s], def_nan);
  typename OutCell<T, METHOD>::type ov;
  int32_t os;
  partial_finish<T, METHOD>(p, def_nan, ov, os);
  reinterpret_cast<typename OutCell<T, METHOD>::type *>(out)[cell] = ov;
  if (st_out) st_out[cell] = os;
}

// Segmented rows (few output cells, long groups, rows wider than the cooperative forms take): every group's member
// list is cut into segments, the ROW regime — the headline kernel, 16-byte or ragged lanes and all — reduces each
// segment as if it were a group of its own (float64 partial sums + contribution counts for sum / average, the typed
// pick for highest / lowest / first / last), and this kernel folds a group's segments, in order, into the output
// cell: 64 adjacent cells x 4 lanes per cell, each lane a contiguous quarter of the segments, combined in lane order
// (deterministic; additive methods are re-associated exactly as in the cooperative forms).
struct SegmentedRows {
  uint32_t S_tot;             // segments of all groups
  const uint32_t *gstart;     // device [S_tot + 1]: the segments as contiguous member runs (they index the plan's `order`)
  const uint32_t *seg_start;  // device [G + 1]: first segment of every group
  void *partial;              // [outer, S_tot, inner]: float64 sums, or typed picks
  int32_t *aux;               // [outer, S_tot, inner]: contribution counts, or the picks' masks (where needed)
};

template <typename T, int METHOD>
__global__ __launch_bounds__(kBlock) void segments_combine_kernel(const void *__restrict__ partial, const int32_t *__restrict__ aux, T *__restrict__ out,
                                                                  int32_t *__restrict__ st_out, const DrillUpAxis a, const SegmentedRows sg) {
  constexpr bool kAdd = (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE);
  __shared__ double l_acc[kBlock];
  __shared__ uint32_t l_cnt[kBlock];
  const uint32_t x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const uint64_t strips = (a.inner + 63) / 64;
  const uint64_t og = blockIdx.x / strips;
  const uint64_t i = (blockIdx.x - og * strips) * 64 + x;
  const uint64_t g = og % a.G, o = og / a.G;
  const bool def_nan = a.def_nan != 0;
  const bool live = i < a.inner;
  const uint32_t s0 = sg.seg_start[g], s1 = sg.seg_start[g + 1];
  const uint32_t per = (s1 - s0 + 3) / 4;
  const uint32_t sb = s0 + y * per < s1 ? s0 + y * per : s1, se = sb + per < s1 ? sb + per : s1;
  const uint64_t base = o * sg.S_tot * a.inner + i;
  double acc = 0.0;
  uint32_t cnt = 0;
  Pick<T, kAdd ? OLAP_FIRST : METHOD> pk;
  pk.init();
  if (live) {
    constexpr int U = 8;
    for (uint32_t s = sb; s < se; s += U) {
      if constexpr (kAdd) {
        double v[U];
        uint32_t c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = s + u < se;
          const uint64_t at = base + (uint64_t)(in ? s + u : sb) * a.inner;
          v[u] = in ? ((const double *)partial)[at] : 0.0;
          c[u] = (in && aux) ? (uint32_t)aux[at] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc += v[u];
          cnt += c[u];
        }
      } else {
        T v[U];
        int32_t f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = s + u < se;
          const uint64_t at = base + (uint64_t)(in ? s + u : sb) * a.inner;
          v[u] = ((const T *)partial)[at];
          f[u] = in ? (aux ? aux[at] : OLAP_STATUS_SET) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pk.add_if(cell_is_set<T>(v[u], f[u], true, def_nan), v[u]);
      }
    }
  }
  if constexpr (kAdd) {
    l_acc[threadIdx.x] = acc;
    l_cnt[threadIdx.x] = cnt;
  } else if constexpr (sizeof(T) == 4) {  // a pick and whether there is one: 4-byte cells ride the count slots
    l_acc[threadIdx.x] = pk.has ? 1.0 : 0.0;
    reinterpret_cast<T *>(l_cnt)[threadIdx.x] = pk.value();
  } else {  // 8-byte cells ride the float64 slots
    reinterpret_cast<T *>(l_acc)[threadIdx.x] = pk.value();
    l_cnt[threadIdx.x] = pk.has ? 1u : 0u;
  }
  __syncthreads();
  if (y != 0 || !live) return;
  T ov;
  int32_t os;
  if constexpr (kAdd) {
    for (uint32_t q = 1; q < 4; ++q) {
      acc += l_acc[q * 64 + x];
      cnt += l_cnt[q * 64 + x];
    }
    finish_cell<T, double>(METHOD == OLAP_AVERAGE ? OLAP_FINISH_AVERAGE : OLAP_FINISH_ROUND, acc, cnt, aux != nullptr, def_nan, ov, os);
  } else {
    for (uint32_t q = 1; q < 4; ++q) {
      T v;
      bool has;
      if constexpr (sizeof(T) == 8) {
        v = reinterpret_cast<const T *>(l_acc)[q * 64 + x];
        has = l_cnt[q * 64 + x] != 0;
      } else {
        v = reinterpret_cast<const T *>(l_cnt)[q * 64 + x];
        has = l_acc[q * 64 + x] != 0.0;
      }
      pk.add_if(has, v);
    }
    ov = pk.has ? pk.value() : Cell<T>::default_value(def_nan);
    os = pk.has ? OLAP_STATUS_SET : 0;
  }
  const uint64_t at = (o * a.G + g) * a.inner + i;
  out[at] = ov;
  if (st_out) st_out[at] = os;
}

// ======================================================================= K1g: drillUp, any maps
// The store method accepts a map on every dimension (in-memory.js:270-274).  One lane per output
// cell walks the cartesian product of its groups' member lists in ascending flat order.
struct DrillUpGeneric {
  int nd;                       // collapsed dims
  uint32_t new_len[kMaxDims];
  uint64_t old_stride[kMaxDims];
  int32_t csr[kMaxDims];        // -1: identity dim; else offset of this dim's gstart in `tab`
  int32_t ord[kMaxDims];        // offset of this dim's order list in `tab`
  const uint32_t *tab;          // device
  uint64_t total;               // output cells
  int def_nan;
};

template <typename T, int METHOD, bool HAS_STATUS>
__global__ __launch_bounds__(kBlock) void drillup_generic_kernel(const T *__restrict__ in,
                                                                 const int32_t *__restrict__ st_in,
                                                                 T *__restrict__ out,
                                                                 int32_t *__restrict__ st_out,
                                                                 const DrillUpGeneric a) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.total) return;
  const bool def_nan = a.def_nan != 0;
  uint32_t lo[kMaxDims], hi[kMaxDims], cur[kMaxDims];
  uint64_t fixed = 0;  // offset contributed by identity dims
  {
    uint64_t c = t;
    for (int d = a.nd - 1; d >= 0; --d) {
      const uint32_t digit = (uint32_t)(c % a.new_len[d]);
      c /= a.new_len[d];
      if (a.csr[d] < 0) {
        fixed += (uint64_t)digit * a.old_stride[d];
        lo[d] = hi[d] = 0;
      } else {
        lo[d] = a.tab[a.csr[d] + digit];
        hi[d] = a.tab[a.csr[d] + digit + 1];
      }
      cur[d] = lo[d];
    }
  }
  bool empty = false;
  for (int d = 0; d < a.nd; ++d)
    if (a.csr[d] >= 0 && lo[d] == hi[d]) empty = true;

  Agg<METHOD> agg;
  agg.init();
  while (!empty) {
    uint64_t off = fixed;
    for (int d = 0; d < a.nd; ++d)
      if (a.csr[d] >= 0) off += (uint64_t)a.tab[a.ord[d] + cur[d]] * a.old_stride[d];
    const T x = in[off];
    const int32_t sx = HAS_STATUS ? st_in[off] : OLAP_STATUS_SET;
    if (cell_is_set<T>(x, sx, HAS_STATUS, def_nan)) agg.add(Cell<T>::to_f64(x), def_nan);
    // odometer, last changed dim fastest == ascending old flat index
    int d = a.nd - 1;
    for (; d >= 0; --d) {
      if (a.csr[d] < 0) continue;
      if (++cur[d] < hi[d]) break;
      cur[d] = lo[d];
    }
    if (d < 0) break;
  }
  agg.finish(def_nan);
  typename OutCell<T, METHOD>::type ov;
  int32_t os;
  emit_out<T, METHOD>(agg.acc, agg.has, agg.count, def_nan, ov, os);
  reinterpret_cast<typename OutCell<T, METHOD>::type *>(out)[t] = ov;
  if (st_out) st_out[t] = os;
}

// ======================================================================= K2: dice / load / reorder
// All three are index-remapped copies.  `Remap` describes, for each collapsed destination
// dimension, how a destination digit contributes to the source offset: either digit*stride
// (untouched dims, no table) or a table entry (int64 offset, -1 = no source cell).
struct Remap {
  int nd;
  uint32_t len[kMaxDims];      // extents of the iteration space (collapsed)
  uint64_t stride[kMaxDims];   // arithmetic dims: offset = digit * stride
  int32_t tab_off[kMaxDims];   // -1: arithmetic; else start of this dim's int64 table in `tab`
  const int64_t *tab;          // device
  uint64_t total;              // iteration space / VEC
  int def_nan;
  int src_def_nan;             // load: the other store's default kind
};

// dice (in-memory.js:213-263) and reorder (:178-211): iterate the DESTINATION, gather from the
// source.  Destination cells without a source stay unset.  VEC > 1 only when the innermost
// collapsed dim is contiguous in both (plan guarantees divisibility).
// IDX: uint32_t when every index of the launch fits 32 bits (the usual case: 64-bit divisions are
// several times dearer and the decode is most of this kernel's instruction count), else uint64_t.
template <typename T, bool HAS_STATUS, int VEC, typename IDX = uint64_t>
__global__ __launch_bounds__(kBlock) void gather_kernel(const T *__restrict__ in,
                                                        const int32_t *__restrict__ st_in,
                                                        T *__restrict__ out,
                                                        int32_t *__restrict__ st_out, const Remap r) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= r.total) return;
  const bool def_nan = r.def_nan != 0;
  IDX c = (IDX)(t * VEC);
  uint64_t src = 0;
  bool ok = true;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < r.nd) {
      const IDX q = c / (IDX)r.len[d];
      const uint32_t digit = (uint32_t)(c - q * (IDX)r.len[d]);
      c = q;
      if (r.tab_off[d] < 0) {
        src += (uint64_t)digit * r.stride[d];
      } else {
        const int64_t o = r.tab[r.tab_off[d] + digit];
        if (o < 0) ok = false;
        else src += (uint64_t)o;
      }
    }
  }
  Vec<T, VEC> ov;
  Vec<int32_t, VEC> os;
  if (ok) {
    // the source may be re-read (a drillDown broadcast reads each parent many times): cached loads;
    // the destination is written once: streaming stores
    const Vec<T, VEC> v = load_vec<T, VEC>(in + src);
    Vec<int32_t, VEC> s;
    if constexpr (HAS_STATUS) s = load_vec<int32_t, VEC>(st_in + src);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool set = cell_is_set<T>(v.v[e], HAS_STATUS ? s.v[e] : OLAP_STATUS_SET, HAS_STATUS, def_nan);
      ov.v[e] = set ? v.v[e] : Cell<T>::default_value(def_nan);
      os.v[e] = set ? OLAP_STATUS_SET : 0;
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      ov.v[e] = Cell<T>::default_value(def_nan);
      os.v[e] = 0;
    }
  }
  store_stream<T, VEC>(out + t * VEC, ov);
  if (st_out) store_stream<int32_t, VEC>(st_out + t * VEC, os);
}

// K5: fused dice -> drillUp.  Iterates the OUTPUT cube; the source offset of an output cell is the
// sum of per-dimension table entries (the dice selections) and the rolled-up dimension contributes
// one offset per group member.  Only surviving cells are read, each once.
struct GatherReduce {
  Remap r;                   // output dims; r.tab_off[axis] is unused
  int axis;                  // collapsed index of the rolled-up dimension
  const uint32_t *gstart;    // [G + 1]
  const int64_t *member_off; // source offset of each member (ascending within a group)
};

template <typename T, int METHOD, bool HAS_STATUS, int VEC, bool FAST>
__global__ __launch_bounds__(kBlock) void gather_reduce_kernel(const T *__restrict__ in,
                                                               const int32_t *__restrict__ st_in,
                                                               T *__restrict__ out,
                                                               int32_t *__restrict__ st_out,
                                                               const GatherReduce a) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.r.total) return;
  const bool def_nan = a.r.def_nan != 0;
  uint64_t c = t * VEC, base = 0;
  uint32_t g = 0;
  bool ok = true;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < a.r.nd) {
      const uint32_t digit = (uint32_t)(c % a.r.len[d]);
      c /= a.r.len[d];
      if (d == a.axis) {
        g = digit;
      } else if (a.r.tab_off[d] < 0) {
        base += (uint64_t)digit * a.r.stride[d];
      } else {
        const int64_t o = a.r.tab[a.r.tab_off[d] + digit];
        if (o < 0) ok = false;
        else base += (uint64_t)o;
      }
    }
  }
  Lane<T, METHOD, HAS_STATUS, VEC, FAST> lane;
  lane.init();
  if (ok) {
    uint32_t j = a.gstart[g];
    const uint32_t jend = a.gstart[g + 1];
    constexpr int U = 4;
    for (; j < jend; j += U) {
      const uint32_t n = (jend - j) < (uint32_t)U ? (jend - j) : (uint32_t)U;
      uint64_t off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) off[u] = base + (uint64_t)a.member_off[(uint32_t)u < n ? j + u : j];
      Vec<T, VEC> v[U];
      Vec<int32_t, VEC> s[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = load_vec<T, VEC>(in + off[u]);
        if constexpr (HAS_STATUS) s[u] = load_vec<int32_t, VEC>(st_in + off[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((uint32_t)u < n) lane.add_row(v[u], s[u], def_nan);
    }
  }
  lane.template finish_and_store<false>(def_nan, out, st_out, t * VEC);
}

// load (in-memory.js:139-176): iterate the SOURCE (the other store, dense over all its cells,
// see the comment at :152-158), scatter into this store with setValue semantics:
// mine.setValue(myIdx, his.getValue(hisIdx)).  The plan merges the dimensions the load leaves alone into arithmetic
// strides; a lane moves VEC cells (16 bytes) when the innermost merged run is contiguous on both sides, the source is
// streamed (every cell is read exactly once), and the index is decoded in 32-bit arithmetic when the other cube has
// fewer than 2^32 cells (a 64-bit division is a long software sequence on CDNA).
template <typename T, bool HAS_STATUS, int VEC, typename IDX>
__global__ __launch_bounds__(kBlock) void load_scatter_kernel(const T *__restrict__ his,
                                                              const int32_t *__restrict__ his_st,
                                                              T *__restrict__ mine,
                                                              int32_t *__restrict__ mine_st, const Remap r) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= r.total) return;
  const bool def_nan = r.def_nan != 0, his_nan = r.src_def_nan != 0;
  IDX c = (IDX)(t * VEC);
  uint64_t dst = 0;
  bool ok = true;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < r.nd) {
      const IDX q = c / (IDX)r.len[d];
      const uint32_t digit = (uint32_t)(c - q * (IDX)r.len[d]);
      c = q;
      if (r.tab_off[d] < 0) {
        dst += (uint64_t)digit * r.stride[d];
      } else {
        const int64_t o = r.tab[r.tab_off[d] + digit];
        if (o < 0) ok = false;
        else dst += (uint64_t)o;
      }
    }
  }
  if (!ok) return;  // an item this store does not have: its cells are not loaded
  const Vec<T, VEC> x = load_stream<T, VEC>(his + t * VEC);
  Vec<int32_t, VEC> xs;
  if constexpr (HAS_STATUS) xs = load_stream<int32_t, VEC>(his_st + t * VEC);
  Vec<T, VEC> ov;
  Vec<int32_t, VEC> os;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const bool his_set = cell_is_set<T>(x.v[e], HAS_STATUS ? xs.v[e] : OLAP_STATUS_SET, HAS_STATUS, his_nan);
    // getValue: the stored value or HIS default; then setValue against MY default.  For integer
    // cells a NaN default has no typed representation: an unset source cell unsets the target.
    bool set;
    T v;
    if (his_set) {
      v = x.v[e];
      set = !Cell<T>::is_default(v, def_nan);
    } else {
      // his.getValue() hands out HIS default (:119), which my setValue keeps unless it is MY default
      // (:122-133): 0 under my NaN default is a set cell; NaN under my 0 default is one for float cells
      v = Cell<T>::default_value(his_nan);
      constexpr bool is_float = (Cell<T>::dtype == OLAP_FLOAT32 || Cell<T>::dtype == OLAP_FLOAT64);
      set = his_nan ? (is_float && !def_nan) : def_nan;
    }
    ov.v[e] = set ? v : Cell<T>::default_value(def_nan);
    os.v[e] = set ? OLAP_STATUS_SET : 0;
  }
  store_stream<T, VEC>(mine + dst, ov);
  if (mine_st) store_stream<int32_t, VEC>(mine_st + dst, os);
}

// The same when the INNERMOST dimension itself is remapped (his items of the last dimension sit elsewhere in mine):
// no 16 contiguous bytes on my side, but a lane still reads R consecutive cells of his with one 16-byte streaming load
// and decodes its index ONCE — the divisions are what bounds the one-cell-per-lane form (~100 instructions per 4 bytes:
// 358 us for 10^8 cells); the cells that follow in the same innermost row only add their own innermost offset, and a
// cell past the row's end is decoded afresh.  [10]^8 with the items of the last dimension permuted: 358 -> 272 us (0.37).
// (Measured and dropped: the GATHER form — walk MY cells with inverse item maps, 16-byte stores, scattered 4-byte
// loads of his: 299 us.  Neither the stores nor the loads are what costs here but the dependent chain index -> table
// entry -> address -> cell per lane.)
template <typename T, bool HAS_STATUS, typename IDX>
__global__ __launch_bounds__(kBlock) void load_scatter_run_kernel(const T *__restrict__ his, const int32_t *__restrict__ his_st, T *__restrict__ mine,
                                                                  int32_t *__restrict__ mine_st, const Remap r, const uint64_t n_cells) {
  constexpr int R = 16 / sizeof(T);
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t * R >= n_cells) return;
  const bool def_nan = r.def_nan != 0, his_nan = r.src_def_nan != 0;
  const int last = r.nd - 1;
  // offset of innermost digit `dg` in mine (-1: no such item here)
  auto inner_off = [&](uint32_t dg) -> int64_t { return r.tab_off[last] < 0 ? (int64_t)((uint64_t)dg * r.stride[last]) : r.tab[r.tab_off[last] + dg]; };
  // decodes cell c: offset contributed by the outer dimensions (-1: an item mine lacks) and the innermost digit
  auto decode = [&](IDX c, uint32_t *digit0) -> int64_t {
    uint64_t dst = 0;
    bool ok = true;
    const IDX q0 = c / (IDX)r.len[last];
    *digit0 = (uint32_t)(c - q0 * (IDX)r.len[last]);
    c = q0;
#pragma unroll
    for (int d = kMaxDims - 2; d >= 0; --d) {
      if (d < last) {
        const IDX q = c / (IDX)r.len[d];
        const uint32_t digit = (uint32_t)(c - q * (IDX)r.len[d]);
        c = q;
        if (r.tab_off[d] < 0) {
          dst += (uint64_t)digit * r.stride[d];
        } else {
          const int64_t o = r.tab[r.tab_off[d] + digit];
          if (o < 0) ok = false;
          else dst += (uint64_t)o;
        }
      }
    }
    return ok ? (int64_t)dst : -1;
  };
  const uint64_t c0 = t * R;
  const bool whole = c0 + R <= n_cells;
  Vec<T, R> x;
  Vec<int32_t, R> xs;
  if (whole) {
    x = load_stream<T, R>(his + c0);
    if constexpr (HAS_STATUS) xs = load_stream<int32_t, R>(his_st + c0);
  } else {
#pragma unroll
    for (int e = 0; e < R; ++e) {
      x.v[e] = c0 + e < n_cells ? his[c0 + e] : T(0);
      if constexpr (HAS_STATUS) xs.v[e] = c0 + e < n_cells ? his_st[c0 + e] : 0;
    }
  }
  uint32_t digit0 = 0;
  int64_t outer = decode((IDX)c0, &digit0);
#pragma unroll
  for (int e = 0; e < R; ++e) {
    if (c0 + e >= n_cells) break;
    if (e > 0 && ++digit0 >= r.len[last]) outer = decode((IDX)(c0 + e), &digit0);  // the next innermost row
    const int64_t in = inner_off(digit0);
    if (outer < 0 || in < 0) continue;  // an item this store does not have: its cells are not loaded
    const uint64_t dst = (uint64_t)outer + (uint64_t)in;
    const bool his_set = cell_is_set<T>(x.v[e], HAS_STATUS ? xs.v[e] : OLAP_STATUS_SET, HAS_STATUS, his_nan);
    bool set;
    T v;
    if (his_set) {
      v = x.v[e];
      set = !Cell<T>::is_default(v, def_nan);
    } else {  // his default against mine (see load_scatter_kernel)
      v = Cell<T>::default_value(his_nan);
      constexpr bool is_float = (Cell<T>::dtype == OLAP_FLOAT32 || Cell<T>::dtype == OLAP_FLOAT64);
      set = his_nan ? (is_float && !def_nan) : def_nan;
    }
    mine[dst] = set ? v : Cell<T>::default_value(def_nan);
    if (mine_st) mine_st[dst] = set ? OLAP_STATUS_SET : 0;
  }
}

// The innermost dimension's items PERMUTED (the same items in another order — hydrating from a cube whose last
// dimension lists them differently) and every other dimension left alone: his row r is my row r with its cells
// rearranged.  A workgroup stages a tile of whole rows: 16-byte streaming loads of HIS cells, every cell written to
// its place in MY row inside LDS, and the tile leaves as 16-byte streaming stores — where the scatter forms above
// issue one 4-byte store per cell ([10]^8, last dimension permuted: 272 us).  load() writes EVERY cell of his
// (in-memory.js:152-158 iterates the other store densely), so a permutation covers every cell of my rows and nothing
// of mine has to be read.
struct LoadPermute {
  uint64_t n_rows;         // rows of `len` cells, the same in both stores
  uint32_t len;            // cells per row (<= kTileBytes / sizeof(T))
  uint32_t rows_per_tile;  // whole rows in kTileBytes
  SmallDiv by_len;         // cell of the tile -> its row
  const uint32_t *perm;    // device, [len]: where HIS j-th cell of a row sits in MY row
  int def_nan, src_def_nan;
};

template <typename T, bool HAS_STATUS>
__global__ __launch_bounds__(kBlock) void load_permute_rows_kernel(const T *__restrict__ his, const int32_t *__restrict__ his_st, T *__restrict__ mine,
                                                                   int32_t *__restrict__ mine_st, const LoadPermute a) {
  constexpr uint32_t V = 16 / sizeof(T);
  constexpr uint32_t kCells = kTileBytes / sizeof(T);
  constexpr int NL = kCells / V / kBlock;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *tile = reinterpret_cast<T *>(lds_raw);
  int32_t *stile = reinterpret_cast<int32_t *>(lds_raw + kTileBytes);                              // only with mine_st
  uint32_t *l_perm = reinterpret_cast<uint32_t *>(lds_raw + kTileBytes + (mine_st ? kCells * 4 : 0));  // [len]
  const uint64_t row0 = (uint64_t)xcd_contiguous(blockIdx.x, gridDim.x) * a.rows_per_tile;
  const uint64_t left = a.n_rows - row0;
  const uint32_t rows = left < a.rows_per_tile ? (uint32_t)left : a.rows_per_tile;
  const uint32_t n = rows * a.len;  // cells of this tile
  const uint64_t base = row0 * a.len;
  const bool def_nan = a.def_nan != 0, his_nan = a.src_def_nan != 0;
  Vec<T, (int)V> v[NL];
  Vec<int32_t, (int)V> sv[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t c = (threadIdx.x + (uint32_t)u * kBlock) * V;
    if (c + V <= n) {
      v[u] = load_stream_cell_aligned<T, (int)V>(his + base + c);
      if constexpr (HAS_STATUS) sv[u] = load_stream_cell_aligned<int32_t, (int)V>(his_st + base + c);
    } else {
#pragma unroll
      for (uint32_t e = 0; e < V; ++e) {
        v[u].v[e] = c + e < n ? his[base + c + e] : T(0);
        if constexpr (HAS_STATUS) sv[u].v[e] = c + e < n ? his_st[base + c + e] : 0;
      }
    }
  }
  for (uint32_t i = threadIdx.x; i < a.len; i += kBlock) l_perm[i] = a.perm[i];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t c0 = (threadIdx.x + (uint32_t)u * kBlock) * V;
#pragma unroll
    for (uint32_t e = 0; e < V; ++e) {
      const uint32_t c = c0 + e;
      if (c < n) {
        const uint32_t t = small_div(c, a.by_len);
        const uint32_t dst = t * a.len + l_perm[c - t * a.len];
        const T x = v[u].v[e];
        const bool his_set = cell_is_set<T>(x, HAS_STATUS ? sv[u].v[e] : OLAP_STATUS_SET, HAS_STATUS, his_nan);
        bool set;
        T val;
        if (his_set) {
          val = x;
          set = !Cell<T>::is_default(val, def_nan);
        } else {  // his default against mine (see load_scatter_kernel)
          val = Cell<T>::default_value(his_nan);
          constexpr bool is_float = (Cell<T>::dtype == OLAP_FLOAT32 || Cell<T>::dtype == OLAP_FLOAT64);
          set = his_nan ? (is_float && !def_nan) : def_nan;
        }
        tile[dst] = set ? val : Cell<T>::default_value(def_nan);
        if (mine_st) stile[dst] = set ? OLAP_STATUS_SET : 0;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const uint32_t c = (threadIdx.x + (uint32_t)u * kBlock) * V;
    if (c + V <= n) {
      store_stream_cell_aligned<T, (int)V>(mine + base + c, *reinterpret_cast<const Vec<T, (int)V> *>(tile + c));
      if (mine_st) store_stream_cell_aligned<int32_t, (int)V>(mine_st + base + c, *reinterpret_cast<const Vec<int32_t, (int)V> *>(stile + c));
    } else {
#pragma unroll
      for (uint32_t e = 0; e < V; ++e)
        if (c + e < n) {
          mine[base + c + e] = tile[c + e];
          if (mine_st) mine_st[base + c + e] = stile[c + e];
        }
    }
  }
}

// ======================================================================= K4: reorder as a brick transpose
// in-memory.js:178-211.  When the output's fastest dimension is not the input's fastest one a plain
// gather reads 4 B per cache line.  Here a workgroup owns a BRICK: a small range of every dimension
// chosen so that the brick is >= 64 cells long both along the input's fastest dimensions and along
// the output's fastest ones.  The brick is read in input order (coalesced), parked in LDS (padded
// against bank conflicts) and written in output order (coalesced).
struct Brick {
  int nd;                        // collapsed dims, output (new) order
  uint32_t len[kMaxDims];        // extent of each dim
  uint32_t chunk[kMaxDims];      // brick extent of each dim (1 for dims outside the brick)
  uint32_t nblk[kMaxDims];       // ceil(len / chunk)
  uint64_t in_stride[kMaxDims];  // stride in the source
  uint64_t out_stride[kMaxDims]; // stride in the destination
  int n_act;                     // dims with chunk > 1 (<= 4: their digits are packed in 4 bytes)
  int32_t act_dim[4];            // the active dims
  uint32_t elems;                // cells per brick
  int ragged;                    // some chunk does not divide its dim: edge bricks are partial
  int def_nan;
  // All bricks are congruent, so the offsets of a brick's cells relative to its origin are tables
  // built once by the plan (L2-resident, read coalesced): e = position in read (source) order,
  // f = position in write (destination) order.
  const uint32_t *rd_off;   // [elems] source offset of cell e
  const uint32_t *wr_off;   // [elems] destination offset of cell f
  const uint32_t *wr_lds;   // [elems] read-order position e of cell f (where it sits in LDS)
  const uint32_t *rd_dig;   // [elems] digits of cell e in the active dims, one byte each
  const uint32_t *wr_dig;   // [elems] same for cell f
  // 16-byte form (reorder_brick4_kernel): every source run and every destination run of a brick is
  // a whole number of aligned 4-cell groups and no brick is ragged
  int quad;
  const uint32_t *rd_off4;  // [elems/4] source offset of read group q (cells 4q..4q+3 in read order)
  const uint32_t *wr_off4;  // [elems/4] destination offset of write group f
  const uint2 *wr_pos4;     // [elems/4] padded LDS positions of the 4 cells of write group f, 16 bits each
};

__device__ __forceinline__ uint32_t lds_pad(uint32_t e) { return e + (e >> 5); }

template <typename T, bool HAS_STATUS>
__global__ __launch_bounds__(1024) void reorder_brick_kernel(const T *__restrict__ in,
                                                               const int32_t *__restrict__ st_in,
                                                               T *__restrict__ out,
                                                               int32_t *__restrict__ st_out, const Brick b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *tile = reinterpret_cast<T *>(lds_raw);
  const size_t tile_bytes = ((size_t)lds_pad(b.elems) * sizeof(T) + 15) & ~(size_t)15;
  int32_t *stile = reinterpret_cast<int32_t *>(lds_raw + tile_bytes);

  // brick origin: wave-uniform scalar arithmetic on blockIdx
  uint64_t c = blockIdx.x, base_in = 0, base_out = 0;
  // per active dim: cells of this brick inside the cube, one byte each (unused bytes: always inside)
  uint32_t lim = b.n_act >= 4 ? 0u : (0x7F7F7F7Fu << (8 * b.n_act));
  bool partial = false;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < b.nd) {
      const uint32_t origin = (uint32_t)(c % b.nblk[d]) * b.chunk[d];
      c /= b.nblk[d];
      base_in += (uint64_t)origin * b.in_stride[d];
      base_out += (uint64_t)origin * b.out_stride[d];
      if (b.ragged) {
        const uint32_t left = b.len[d] - origin;
        const uint32_t l = left < b.chunk[d] ? left : b.chunk[d];
        partial = partial || l < b.chunk[d];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < b.n_act && b.act_dim[k] == d) lim |= l << (8 * k);
      }
    }
  }
  // a cell is inside iff each digit byte is below its limit byte: SWAR compare of 4 bytes
  auto inside = [&](uint32_t dig) {
    // bytes are < 128 (chunk <= 127), so (dig | 0x80808080) - lim keeps bit 7 of a byte iff dig >= lim there
    return (((dig | 0x80808080u) - lim) & 0x80808080u) == 0u;
  };
  const bool def_nan = b.def_nan != 0;
  const T *src = in + base_in;
  const int32_t *ssrc = HAS_STATUS ? st_in + base_in : nullptr;
  // both phases are latency chains (table entry -> address -> data), so each lane keeps UB of them
  // in flight
  constexpr int UB = 8;
  const uint32_t nthreads = blockDim.x;  // 256 for small bricks, 1024 for large ones (occupancy)
  for (uint32_t e0 = threadIdx.x; e0 < b.elems; e0 += nthreads * UB) {
    uint32_t off[UB];
    bool ok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t e = e0 + u * nthreads;
      ok[u] = e < b.elems;
      if (ok[u] && partial) ok[u] = inside(b.rd_dig[e]);
      off[u] = ok[u] ? b.rd_off[e] : 0u;
    }
    T x[UB];
    int32_t sx[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (ok[u]) {
        x[u] = src[off[u]];
        if constexpr (HAS_STATUS) sx[u] = ssrc[off[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (ok[u]) {
        const uint32_t e = e0 + u * nthreads;
        tile[lds_pad(e)] = x[u];
        if constexpr (HAS_STATUS) stile[lds_pad(e)] = sx[u];
      }
    }
  }
  __syncthreads();
  T *dst = out + base_out;
  int32_t *sdst = st_out ? st_out + base_out : nullptr;
  for (uint32_t f0 = threadIdx.x; f0 < b.elems; f0 += nthreads * UB) {
    uint32_t off[UB], pos[UB];
    bool ok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t f = f0 + u * nthreads;
      ok[u] = f < b.elems;
      if (ok[u] && partial) ok[u] = inside(b.wr_dig[f]);
      off[u] = ok[u] ? b.wr_off[f] : 0u;
      pos[u] = ok[u] ? b.wr_lds[f] : 0u;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (ok[u]) {
        const T x = tile[lds_pad(pos[u])];
        const bool set = cell_is_set<T>(x, HAS_STATUS ? stile[lds_pad(pos[u])] : OLAP_STATUS_SET, HAS_STATUS, def_nan);
        dst[off[u]] = set ? x : Cell<T>::default_value(def_nan);
        if (sdst) sdst[off[u]] = set ? OLAP_STATUS_SET : 0;
      }
    }
  }
}

// 16-byte form.  Cells 4q..4q+3 of the read order are adjacent in the source (one 16 B load, one
// 16 B LDS store); a write group is 4 adjacent destination cells whose LDS positions come packed
// from the table.  LDS position of read-order cell e: e + 4*(e/32) (keeps groups 16 B aligned and
// rotates the banks every 32 cells).
__device__ __forceinline__ uint32_t lds_pad4(uint32_t e) { return e + ((e >> 5) << 2); }

template <typename T, bool HAS_STATUS, int UB>
__global__ __launch_bounds__(1024) void reorder_brick4_kernel(const T *__restrict__ in,
                                                                const int32_t *__restrict__ st_in,
                                                                T *__restrict__ out,
                                                                int32_t *__restrict__ st_out, const Brick b) {
  static_assert(sizeof(T) == 4, "16-byte groups of 4 cells");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *tile = reinterpret_cast<T *>(lds_raw);
  int32_t *stile = reinterpret_cast<int32_t *>(lds_raw + (size_t)lds_pad4(b.elems) * 4);
  uint64_t c = xcd_contiguous(blockIdx.x, gridDim.x), base_in = 0, base_out = 0;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < b.nd) {
      const uint32_t origin = (uint32_t)(c % b.nblk[d]) * b.chunk[d];
      c /= b.nblk[d];
      base_in += (uint64_t)origin * b.in_stride[d];
      base_out += (uint64_t)origin * b.out_stride[d];
    }
  }
  const bool def_nan = b.def_nan != 0;
  const T *src = in + base_in;
  const int32_t *ssrc = HAS_STATUS ? st_in + base_in : nullptr;
  const uint32_t nq = b.elems >> 2;
  const uint32_t nthreads = blockDim.x;  // 512: a 10^4-cell brick leaves room for 3 workgroups per CU, so each must bring many waves
  for (uint32_t q0 = threadIdx.x; q0 < nq; q0 += nthreads * UB) {
    uint32_t off[UB];
    Vec<T, 4> x[UB];
    Vec<int32_t, 4> sx[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t q = q0 + u * nthreads;
      off[u] = q < nq ? b.rd_off4[q] : 0u;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (q0 + u * nthreads < nq) {
        x[u] = load_stream<T, 4>(src + off[u]);
        if constexpr (HAS_STATUS) sx[u] = load_stream<int32_t, 4>(ssrc + off[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t q = q0 + u * nthreads;
      if (q < nq) {
        *reinterpret_cast<Vec<T, 4> *>(tile + lds_pad4(q * 4)) = x[u];
        if constexpr (HAS_STATUS) *reinterpret_cast<Vec<int32_t, 4> *>(stile + lds_pad4(q * 4)) = sx[u];
      }
    }
  }
  __syncthreads();
  T *dst = out + base_out;
  int32_t *sdst = st_out ? st_out + base_out : nullptr;
  for (uint32_t f0 = threadIdx.x; f0 < nq; f0 += nthreads * UB) {
    uint32_t off[UB];
    uint2 pk[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const uint32_t f = f0 + u * nthreads;
      off[u] = f < nq ? b.wr_off4[f] : 0u;
      pk[u] = f < nq ? b.wr_pos4[f] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (f0 + u * nthreads < nq) {
        const uint32_t pos[4] = {pk[u].x & 0xFFFFu, pk[u].x >> 16, pk[u].y & 0xFFFFu, pk[u].y >> 16};
        Vec<T, 4> ov;
        Vec<int32_t, 4> os;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const T x = tile[pos[e]];
          const bool set = cell_is_set<T>(x, HAS_STATUS ? stile[pos[e]] : OLAP_STATUS_SET, HAS_STATUS, def_nan);
          ov.v[e] = set ? x : Cell<T>::default_value(def_nan);
          os.v[e] = set ? OLAP_STATUS_SET : 0;
        }
        store_stream<T, 4>(dst + off[u], ov);
        if (sdst) store_stream<int32_t, 4>(sdst + off[u], os);
      }
    }
  }
}

// dice of ONE dimension over rows that are not whole 16-byte groups, any row length of at least one
// group (cubes with odd extents).  The destination [outer, k_new, inner] is one contiguous run, so a
// lane owns ALIGNED 16-byte groups of it and fetches each from wherever its cells lie in the source
// with ONE 16-byte load at a cell-aligned address (load_stream_cell_aligned: gfx950 takes it at no
// measurable cost) — no staging of cells through LDS, U groups in flight per lane, every store a
// whole aligned group.  Integer work is what this form has to watch (v_mul_lo/hi_u32 are quarter rate:
// a first version that decoded (outer, item) and formed the 64-bit source offset per group, four
// times over for a straddling one, measured 89 us where the bytes take 45): the workgroup decodes its
// first cell once with 64-bit divisions, its first threads put the source offset of each of the few
// destination rows it touches into LDS, and a lane gets its row from SmallDiv (a compare or one
// mul-hi) and its source cell with one LDS read and an add.
struct DiceRows {
  uint64_t outer, k_old, k_new, inner;  // source [outer, k_old, inner] -> destination [outer, k_new, inner]
  const int32_t *sel;                   // device, [k_new]: the old item of each new item, -1: a row of defaults
  int def_nan;
};

struct DiceDirect {
  DiceRows p;
  SmallDiv by_inner, by_k;
  uint32_t rows;  // destination rows a workgroup can touch, plus the one a last group runs into
};

constexpr int kDiceDirectGroups = 4;  // 16-byte groups in flight per lane

// whether dice_direct_kernel takes this shape (the plan asks before choosing it)
inline int dice_direct_groups() {
  static const int u = [] {
    const char *e = getenv("OLAP_DICE_DIRECT_GROUPS");
    const int v = e ? atoi(e) : kDiceDirectGroups;
    return v == 1 || v == 2 || v == 8 ? v : 4;
  }();
  return u;
}

template <typename T>
inline bool dice_direct_fits(const DiceRows &p, DiceDirect *out) {
  constexpr uint64_t V = 16 / sizeof(T);
  const uint64_t SPAN = (uint64_t)dice_direct_groups() * kBlock * V;
  if (p.inner < V || p.inner >= (1ull << 31) || p.k_new == 0 || p.k_new > 0xFFFFFFFFull || p.k_old > 0xFFFFFFFFull) return false;
  DiceDirect d;
  d.p = p;
  const uint64_t rel_max = p.inner - 1 + SPAN;  // first cell of a lane's group, counted from the workgroup's first row
  const uint64_t rows_max = rel_max / p.inner;
  d.rows = (uint32_t)(rows_max + 2);
  if (!small_div_for(p.inner, rel_max, &d.by_inner) || !small_div_for(p.k_new, p.k_new - 1 + d.rows, &d.by_k)) return false;
  const long double cells = (long double)p.outer * (long double)p.k_new * (long double)p.inner;
  if (cells / (long double)SPAN >= 2147483000.0L) return false;
  if (out) *out = d;
  return true;
}

template <typename T, bool HAS_STATUS, int U>
__global__ __launch_bounds__(kBlock) void dice_direct_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in,
                                                             T *__restrict__ out, int32_t *__restrict__ st_out,
                                                             const DiceDirect a) {
  constexpr uint32_t V = 16 / sizeof(T);
  constexpr uint32_t SPAN = U * kBlock * V;  // destination cells per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  int64_t *row_src = reinterpret_cast<int64_t *>(lds_raw);  // [a.rows]: first source cell of the workgroup's t-th destination row, -1: default
  const uint32_t inner = (uint32_t)a.p.inner, k_new = (uint32_t)a.p.k_new;
  const uint64_t n_out = a.p.outer * a.p.k_new * a.p.inner;
  const uint64_t c_base = (uint64_t)xcd_contiguous(blockIdx.x, gridDim.x) * SPAN;
  const uint64_t p_base = c_base / a.p.inner;
  const uint32_t i_base = (uint32_t)(c_base - p_base * a.p.inner);
  const uint64_t o_base = p_base / a.p.k_new;
  const uint32_t j_base = (uint32_t)(p_base - o_base * a.p.k_new);
  for (uint32_t t = threadIdx.x; t < a.rows; t += kBlock) {
    const uint32_t jr = j_base + t;
    const uint32_t dq = small_div(jr, a.by_k);
    const uint32_t j = jr - dq * k_new;
    const uint64_t o = o_base + dq;
    int64_t src = -1;
    if (o < a.p.outer) {
      const int32_t sj = a.p.sel[j];
      if (sj >= 0) src = (int64_t)((o * a.p.k_old + (uint64_t)sj) * a.p.inner);
    }
    row_src[t] = src;
  }
  __syncthreads();
  const bool def_nan = a.p.def_nan != 0;
  const T dflt = Cell<T>::default_value(def_nan);
  // A group that runs over the end of its destination row (inner >= V: into the next row at most) takes its first
  // m cells from the LAST whole group of its source row and the others from the FIRST whole group of the next
  // row's source: two whole-group loads into registers of their own, rotated into place once everything has
  // arrived.  (Fetching those cells one by one in a divergent branch made the compiler wait for each group's load
  // before issuing the next — the same registers were written on both paths — and left ONE load in flight per lane.)
  Vec<T, (int)V> xv[U], yv[U];
  Vec<int32_t, (int)V> xs[U], ys[U];
  uint32_t mm[U];  // cells of group u that belong to its first row (>= V: all of them)
  bool live_x[U], live_y[U];  // that row / the next one has a source (else: a row of defaults)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t off = ((uint32_t)u * kBlock + threadIdx.x) * V;
    const uint64_t c0 = c_base + off;
#pragma unroll
    for (uint32_t e = 0; e < V; ++e) {
      xv[u].v[e] = dflt;
      yv[u].v[e] = dflt;
      xs[u].v[e] = 0;
      ys[u].v[e] = 0;
    }
    mm[u] = V;
    live_x[u] = live_y[u] = false;
    if (c0 >= n_out) continue;
    const uint32_t rel = i_base + off;
    const uint32_t t = small_div(rel, a.by_inner);
    const uint32_t i0 = rel - t * inner;
    const uint32_t m = inner - i0;  // cells left in this destination row
    mm[u] = m < V ? m : V;
    const int64_t src = row_src[t];
    if (src >= 0) {
      live_x[u] = true;
      const int64_t at = src + (m >= V ? i0 : inner - V);
      xv[u] = load_stream_cell_aligned<T, (int)V>(in + at);
      if constexpr (HAS_STATUS) xs[u] = load_stream_cell_aligned<int32_t, (int)V>(st_in + at);
    }
    if (m < V) {
      const int64_t next = row_src[t + 1];  // -1 past the end of the cube
      if (next >= 0) {
        live_y[u] = true;
        yv[u] = load_stream_cell_aligned<T, (int)V>(in + next);
        if constexpr (HAS_STATUS) ys[u] = load_stream_cell_aligned<int32_t, (int)V>(st_in + next);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t c0 = c_base + ((uint64_t)u * kBlock + threadIdx.x) * V;
    if (c0 >= n_out) continue;
    Vec<T, (int)V> ov;
    Vec<int32_t, (int)V> os;
#pragma unroll
    for (uint32_t e = 0; e < V; ++e) {
      T x = xv[u].v[e];
      int32_t sx = HAS_STATUS ? xs[u].v[e] : OLAP_STATUS_SET;
#pragma unroll
      for (uint32_t m = 1; m < V; ++m) {  // the group holds m cells of its first row
        const bool is = mm[u] == m;
        if (e < m) {
          x = is ? xv[u].v[V - m + e] : x;
          if constexpr (HAS_STATUS) sx = is ? xs[u].v[V - m + e] : sx;
        } else {
          x = is ? yv[u].v[e - m] : x;
          if constexpr (HAS_STATUS) sx = is ? ys[u].v[e - m] : sx;
        }
      }
      if (e == V - 1) {
        // (never taken: mm >= 1.  It keeps the last cell of the second load alive — with that register free the
        // allocator reuses it for the next group's scalars and the compiler waits for the load right after issuing it)
        x = mm[u] == 0 ? yv[u].v[V - 1] : x;
        if constexpr (HAS_STATUS) sx = mm[u] == 0 ? ys[u].v[V - 1] : sx;
      }
      const bool set = (e < mm[u] ? live_x[u] : live_y[u]) && cell_is_set<T>(x, sx, HAS_STATUS, def_nan);
      ov.v[e] = set ? x : dflt;
      os.v[e] = set ? OLAP_STATUS_SET : 0;
    }
    if (c0 + V <= n_out) {
      store_stream<T, (int)V>(out + c0, ov);
      if (st_out) store_stream<int32_t, (int)V>(st_out + c0, os);
    } else {
#pragma unroll
      for (uint32_t e = 0; e < V; ++e)
        if (c0 + e < n_out) {
          out[c0 + e] = ov.v[e];
          if (st_out) st_out[c0 + e] = os.v[e];
        }
    }
  }
}

// ======================================================================= K3: drillDown
// in-memory.js:336-430.  One lane per NEW cell: parent offset, sibling count n and this child's
// ordinal c (its rank among the parent's children in ascending new index) come from per-dim
// tables; the integer remainder spreading (:403-417) is replayed in float64 exactly.
struct DrillDown {
  int nd;
  uint32_t new_len[kMaxDims];
  uint64_t old_stride[kMaxDims];
  int32_t tab_off[kMaxDims];  // -1: identity dim (parent digit = digit, 1 child); else offset into tab
  const uint32_t *tab;        // per table dim: parent[len], count[len], rank[len]
  uint64_t total;             // new cells
  int def_nan;
  int method;
  int use_rounding;
  const double *dist;  // device, or nullptr
  uint64_t n_dist;
  double added_len;    // newSize / oldSize            (:392)
  double chunk;        // newSize / sharedDimSize      (:395)
  unsigned long long *err;  // device: smallest failing new index (init ~0ull)
};

template <typename T, bool HAS_STATUS>
__global__ __launch_bounds__(kBlock) void drilldown_kernel(const T *__restrict__ in,
                                                           const int32_t *__restrict__ st_in,
                                                           T *__restrict__ out,
                                                           int32_t *__restrict__ st_out,
                                                           const DrillDown a) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.total) return;
  const bool def_nan = a.def_nan != 0;
  uint64_t c = t, parent = 0, n = 1, ordinal = 0;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < a.nd) {
      const uint32_t L = a.new_len[d];
      const uint32_t digit = (uint32_t)(c % L);
      c /= L;
      if (a.tab_off[d] < 0) {
        parent += (uint64_t)digit * a.old_stride[d];
      } else {
        const uint32_t *tb = a.tab + a.tab_off[d];
        parent += (uint64_t)tb[digit] * a.old_stride[d];
        const uint64_t cnt = tb[L + digit];
        ordinal += (uint64_t)tb[2 * L + digit] * n;  // dims to the right vary fastest
        n *= cnt;
      }
    }
  }
  const T x = in[parent];
  const int32_t sx = HAS_STATUS ? st_in[parent] : OLAP_STATUS_SET;
  const double old_value = Cell<T>::to_f64(x);
  // `if (!oldValue) continue` (:386-387): unset, 0, -0 and NaN are all skipped
  bool has = cell_is_set<T>(x, sx, HAS_STATUS, def_nan) && old_value == old_value && old_value != 0.0;
  double r = 0.0;
  if (has) {
    const double nn = (double)n;
    if (a.dist) {  // :391-400
      const double di = floor((double)t / a.chunk) * a.added_len + fmod((double)t, a.added_len);
      const bool okay = di >= 0.0 && di < (double)a.n_dist && di == floor(di) && a.dist[(uint64_t)di] == a.dist[(uint64_t)di];
      if (!okay) {
        atomicMin(a.err, (unsigned long long)t);
        has = false;
      } else {
        r = old_value * a.dist[(uint64_t)di];
      }
    } else if (a.method == OLAP_SUM) {
      if (a.use_rounding) {  // :403-417
        const double value = floor(old_value / nn);
        const double remainder = fmod(old_value, nn);
        const double cid = (double)ordinal;
        const double one_over = remainder / nn;
        const bool last_is_same = floor(cid * one_over) == floor((cid - 1.0) * one_over);
        r = last_is_same ? floor(value) : floor(value) + 1.0;
      } else {
        r = old_value / nn;  // :419
      }
    } else {
      r = old_value;  // :422
    }
    if (has && is_default_f64(r, def_nan)) has = false;  // setValue
  }
  T ov;
  int32_t os;
  emit_cell<T>(r, has, def_nan, ov, os);
  out[t] = ov;
  if (st_out) st_out[t] = os;
}

// One refined axis (what Cube.drillDown / addDimension ask for, src/cube.js:971, :919-927), no
// distributions: the mirror image of drillup_rows_kernel.  View [outer, G, inner] -> [outer, K, inner];
// a workgroup owns 256 adjacent VEC-wide slots of ONE parent row (outer, g), reads them once, and
// streams the children's values to each of the group's rows — every parent cell is read once and
// every store is a full 16 B per lane.  n = number of children (uniform per workgroup); the integer
// remainder spreading needs only the child's ordinal, which is its position in the group's list.
constexpr uint32_t kChildrenPerBlock = 8;  // children rows written by one workgroup (measured: 4 is latency-bound, 16 no better)

template <typename T, bool HAS_STATUS, int VEC>
__global__ __launch_bounds__(kBlock) void drilldown_rows_kernel(const T *__restrict__ in,
                                                                const int32_t *__restrict__ st_in,
                                                                T *__restrict__ out,
                                                                int32_t *__restrict__ st_out,
                                                                const DrillUpAxis a, int divide, int use_rounding,
                                                                uint32_t segments) {
  // blockIdx.x = ((og * segments) + seg) * blocks_per_row + chunk   (uniform math)
  const uint32_t bpr = (uint32_t)a.blocks_per_row;
  const uint32_t bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const uint64_t ogs = bid / bpr;
  const uint32_t chunk = bid - (uint32_t)ogs * bpr;
  const uint32_t seg = (uint32_t)(ogs % segments);
  const uint64_t og = ogs / segments;
  const uint64_t g = og % a.G;
  const uint64_t o = og / a.G;
  const uint32_t gbeg = a.gstart[g], gend = a.gstart[g + 1];
  const uint32_t jbeg = gbeg + seg * kChildrenPerBlock;
  if (jbeg >= gend) return;  // this group has fewer children than the longest one (whole workgroup leaves)
  const uint32_t jend = jbeg + kChildrenPerBlock < gend ? jbeg + kChildrenPerBlock : gend;
  const uint64_t iv = (uint64_t)chunk * kBlock + threadIdx.x;
  if (iv >= a.n_vec) return;
  const uint64_t i0 = iv * VEC;
  const bool def_nan = a.def_nan != 0;
  const uint64_t pidx = (o * a.G + g) * a.inner + i0;
  const Vec<T, VEC> pv = load_vec<T, VEC>(in + pidx);
  Vec<int32_t, VEC> ps;
  if constexpr (HAS_STATUS) ps = load_vec<int32_t, VEC>(st_in + pidx);
  const double n = (double)(gend - gbeg);
  double base[VEC], one_over[VEC];
  bool has[VEC];
  Vec<T, VEC> same_v;     // the value every child receives when no remainder is spread
  Vec<int32_t, VEC> same_s;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const double old_value = Cell<T>::to_f64(pv.v[e]);
    // `if (!oldValue) continue` (in-memory.js:386-387): unset, 0, -0 and NaN parents give nothing
    has[e] = cell_is_set<T>(pv.v[e], HAS_STATUS ? ps.v[e] : OLAP_STATUS_SET, HAS_STATUS, def_nan) && old_value == old_value &&
             old_value != 0.0;
    base[e] = divide ? (use_rounding ? floor(floor(old_value / n)) : old_value / n) : old_value;  // :404, :419, :422
    one_over[e] = fmod(old_value, n) / n;                                                          // :405-407
    emit_cell<T>(base[e], has[e] && !is_default_f64(base[e], def_nan), def_nan, same_v.v[e], same_s.v[e]);
  }
  T *orow = out + (o * a.K) * a.inner + i0;
  int32_t *srow = st_out ? st_out + (o * a.K) * a.inner + i0 : nullptr;
  const bool spread = divide && use_rounding;
  for (uint32_t j = jbeg; j < jend; ++j) {
    const uint64_t k = a.order ? (uint64_t)a.order[j] : (uint64_t)j;
    Vec<T, VEC> ov = same_v;
    Vec<int32_t, VEC> os = same_s;
    if (spread) {  // :403-417, replayed with the same float64 operations; cid = the child's ordinal
      const double cid = (double)(j - gbeg);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool last_is_same = floor(cid * one_over[e]) == floor((cid - 1.0) * one_over[e]);
        const double r = last_is_same ? base[e] : base[e] + 1.0;
        emit_cell<T>(r, has[e] && !is_default_f64(r, def_nan), def_nan, ov.v[e], os.v[e]);
      }
    }
    store_stream<T, VEC>(orow + k * a.inner, ov);
    if (srow) store_stream<int32_t, VEC>(srow + k * a.inner, os);
  }
}

// The same, for rows that do not start on a 128-byte line (inner * sizeof(T) % 128 != 0) and no
// remainder spreading: the children's rows are then misaligned by a different amount each, and a
// workgroup that stores "its" 256 slots writes partial lines at both ends of every wave (measured:
// 75-80 us against 61-65 us for the line-aligned neighbour shape, 1e8 float32 cells).  Here the
// child values of the parent window are computed once into LDS and every child row is stored from a
// window shifted by that row's own misalignment, so each wave stores whole lines; only the two ends
// of a row are partial.
// ANY: rows are not even whole 16-byte groups (inner % VEC != 0, the usual case for cubes with odd
// extents): the parent window is staged cell by cell, a child row's window starts at any cell
// offset (4 LDS reads per group instead of one) and the groups that straddle a row's ends are
// stored cell by cell — everything in between still leaves as aligned 16-byte stores.
template <typename T, bool HAS_STATUS, int VEC, bool ANY>
__global__ __launch_bounds__(kBlock) void drilldown_rows_lines_kernel(const T *__restrict__ in,
                                                                      const int32_t *__restrict__ st_in,
                                                                      T *__restrict__ out,
                                                                      int32_t *__restrict__ st_out,
                                                                      const DrillUpAxis a, int divide, uint32_t segments,
                                                                      uint32_t bpr) {
  constexpr uint32_t LINE = 128 / sizeof(T);  // cells per line
  constexpr uint32_t CH = kBlock * VEC;       // cells a workgroup stores per child row
  constexpr uint32_t SLOTS = (CH + LINE) / VEC;
  __shared__ alignas(16) T lv[CH + LINE];
  __shared__ alignas(16) int32_t ls[CH + LINE];
  const uint32_t bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const uint64_t ogs = bid / bpr;
  const uint32_t chunk = bid - (uint32_t)ogs * bpr;
  const uint32_t seg = (uint32_t)(ogs % segments);
  const uint64_t og = ogs / segments;
  const uint64_t g = og % a.G;
  const uint64_t o = og / a.G;
  const uint32_t gbeg = a.gstart[g], gend = a.gstart[g + 1];
  const uint32_t jbeg = gbeg + seg * kChildrenPerBlock;
  if (jbeg >= gend) return;  // whole workgroup
  const uint32_t jend = jbeg + kChildrenPerBlock < gend ? jbeg + kChildrenPerBlock : gend;
  const bool def_nan = a.def_nan != 0;
  const double n = (double)(gend - gbeg);
  // LDS holds the children's value for parent cells [w0, w0 + CH + LINE)
  const int64_t w0 = (int64_t)chunk * CH - (int64_t)LINE;
  const T *prow = in + (o * a.G + g) * a.inner;
  const int32_t *psrow = HAS_STATUS ? st_in + (o * a.G + g) * a.inner : nullptr;
  if constexpr (ANY) {
    // every lane's cells are fetched before any is used (unconditional loads, from the row's first cell where the
    // window has none: a load inside the range test is waited for before the next one is issued)
    constexpr uint32_t NC = (CH + LINE + kBlock - 1) / kBlock;
    T pv[NC];
    int32_t ps[NC];
#pragma unroll
    for (uint32_t q = 0; q < NC; ++q) {
      const uint32_t c = threadIdx.x + q * kBlock;
      const int64_t i = w0 + (int64_t)c;
      const int64_t at = c < CH + LINE && i >= 0 && i < (int64_t)a.inner ? i : 0;
      pv[q] = prow[at];
      ps[q] = HAS_STATUS ? psrow[at] : OLAP_STATUS_SET;
    }
#pragma unroll
    for (uint32_t q = 0; q < NC; ++q) {
      const uint32_t c = threadIdx.x + q * kBlock;
      const int64_t i = w0 + (int64_t)c;
      T ov = T(0);
      int32_t os = 0;
      if (i >= 0 && i < (int64_t)a.inner) {
        const double old_value = Cell<T>::to_f64(pv[q]);
        const bool has = cell_is_set<T>(pv[q], ps[q], HAS_STATUS, def_nan) && old_value == old_value && old_value != 0.0;  // in-memory.js:386-387
        const double r = divide ? old_value / n : old_value;  // :419, :422
        emit_cell<T>(r, has && !is_default_f64(r, def_nan), def_nan, ov, os);
      }
      if (c < CH + LINE) {
        lv[c] = ov;
        ls[c] = os;
      }
    }
  }
  for (uint32_t slot = threadIdx.x; !ANY && slot < SLOTS; slot += kBlock) {
    const int64_t i = w0 + (int64_t)slot * VEC;
    Vec<T, VEC> ov;
    Vec<int32_t, VEC> os;
    if (i >= 0 && i < (int64_t)a.inner) {
      const Vec<T, VEC> pv = load_vec<T, VEC>(prow + i);
      Vec<int32_t, VEC> ps;
      if constexpr (HAS_STATUS) ps = load_vec<int32_t, VEC>(psrow + i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const double old_value = Cell<T>::to_f64(pv.v[e]);
        const bool has = cell_is_set<T>(pv.v[e], HAS_STATUS ? ps.v[e] : OLAP_STATUS_SET, HAS_STATUS, def_nan) &&
                         old_value == old_value && old_value != 0.0;  // in-memory.js:386-387
        const double r = divide ? old_value / n : old_value;           // :419, :422
        emit_cell<T>(r, has && !is_default_f64(r, def_nan), def_nan, ov.v[e], os.v[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        ov.v[e] = T(0);
        os.v[e] = 0;
      }
    }
    *reinterpret_cast<Vec<T, VEC> *>(&lv[slot * VEC]) = ov;
    *reinterpret_cast<Vec<int32_t, VEC> *>(&ls[slot * VEC]) = os;
  }
  __syncthreads();
  for (uint32_t j = jbeg; j < jend; ++j) {
    const uint64_t k = a.order ? (uint64_t)a.order[j] : (uint64_t)j;
    T *row = out + (o * a.K + k) * a.inner;
    const uint32_t shift = (uint32_t)(((uintptr_t)row / sizeof(T)) % LINE);  // cells past the line start (a multiple of VEC unless ANY)
    const int64_t i = (int64_t)chunk * CH + (int64_t)threadIdx.x * VEC - (int64_t)shift;
    const uint32_t at = threadIdx.x * VEC + LINE - shift;  // = i - w0
    if constexpr (!ANY) {
      if (i < 0 || i >= (int64_t)a.inner) continue;
      store_stream<T, VEC>(row + i, *reinterpret_cast<const Vec<T, VEC> *>(&lv[at]));
      if (st_out) store_stream<int32_t, VEC>(st_out + (o * a.K + k) * a.inner + i, *reinterpret_cast<const Vec<int32_t, VEC> *>(&ls[at]));
    } else {
      if (i + (int64_t)VEC <= 0 || i >= (int64_t)a.inner) continue;
      Vec<T, VEC> ov;
      Vec<int32_t, VEC> os;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        ov.v[e] = lv[at + e];
        os.v[e] = ls[at + e];
      }
      int32_t *srow = st_out ? st_out + (o * a.K + k) * a.inner : nullptr;
      if (i >= 0 && i + (int64_t)VEC <= (int64_t)a.inner) {  // (row + i) is 16 B aligned by construction
        store_stream<T, VEC>(row + i, ov);
        if (srow) store_stream<int32_t, VEC>(srow + i, os);
      } else {  // the group straddles an end of the row
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (i + e >= 0 && i + e < (int64_t)a.inner) {
            row[i + e] = ov.v[e];
            if (srow) srow[i + e] = os.v[e];
          }
        }
      }
    }
  }
}

// Float cells, no distributions: every child of a parent receives the SAME value (old / n, or a
// copy), so drillDown is split into a tiny pass over the OLD cells (one float64 division per
// parent instead of one per child) and a pure broadcast of the quotients with gather_kernel, which
// runs at copy bandwidth with 16 B accesses.
struct DrillDownScale {
  int nd;
  uint32_t old_len[kMaxDims];
  int32_t tab_off[kMaxDims];  // -1: untouched dim (one child per cell); else child counts per old index
  const uint32_t *tab;        // device
  uint64_t total;             // old cells
  int def_nan;
  int divide;                 // method == 'sum'
};

template <typename T, bool HAS_STATUS>
__global__ __launch_bounds__(kBlock) void drilldown_scale_kernel(const T *__restrict__ in,
                                                                 const int32_t *__restrict__ st_in,
                                                                 T *__restrict__ q, const DrillDownScale a) {
  const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.total) return;
  const bool def_nan = a.def_nan != 0;
  uint64_t c = t, n = 1;
#pragma unroll
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < a.nd) {
      const uint32_t digit = (uint32_t)(c % a.old_len[d]);
      c /= a.old_len[d];
      if (a.tab_off[d] >= 0) n *= a.tab[a.tab_off[d] + digit];
    }
  }
  const T x = in[t];
  const double old_value = Cell<T>::to_f64(x);
  // `if (!oldValue) continue` (in-memory.js:386-387): unset, 0, -0 and NaN are all skipped
  const bool has = cell_is_set<T>(x, HAS_STATUS ? st_in[t] : OLAP_STATUS_SET, HAS_STATUS, def_nan) &&
                   old_value == old_value && old_value != 0.0;
  const double r = a.divide ? old_value / (double)n : old_value;  // :419 / :422
  T ov;
  int32_t os;
  emit_cell<T>(r, has && !is_default_f64(r, def_nan), def_nan, ov, os);
  q[t] = ov;
}

// ======================================================================= element-wise helpers
template <typename T>
__global__ __launch_bounds__(kBlock) void canonicalize_kernel(T *values, int32_t *status, uint64_t n, int def_nan_i,
                                                              int use_status) {
  const bool def_nan = def_nan_i != 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const T v = values[i];
    const bool set = cell_is_set<T>(v, use_status ? status[i] : OLAP_STATUS_SET, use_status != 0, def_nan);
    if (!set) values[i] = Cell<T>::default_value(def_nan);
    if (status) status[i] = set ? OLAP_STATUS_SET : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void from_f64_kernel(const double *src, T *values, int32_t *status, uint64_t n,
                                                          int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const double d = src[i];
    // setValue on the JS number first (a NaN under a NaN default is "unset" even for integer types)
    const bool has = !is_default_f64(d, def_nan);
    T ov;
    int32_t os;
    emit_cell<T>(d, has, def_nan, ov, os);
    values[i] = ov;
    if (status) status[i] = os;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void to_f64_kernel(const T *values, double *dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    dst[i] = Cell<T>::to_f64(values[i]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void fill_seeded_kernel(T *values, int32_t *status, uint64_t n, uint64_t first,
                                                             uint32_t seed, double frac) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t cell = first + i;
    const double v = (double)(float)(0.5 + mulberry32_at(seed, 2 * cell + 1));
    const bool keep = mulberry32_at(seed, 2 * cell + 2) < frac;
    T ov;
    int32_t os;
    emit_cell<T>(v, keep, false, ov, os);
    values[i] = ov;
    if (status) status[i] = os;
  }
}

// second half of a sharded average (in-memory.js:323-331) on reduced (sum, count) pairs
template <typename T>
__global__ __launch_bounds__(kBlock) void average_finish_kernel(T *values, const int32_t *counts, int32_t *status, uint64_t n,
                                                                int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const T v = values[i];
    const uint32_t c16 = (uint32_t)counts[i] & 0xFFFFu;  // Uint16Array counter
    double r = Cell<T>::to_f64(v);
    bool has = counts[i] != 0 && !Cell<T>::is_default(v, def_nan);
    if (c16) {
      r = (has ? r : (def_nan ? __builtin_nan("") : 0.0)) / (double)c16;
      has = !is_default_f64(r, def_nan);
    }
    T ov;
    int32_t os;
    emit_cell<T>(r, has, def_nan, ov, os);
    values[i] = ov;
    if (status) status[i] = os;
  }
}

// float64 total + count of set cells (in-memory.js:22-28), in two deterministic stages: every
// workgroup reduces a grid-strided share of the cells (16 B streaming loads, 4 in flight, wave
// shuffle, LDS) into its own slot of `partial`; total_finish_kernel adds the slots up in a fixed order.
struct TotalPartial {
  double sum;
  unsigned long long count;
};

template <typename T, bool VECTOR>
__global__ __launch_bounds__(kBlock) void total_kernel(const T *values, const int32_t *status, uint64_t n, int def_nan_i,
                                                       TotalPartial *partial) {
  constexpr int V = 16 / sizeof(T);
  const bool def_nan = def_nan_i != 0;
  const bool hs = status != nullptr;
  double acc = 0.0;
  unsigned long long cnt = 0;
  if constexpr (VECTOR) {
    // every workgroup streams ONE contiguous range of the buffer (a grid-stride sweep hands it 4 KB pieces 16 MB apart)
    const uint64_t n_groups = n / V;
    constexpr int U = 4;
    const uint64_t chunk = (n_groups + gridDim.x - 1) / gridDim.x;
    const uint64_t q_begin = (uint64_t)blockIdx.x * chunk;
    const uint64_t q_end = q_begin + chunk < n_groups ? q_begin + chunk : n_groups;
    uint32_t cnt32 = 0;  // (at most chunk * V / kBlock cells per lane: the plan keeps it below 2^32)
    for (uint64_t q = q_begin + threadIdx.x; q < q_end; q += (uint64_t)kBlock * U) {
      Vec<T, V> x[U];
      Vec<int32_t, V> sx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t qq = q + (uint64_t)u * kBlock;
        if (qq < q_end) {
          x[u] = load_stream<T, V>(values + qq * V);
          if (hs) sx[u] = load_stream<int32_t, V>(status + qq * V);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (q + (uint64_t)u * kBlock < q_end) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const bool set = cell_is_set<T>(x[u].v[e], hs ? sx[u].v[e] : OLAP_STATUS_SET, hs, def_nan);
            acc += set ? Cell<T>::to_f64(x[u].v[e]) : 0.0;
            cnt32 += set ? 1u : 0u;
          }
        }
      }
    }
    cnt = cnt32;
    if (blockIdx.x == 0 && threadIdx.x < n - n_groups * V) {  // the last n % V cells
      const uint64_t i = n_groups * V + threadIdx.x;
      if (cell_is_set<T>(values[i], hs ? status[i] : OLAP_STATUS_SET, hs, def_nan)) {
        acc += Cell<T>::to_f64(values[i]);
        ++cnt;
      }
    }
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
      const T v = values[i];
      if (cell_is_set<T>(v, hs ? status[i] : OLAP_STATUS_SET, hs, def_nan)) {
        acc += Cell<T>::to_f64(v);
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_down(acc, off, 64);
    cnt += __shfl_down(cnt, off, 64);
  }
  __shared__ double s_acc[kBlock / 64];
  __shared__ unsigned long long s_cnt[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_acc[wave] = acc;
    s_cnt[wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    TotalPartial p{0.0, 0};
    for (int w = 0; w < kBlock / 64; ++w) {
      p.sum += s_acc[w];
      p.count += s_cnt[w];
    }
    partial[blockIdx.x] = p;
  }
}

template <int UNUSED>  // (a template only so that every translation unit may carry it)
__global__ __launch_bounds__(kBlock) void total_finish_kernel(const TotalPartial *partial, uint32_t n_partial, double *total,
                                                              unsigned long long *count) {
  double acc = 0.0;
  unsigned long long cnt = 0;
  for (uint32_t i = threadIdx.x; i < n_partial; i += kBlock) {
    acc += partial[i].sum;
    cnt += partial[i].count;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_down(acc, off, 64);
    cnt += __shfl_down(cnt, off, 64);
  }
  __shared__ double s_acc[kBlock / 64];
  __shared__ unsigned long long s_cnt[kBlock / 64];
  if ((threadIdx.x & 63) == 0) {
    s_acc[threadIdx.x >> 6] = acc;
    s_cnt[threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    unsigned long long c = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
      a += s_acc[w];
      c += s_cnt[w];
    }
    *total = a;
    *count = c;
  }
}

// single-cell access for getValue / setValue on a handle store
// getValue (in-memory.js:118-120) without a blocking copy: one lane writes the cell (as a JS number) and its status word
// into pinned host memory; the caller waits for the stream once.
struct CellOut {
  double value;
  int32_t status;
  int32_t pad;
};
template <typename T>
__global__ void get_cell_kernel(const T *values, const int32_t *status, uint64_t index, CellOut *out) {
  out->value = Cell<T>::to_f64(values[index]);
  out->status = status ? status[index] : OLAP_STATUS_SET;
}

template <typename T>
__global__ void set_cell_kernel(T *values, int32_t *status, uint64_t index, double value, int is_null, int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  const bool has = !is_null && !is_default_f64(value, def_nan);
  T ov;
  int32_t os;
  emit_cell<T>(value, has, def_nan, ov, os);
  values[index] = ov;
  if (status) status[index] = os;
}

// ======================================================================= computed measures
// Element-wise postfix interpreter ("next" row f4).  Opcodes mirror olap-in-memory_amd/js/formula.js.
enum FormulaOp {
  F_CONST = 0, F_INPUT = 1, F_SCALAR = 2, F_ADD = 3, F_SUB = 4, F_MUL = 5, F_DIV = 6, F_MOD = 7, F_POW = 8, F_NEG = 9,
  F_NANADD = 10, F_SELECT = 11, F_MIN = 12, F_MAX = 13, F_ATAN2 = 14, F_HYPOT = 15, F_ROUNDTO = 16, F_ISNAN = 17,
  F_ABS = 20, F_CEIL = 21, F_FLOOR = 22, F_ROUND = 23, F_TRUNC = 24, F_SQRT = 25, F_CBRT = 26, F_EXP = 27, F_LN = 28,
  F_LOG10 = 29, F_LOG2 = 30, F_SIGN = 31, F_SIN = 32, F_COS = 33, F_TAN = 34, F_ASIN = 35, F_ACOS = 36, F_ATAN = 37, F_NOT = 38
};

struct FormulaProgram {
  int n_code;
  int32_t code[OLAP_FORMULA_MAX_CODE];
  double consts[OLAP_FORMULA_MAX_CONSTS];
  int n_inputs;
  const void *in_values[OLAP_FORMULA_MAX_INPUTS];
  const int32_t *in_status[OLAP_FORMULA_MAX_INPUTS];
  int in_dtype[OLAP_FORMULA_MAX_INPUTS];
  int in_def_nan[OLAP_FORMULA_MAX_INPUTS];
  double scalars[OLAP_FORMULA_MAX_INPUTS];
};

__device__ __forceinline__ bool js_truthy(double v) { return v == v && v != 0.0; }
__device__ __forceinline__ double js_round(double v) { return floor(v + 0.5); }  // Math.round: halves go up

template <int STACK>  // a template only so that the header may be included by several translation units
__global__ __launch_bounds__(kBlock) void eval_formula_kernel(const FormulaProgram p, double *__restrict__ out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    double st[STACK];
    int sp = 0;
    for (int pc = 0; pc < p.n_code; ++pc) {
      const int op = p.code[pc];
      if (op == F_CONST) {
        st[sp++] = p.consts[p.code[++pc]];
      } else if (op == F_INPUT) {
        const int k = p.code[++pc];
        double v;
        switch (p.in_dtype[k]) {  // getValue(i): the stored value, or the default where unset
          case OLAP_INT32: v = (double)((const int32_t *)p.in_values[k])[i]; break;
          case OLAP_UINT32: v = (double)((const uint32_t *)p.in_values[k])[i]; break;
          case OLAP_FLOAT32: v = (double)((const float *)p.in_values[k])[i]; break;
          default: v = ((const double *)p.in_values[k])[i]; break;
        }
        if (p.in_status[k] && !(p.in_status[k][i] & OLAP_STATUS_SET)) v = p.in_def_nan[k] ? __builtin_nan("") : 0.0;
        st[sp++] = v;
      } else if (op == F_SCALAR) {
        st[sp++] = p.scalars[p.code[++pc]];
      } else if (op == F_SELECT) {
        const double c = st[sp - 3], a = st[sp - 2], b = st[sp - 1];
        sp -= 2;
        st[sp - 1] = js_truthy(c) ? a : b;
      } else if (op == F_NEG || op == F_ISNAN || op >= F_ABS) {
        const double a = st[sp - 1];
        double r;
        switch (op) {
          case F_NEG: r = -a; break;
          case F_ISNAN: r = (a != a) ? 1.0 : 0.0; break;
          case F_ABS: r = fabs(a); break;
          case F_CEIL: r = ceil(a); break;
          case F_FLOOR: r = floor(a); break;
          case F_ROUND: r = js_round(a); break;
          case F_TRUNC: r = trunc(a); break;
          case F_SQRT: r = sqrt(a); break;
          case F_CBRT: r = cbrt(a); break;
          case F_EXP: r = exp(a); break;
          case F_LN: r = log(a); break;
          case F_LOG10: r = log10(a); break;
          case F_LOG2: r = log2(a); break;
          case F_SIGN: r = (a != a) ? a : (a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : a)); break;
          case F_SIN: r = sin(a); break;
          case F_COS: r = cos(a); break;
          case F_TAN: r = tan(a); break;
          case F_ASIN: r = asin(a); break;
          case F_ACOS: r = acos(a); break;
          case F_ATAN: r = atan(a); break;
          default: r = js_truthy(a) ? 0.0 : 1.0; break;  // F_NOT
        }
        st[sp - 1] = r;
      } else {
        const double a = st[sp - 2], b = st[sp - 1];
        --sp;
        double r;
        switch (op) {
          case F_ADD: r = a + b; break;
          case F_SUB: r = a - b; break;
          case F_MUL: r = a * b; break;
          case F_DIV: r = a / b; break;
          case F_MOD: r = fmod(a, b); break;
          case F_POW: r = pow(a, b); break;
          case F_NANADD: r = (a != a && b == b) ? b : ((a == a && b != b) ? a : a + b); break;  // src/parser.js:18-23
          case F_MIN: r = js_min(a, b); break;
          case F_MAX: r = js_max(a, b); break;
          case F_ATAN2: r = atan2(a, b); break;
          case F_HYPOT: r = hypot(a, b); break;
          default: {  // F_ROUNDTO
            const double f = pow(10.0, trunc(b));
            r = js_round(a * f) / f;
            break;
          }
        }
        st[sp - 1] = r;
      }
    }
    out[i] = sp > 0 ? st[sp - 1] : __builtin_nan("");
  }
}

// ======================================================================= sparse <-> dense
// Stream compaction of the set cells (ascending), for the reference's wire format: each workgroup
// owns a contiguous chunk; pass 1 counts, the host prefix-sums the (few thousand) chunk counts,
// pass 2 writes.  Inside a chunk the order comes from 64-bit wave ballots + popcounts.
template <typename T>
__global__ __launch_bounds__(kBlock) void compact_count_kernel(const T *values, const int32_t *status, uint64_t n,
                                                               uint64_t chunk, int def_nan_i, unsigned long long *counts) {
  const bool def_nan = def_nan_i != 0;
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  unsigned long long c = 0;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock)
    c += cell_is_set<T>(values[i], status ? status[i] : OLAP_STATUS_SET, status != nullptr, def_nan) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  __shared__ unsigned long long s[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void compact_write_kernel(const T *values, const int32_t *status, uint64_t n,
                                                               uint64_t chunk, int def_nan_i,
                                                               const unsigned long long *offsets, uint32_t *idx_out,
                                                               T *val_out) {
  const bool def_nan = def_nan_i != 0;
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  __shared__ unsigned long long s_wave[kBlock / 64];
  __shared__ unsigned long long s_base;
  if (threadIdx.x == 0) s_base = offsets[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t base = lo; base < hi; base += kBlock) {  // uniform trip count for the whole workgroup
    const uint64_t i = base + threadIdx.x;
    T v = T(0);
    bool set = false;
    if (i < hi) {
      v = values[i];
      set = cell_is_set<T>(v, status ? status[i] : OLAP_STATUS_SET, status != nullptr, def_nan);
    }
    const unsigned long long ballot = __ballot(set);
    if (lane == 0) s_wave[wave] = __popcll(ballot);
    __syncthreads();
    unsigned long long before = s_base;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (set) {
      const unsigned long long at = before + __popcll(ballot & ((1ull << lane) - 1ull));
      idx_out[at] = (uint32_t)i;
      val_out[at] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_sparse_kernel(T *values, const uint32_t *idx, const T *vals, uint64_t n_set,
                                                                uint64_t size) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_set; i += (uint64_t)gridDim.x * kBlock)
    if (idx[i] < size) values[idx[i]] = vals[i];
}

// ======================================================================= launchers
template <typename T>
struct Launch {
  static hipError_t drillup_axis(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out,
                                 int32_t *st_out, const DrillUpAxis &a, hipStream_t stream);
  // the same roll-up over nb (<= kMaxBatch) buffer pairs in one launch; all with or all without a mask
  static hipError_t drillup_axis_batch(int method, bool has_status, int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a,
                                       hipStream_t stream);
  // pairs with DIFFERENT rules (b.method[]) in one launch: the row regime (16-byte lanes, or 8-byte lanes of 4-byte cells)
  // and the row-tile regime — anything else returns hipErrorNotSupported and the caller launches rule by rule.  `deep`:
  // some pair's rule wants 4 rows in flight.
  static hipError_t drillup_rows_mixed(bool has_status, int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a, bool deep,
                                       hipStream_t stream);
  static hipError_t drillup_reduce(int method, bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                   const DrillUpAxis &a, const DrillUpReduce &rd, hipStream_t stream);
  static hipError_t drillup_generic(int method, bool has_status, const T *in, const int32_t *st_in, T *out,
                                    int32_t *st_out, const DrillUpGeneric &a, hipStream_t stream);
  static hipError_t gather(bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                           const Remap &r, hipStream_t stream);
  static hipError_t gather_reduce(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out,
                                  int32_t *st_out, const GatherReduce &a, hipStream_t stream);
  static hipError_t reorder_brick(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                  const Brick &b, uint64_t n_bricks, hipStream_t stream);
  static hipError_t load_scatter(bool has_status, int vec, const T *his, const int32_t *his_st, T *mine, int32_t *mine_st,
                                 const Remap &r, hipStream_t stream);
  static hipError_t drilldown(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                              const DrillDown &a, hipStream_t stream);
  static hipError_t drilldown_rows(bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                   const DrillUpAxis &a, int divide, int use_rounding, uint32_t longest_group, hipStream_t stream);
  static hipError_t dice_direct(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out, const DiceRows &a,
                                hipStream_t stream);
  static hipError_t drilldown_rows_lines(bool has_status, bool any_shift, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                         const DrillUpAxis &a, int divide, uint32_t longest_group, hipStream_t stream);
  static hipError_t drilldown_scale(bool has_status, const T *in, const int32_t *st_in, T *q, const DrillDownScale &a,
                                    hipStream_t stream);
  static hipError_t canonicalize(T *values, int32_t *status, uint64_t n, int def_nan, int use_status,
                                 hipStream_t stream);
  static hipError_t from_f64(const double *src, T *values, int32_t *status, uint64_t n, int def_nan,
                             hipStream_t stream);
  static hipError_t to_f64(const T *values, double *dst, uint64_t n, hipStream_t stream);
  static hipError_t fill_seeded(T *values, int32_t *status, uint64_t n, uint64_t first, uint32_t seed, double frac,
                                hipStream_t stream);
  static hipError_t average_finish(T *values, const int32_t *counts, int32_t *status, uint64_t n, int def_nan,
                                   hipStream_t stream);
  static hipError_t drillup_segmented(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                      const DrillUpAxis &a, const SegmentedRows &sg, hipStream_t stream);
  static hipError_t load_permute(bool has_status, const T *his, const int32_t *his_st, T *mine, int32_t *mine_st, const LoadPermute &a,
                                 hipStream_t stream);
  static hipError_t total(const T *values, const int32_t *status, uint64_t n, int def_nan, void *workspace, double *total,
                          unsigned long long *count, hipStream_t stream);
  static hipError_t compact_count(const T *values, const int32_t *status, uint64_t n, uint64_t chunk, unsigned n_chunks,
                                  int def_nan, unsigned long long *counts, hipStream_t stream);
  static hipError_t compact_write(const T *values, const int32_t *status, uint64_t n, uint64_t chunk, unsigned n_chunks,
                                  int def_nan, const unsigned long long *offsets, uint32_t *idx_out, T *val_out,
                                  hipStream_t stream);
  static hipError_t scatter_sparse(T *values, const uint32_t *idx, const T *vals, uint64_t n_set, uint64_t size,
                                   hipStream_t stream);
  static hipError_t get_cell(const T *values, const int32_t *status, uint64_t index, CellOut *out, hipStream_t stream);
  static hipError_t set_cell(T *values, int32_t *status, uint64_t index, double value, int is_null, int def_nan,
                             hipStream_t stream);
};

inline size_t lds_pad_host(size_t e) { return e + (e >> 5); }
inline unsigned grid_for(uint64_t threads) { return (unsigned)((threads + kBlock - 1) / kBlock); }
inline unsigned grid_stride_for(uint64_t n) {
  const uint64_t want = (n + kBlock - 1) / kBlock;
  const uint64_t cap = 256ull * 8ull;  // 256 CUs x 8 resident workgroups
  return (unsigned)(want < 1 ? 1 : (want < cap ? want : cap));
}

#ifdef OLAP_KERNELS_IMPL

// Whether the LDS row-tile regime takes this roll-up (small `inner`: short row pieces make the flat regime's accesses
// waste much of every cache line; whole rows of K*inner cells staged per workgroup, kTileBytes of cells), and how.
struct TileGeometry {
  DrillUpTile tl{};
  uint64_t tiles = 0;
  size_t lds = 0;
  int mode = 0;  // table mode of drillup_tile_kernel
};
template <typename T>
static bool tile_geometry(const DrillUpAxis &a, bool has_status, TileGeometry *out) {
  const bool contig = a.order == nullptr;
  const uint64_t row_elems = a.K * a.inner;
  const uint64_t budget = kTileBytes / sizeof(T);
  constexpr uint64_t V = 16 / sizeof(T);
  static const bool no_permute = getenv("OLAP_TILE_NO_PERMUTE") != nullptr;
  // interleaved groups: cells permuted into group order while they are staged (MODE 3, tables from the plan)
  const bool permute = !contig && a.perm_cell && !no_permute;
  const uint64_t csr_bytes = permute ? 2 * a.G * 4 : (a.G + 1 + (contig ? 0 : a.K)) * 4;
  // rows per tile: as many as fit; every tile must start 16 B aligned, i.e. R*row_elems % V == 0
  const uint64_t R = tile_rows_for(row_elems, permute ? (uint64_t)a.perm_pitch * a.inner : row_elems, budget, V);
  uint64_t tile_max_inner = 128;  // tools/sweep3.py: the LDS form wins up to ~100 cells per row piece
  if (const char *e = getenv("OLAP_TILE_MAX_INNER")) tile_max_inner = (uint64_t)atoll(e);
  if (!(a.aligned16 && a.inner < tile_max_inner && R > 0 && csr_bytes <= 16 * 1024 && a.G * a.inner <= 0xFFFFFFFFull)) return false;
  DrillUpTile &tl = out->tl;
  tl.rows_per_tile = (uint32_t)R;
  tl.row_elems = (uint32_t)row_elems;
  tl.out_row = (uint32_t)(a.G * a.inner);
  tl.inner = (uint32_t)a.inner;
  if (permute) {
    tl.perm_cell = a.perm_cell;
    tl.perm_grp = a.perm_grp;
    tl.pitch_cells = (uint32_t)(a.perm_pitch * a.inner);
    if (!small_div_for(row_elems, budget, &tl.by_row)) return false;  // (row_elems <= 4096: always exact)
  }
  out->tiles = (a.outer + tl.rows_per_tile - 1) / tl.rows_per_tile;
  const bool all = contig && a.G == 1;
  out->lds = kTileBytes + (has_status ? budget * 4 : 0) + (all ? 0 : csr_bytes);
  out->mode = all ? 1 : contig ? 2 : permute ? 3 : 0;
  return out->tiles < 0x7FFFFFFFull;
}

// Lanes per workgroup of the row regime.  A row takes a whole number of workgroups: 271 slots fill 271 of 512 lanes at 256
// lanes per workgroup, 271 of 320 at 64 — the widest workgroup that fills >= 85 % of its lanes, else the best filled.
// A row of up to 256 slots takes ONE workgroup, the smallest that holds it: two workgroups per row of which the second
// is nearly empty is the worst choice by far (135 slots: 128 + 7 lanes 99 us, one workgroup of 256 lanes 75 us, 64 + 64
// + 7 lanes 78 us; 69 slots: 64 + 5 lanes 100 us, one of 128 lanes 76 us — full and empty workgroups alternate, and
// with an even number of CUs per XCD the full ones keep landing on the same half of them).
static inline unsigned rows_lanes_for(uint64_t n_vec) {
  if (n_vec <= 64) return 64u;
  if (n_vec <= 128) return 128u;
  if (n_vec <= 256) return 256u;
  unsigned lanes = kBlock;
  double best = 0.0;
  for (unsigned cand : {256u, 128u, 64u}) {
    const double fill = (double)n_vec / (double)(((n_vec + cand - 1) / cand) * cand);
    if (fill >= 0.85) return cand;
    if (fill > best) { best = fill; lanes = cand; }
  }
  return lanes;
}

// Row regime when a row of VEC-slots fills at least one wavefront-sized piece of a workgroup
// reasonably (>= 128 slots); otherwise the flat regime.  FAST = additive method, zero default, no
// mask read.  The grid of the row regime is outer*G*blocks_per_row workgroups.
template <typename T, int METHOD, bool HS, int VEC>
static hipError_t drillup_axis_launch(const Batch<T> &b, unsigned nb, const DrillUpAxis &a, hipStream_t stream) {
  constexpr bool kAdditive = (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE);
  const bool fast = kAdditive && !HS && !a.def_nan;
  const bool contig = a.order == nullptr;
  // a row takes a whole number of workgroups: 271 slots fill 271 of 512 lanes at 256 lanes per
  // workgroup, 271 of 320 at 64 — the row regime picks the largest workgroup that fills >= 85 %
  unsigned row_lanes = kBlock;
  {
    static const int forced = getenv("OLAP_ROWS_LANES") ? atoi(getenv("OLAP_ROWS_LANES")) : 0;
    row_lanes = rows_lanes_for(a.n_vec);
    if (forced == 64 || forced == 128 || forced == 256) row_lanes = (unsigned)forced;
  }
  const uint64_t row_blocks = a.outer * a.G * ((a.n_vec + row_lanes - 1) / row_lanes);
  const bool rows = a.n_vec >= 128 && row_blocks < 0x7FFFFFFFull;
  // Rows in flight per lane.  With full 16 B lanes and rows that fill whole workgroups ONE is
  // fastest for the plain streaming forms (tools/microbench.hip, profiles/microbench_r01.txt: 67 us
  // against 70-71 us at 2..10 on the 10^8 cube): 32 waves per CU already keep 32 KiB in flight and
  // the waves sweep the K rows together.  Narrow lanes or short rows (e.g. inner = 274) need the
  // depth back, and so does the heavier exact state machine.
  constexpr int U = 4;
  bool shallow = (kAdditive || IsPick<METHOD>::value) && VEC * sizeof(T) >= 16 && a.n_vec >= 1024;
  if (a.depth == 4) shallow = false;
  if (const char *e = getenv("OLAP_ROWS_DEPTH")) shallow = atoi(e) == 1;  // A/B: 1 or 4 rows in flight
  if (!rows) {
    // LDS tile regime for small `inner` (short row pieces make the flat regime's accesses waste much
    // of every cache line): whole rows of K*inner cells staged per workgroup, kTileBytes of cells
    TileGeometry tg;
    if (tile_geometry<T>(a, HS, &tg)) {
      const DrillUpTile &tl = tg.tl;
      const uint64_t tiles = tg.tiles;
      const size_t lds = tg.lds;
      const bool all = tg.mode == 1, permute = tg.mode == 3;
      if (tiles < 0x7FFFFFFFull) {
#define OLAP_TILE(F, M) hipLaunchKernelGGL((drillup_tile_kernel<T, METHOD, HS, F, M>), dim3((unsigned)tiles, nb), kBlock, lds, stream, b, a, tl)
        if constexpr (kAdditive && !HS) {
          if (fast) {
            if (all) OLAP_TILE(true, 1); else if (contig) OLAP_TILE(true, 2); else if (permute) OLAP_TILE(true, 3); else OLAP_TILE(true, 0);
            return hipGetLastError();
          }
        }
        if (all) OLAP_TILE(false, 1); else if (contig) OLAP_TILE(false, 2); else if (permute) OLAP_TILE(false, 3); else OLAP_TILE(false, 0);
#undef OLAP_TILE
        return hipGetLastError();
      }
    }
  }
  if (a.gtile && a.n_gtile > 0 && a.aligned16 && a.outer * a.n_gtile < 0x7FFFFFFFull) {  // (the plan decides; also over rows of >= 128 slots)
    const size_t lds = kTileBytes + (HS ? kTileBytes / sizeof(T) * 4 : 0) + (kGroupTileMaxGroups + 1) * 4;
    const unsigned blocks = (unsigned)(a.outer * a.n_gtile);
    const uint64_t n_cells = a.outer * a.K * a.inner;
#define OLAP_GTILE(F) hipLaunchKernelGGL((drillup_gtile_kernel<T, METHOD, HS, F>), dim3(blocks, nb), kBlock, lds, stream, b, a, n_cells)
    if constexpr (kAdditive && !HS) {
      if (fast) { OLAP_GTILE(true); return hipGetLastError(); }
    }
    OLAP_GTILE(false);
#undef OLAP_GTILE
    return hipGetLastError();
  }
#define OLAP_ROWS(C, F) hipLaunchKernelGGL((drillup_rows_kernel<T, METHOD, HS, VEC, U, C, F>), dim3((unsigned)row_blocks, nb), row_lanes, 0, stream, b, ar)
#define OLAP_ROWS1(C, F) hipLaunchKernelGGL((drillup_rows_kernel<T, METHOD, HS, VEC, 1, C, F>), dim3((unsigned)row_blocks, nb), row_lanes, 0, stream, b, ar)
#define OLAP_FLAT(F)                                                                                                                              \
  do {                                                                                                                                             \
    if (a.total < 0xFFFFFF00ull) hipLaunchKernelGGL((drillup_flat_kernel<T, METHOD, HS, VEC, F, uint32_t>), dim3(grid_for(a.total), nb), kBlock, 0, stream, b, a); \
    else hipLaunchKernelGGL((drillup_flat_kernel<T, METHOD, HS, VEC, F, uint64_t>), dim3(grid_for(a.total), nb), kBlock, 0, stream, b, a);                        \
  } while (0)
  DrillUpAxis ar = a;  // the row regime's own workgroup width
  ar.blocks_per_row = (a.n_vec + row_lanes - 1) / row_lanes;
  ar.lanes = row_lanes;
  ar.grid = rows ? (uint32_t)row_blocks : 0u;
  if constexpr (VEC * sizeof(T) < 16) {
    // rows that are not whole 16-byte groups (or buffers that are not 16-byte aligned): 16-byte slots at
    // cell-aligned addresses instead of 4-byte lanes
    constexpr int RV = 16 / sizeof(T);
    static const bool no_ragged = getenv("OLAP_NO_RAGGED_ROWS") != nullptr;
    // (8-byte lanes — 4-byte cells in rows of an even, not fourfold, number of cells — too: [3652,100,274] city -> country
    // 79-82 us with 8-byte lanes, 76 us with these; OLAP_NO_RAGGED_VEC2 keeps the 8-byte lanes)
    static const bool no_ragged2 = getenv("OLAP_NO_RAGGED_VEC2") != nullptr;
    if (rows && !no_ragged && (VEC == 1 || !no_ragged2)) {
      DrillUpAxis rg = a;
      rg.n_vec = (a.inner + RV - 1) / RV;
      unsigned lanes = rows_lanes_for(rg.n_vec);
      {
        static const int forced = getenv("OLAP_ROWS_LANES") ? atoi(getenv("OLAP_ROWS_LANES")) : 0;
        if (forced == 64 || forced == 128 || forced == 256) lanes = (unsigned)forced;
      }
      rg.blocks_per_row = (rg.n_vec + lanes - 1) / lanes;
      rg.lanes = lanes;
      const uint64_t blocks = a.outer * a.G * rg.blocks_per_row;
      rg.grid = blocks < 0x7FFFFFFFull ? (uint32_t)blocks : 0u;
      if (blocks < 0x7FFFFFFFull) {
#define OLAP_RAGGED(C, F) hipLaunchKernelGGL((drillup_rows_kernel<T, METHOD, HS, RV, U, C, F, true, true>), dim3((unsigned)blocks, nb), lanes, 0, stream, b, rg)
        if constexpr (kAdditive && !HS) {
          if (fast) {
            if (contig) OLAP_RAGGED(true, true); else OLAP_RAGGED(false, true);
            return hipGetLastError();
          }
        }
        if (contig) OLAP_RAGGED(true, false); else OLAP_RAGGED(false, false);
#undef OLAP_RAGGED
        return hipGetLastError();
      }
    }
  }
  if (rows) {
    if constexpr (kAdditive && !HS) {
      if (fast) {
        if (shallow) { if (contig) OLAP_ROWS1(true, true); else OLAP_ROWS1(false, true); }
        else { if (contig) OLAP_ROWS(true, true); else OLAP_ROWS(false, true); }
        return hipGetLastError();
      }
    }
    if constexpr (IsPick<METHOD>::value && !HS) {
      if (shallow) {
        if (contig) OLAP_ROWS1(true, false); else OLAP_ROWS1(false, false);
        return hipGetLastError();
      }
    }
    if (HS && shallow) {  // values + mask: two wide streams per row already, one row in flight (155 -> 147 us)
      if (contig) OLAP_ROWS1(true, false); else OLAP_ROWS1(false, false);
      return hipGetLastError();
    }
    if (contig) OLAP_ROWS(true, false); else OLAP_ROWS(false, false);
  } else {
    if constexpr (kAdditive && !HS) {
      if (fast) { OLAP_FLAT(true); return hipGetLastError(); }
    }
    OLAP_FLAT(false);
  }
#undef OLAP_ROWS
#undef OLAP_ROWS1
#undef OLAP_FLAT
  return hipGetLastError();
}

template <typename T, int METHOD, bool HS>
static hipError_t drillup_axis_vec(int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a, hipStream_t stream) {
  if (vec == 4) return drillup_axis_launch<T, METHOD, HS, 4>(b, nb, a, stream);
  if (vec == 2) return drillup_axis_launch<T, METHOD, HS, 2>(b, nb, a, stream);
  return drillup_axis_launch<T, METHOD, HS, 1>(b, nb, a, stream);
}

template <typename T, bool HS>
static hipError_t drillup_axis_method(int method, int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a, hipStream_t stream) {
  switch (method) {
    case OLAP_SUM: return drillup_axis_vec<T, OLAP_SUM, HS>(vec, b, nb, a, stream);
    case OLAP_AVERAGE: return drillup_axis_vec<T, OLAP_AVERAGE, HS>(vec, b, nb, a, stream);
    case OLAP_HIGHEST: return drillup_axis_vec<T, OLAP_HIGHEST, HS>(vec, b, nb, a, stream);
    case OLAP_LOWEST: return drillup_axis_vec<T, OLAP_LOWEST, HS>(vec, b, nb, a, stream);
    case OLAP_FIRST: return drillup_axis_vec<T, OLAP_FIRST, HS>(vec, b, nb, a, stream);
    case OLAP_LAST: return drillup_axis_vec<T, OLAP_LAST, HS>(vec, b, nb, a, stream);
    case OLAP_PARTIAL_AVERAGE: return drillup_axis_vec<T, OLAP_PARTIAL_AVERAGE, HS>(vec, b, nb, a, stream);
    default: return drillup_axis_vec<T, OLAP_PRODUCT, HS>(vec, b, nb, a, stream);
  }
}

template <typename T>
hipError_t Launch<T>::drillup_axis(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out,
                                   int32_t *st_out, const DrillUpAxis &a, hipStream_t stream) {
  return drillup_axis_batch(method, has_status, vec, Batch<T>::one(in, st_in, out, st_out), 1, a, stream);
}

template <typename T>
hipError_t Launch<T>::drillup_axis_batch(int method, bool has_status, int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a,
                                         hipStream_t stream) {
  if (a.total == 0 || nb == 0) return hipSuccess;
  return has_status ? drillup_axis_method<T, true>(method, vec, b, nb, a, stream)
                    : drillup_axis_method<T, false>(method, vec, b, nb, a, stream);
}

template <typename T, int VEC>
static hipError_t drillup_rows_mixed_vec(bool has_status, const Batch<T> &b, unsigned nb, const DrillUpAxis &a, bool deep, hipStream_t stream) {
  const unsigned row_lanes = rows_lanes_for(a.n_vec);  // (as drillup_axis_launch picks it)
  DrillUpAxis ar = a;
  ar.blocks_per_row = (a.n_vec + row_lanes - 1) / row_lanes;
  ar.lanes = row_lanes;
  const uint64_t row_blocks = a.outer * a.G * ar.blocks_per_row;
  if (row_blocks >= 0x7FFFFFFFull) return hipErrorNotSupported;
  ar.grid = (uint32_t)row_blocks;
  const bool shallow = !deep && VEC * sizeof(T) >= 16 && a.n_vec >= 1024;
  const bool contig = a.order == nullptr;
  const dim3 grid((unsigned)row_blocks, nb);
#define OLAP_MIXED(HS, UU, C) hipLaunchKernelGGL((drillup_rows_mixed_kernel<T, HS, VEC, UU, C>), grid, row_lanes, 0, stream, b, ar)
  if (has_status) {
    if (shallow) { if (contig) OLAP_MIXED(true, 1, true); else OLAP_MIXED(true, 1, false); }
    else { if (contig) OLAP_MIXED(true, 4, true); else OLAP_MIXED(true, 4, false); }
  } else {
    if (shallow) { if (contig) OLAP_MIXED(false, 1, true); else OLAP_MIXED(false, 1, false); }
    else { if (contig) OLAP_MIXED(false, 4, true); else OLAP_MIXED(false, 4, false); }
  }
#undef OLAP_MIXED
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drillup_rows_mixed(bool has_status, int vec, const Batch<T> &b, unsigned nb, const DrillUpAxis &a, bool deep,
                                         hipStream_t stream) {
  constexpr int FULL = 16 / sizeof(T);
  if (a.total == 0 || nb == 0) return hipSuccess;
  // the row regime with 16-byte lanes, or 8-byte lanes of 4-byte cells (rows of an even number of cells: config 5's
  // [120,100,274] -> country); single cells per lane and the other regimes go rule by rule
  if (!a.aligned16) return hipErrorNotSupported;
  if (a.n_vec < 128) {  // not the row regime: the row-tile regime takes mixed rules too, the others do not
    TileGeometry tg;
    if (!tile_geometry<T>(a, has_status, &tg)) return hipErrorNotSupported;
    const dim3 grid((unsigned)tg.tiles, nb);
#define OLAP_TILE_MIXED(HS, M) hipLaunchKernelGGL((drillup_tile_mixed_kernel<T, HS, M>), grid, kBlock, tg.lds, stream, b, a, tg.tl)
    if (has_status) {
      switch (tg.mode) {
        case 1: OLAP_TILE_MIXED(true, 1); break;
        case 2: OLAP_TILE_MIXED(true, 2); break;
        case 3: OLAP_TILE_MIXED(true, 3); break;
        default: OLAP_TILE_MIXED(true, 0); break;
      }
    } else {
      switch (tg.mode) {
        case 1: OLAP_TILE_MIXED(false, 1); break;
        case 2: OLAP_TILE_MIXED(false, 2); break;
        case 3: OLAP_TILE_MIXED(false, 3); break;
        default: OLAP_TILE_MIXED(false, 0); break;
      }
    }
#undef OLAP_TILE_MIXED
    return hipGetLastError();
  }
  if (vec == FULL) return drillup_rows_mixed_vec<T, FULL>(has_status, b, nb, a, deep, stream);
  if constexpr (FULL == 4) {
    if (vec == 2) return drillup_rows_mixed_vec<T, 2>(has_status, b, nb, a, deep, stream);
  }
  return hipErrorNotSupported;
}

template <typename T, int METHOD>
static hipError_t drillup_reduce_launch(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                        const DrillUpAxis &a, const DrillUpReduce &rd, hipStream_t stream) {
  const uint64_t cells = a.outer * a.G * a.inner;
  constexpr bool kAdditive = (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE);
  const bool fast = kAdditive && !has_status && !a.def_nan;
  // cooperative form: rows of up to 128 cells, or — 16-byte form only — of up to 1 024 (the scalar form gives a lane
  // one cell of one row per step: rows must fit the unit's lanes; otherwise the lane-per-cell split form below)
  const bool vec4_now = rd.vec4 && a.aligned16 && (rd.rows > 0 || sizeof(T) == 4);  // (16-byte form: 4 cells of 4 bytes, 2 of 8)
  if (rd.rows > 0 && (vec4_now || a.inner <= rd.unit)) {
    const uint64_t upb = kBlock / rd.unit;
    const unsigned grid = (unsigned)((a.outer * a.G * rd.S + upb - 1) / upb);
    if (vec4_now) {
      if (has_status) hipLaunchKernelGGL((drillup_reduce4_kernel<T, METHOD, true, false>), grid, kBlock, 0, stream, in, st_in, out, st_out, a, rd);
      else if (kAdditive && fast) hipLaunchKernelGGL((drillup_reduce4_kernel<T, METHOD, false, kAdditive>), grid, kBlock, 0, stream, in, st_in, out, st_out, a, rd);
      else hipLaunchKernelGGL((drillup_reduce4_kernel<T, METHOD, false, false>), grid, kBlock, 0, stream, in, st_in, out, st_out, a, rd);
      if (rd.S == 1) return hipGetLastError();  // one segment per group: the result left from the reduction
    } else {
      DrillUpReduce r1 = rd;
      if (rd.vec4) {  // plan sized `rows` for 16 B lanes; the scalar form covers unit cells per step
        uint32_t rows = 1;
        while ((uint64_t)rows * 2 * a.inner <= rd.unit) rows *= 2;
        r1.rows = rows;
      }
      if (has_status) hipLaunchKernelGGL((drillup_reduce_kernel<T, METHOD, true, false>), grid, kBlock, 0, stream, in, st_in, a, r1);
      else if (kAdditive && fast) hipLaunchKernelGGL((drillup_reduce_kernel<T, METHOD, false, kAdditive>), grid, kBlock, 0, stream, in, st_in, a, r1);
      else hipLaunchKernelGGL((drillup_reduce_kernel<T, METHOD, false, false>), grid, kBlock, 0, stream, in, st_in, a, r1);
    }
  } else if (rd.rows == 0 && rd.vec4 && a.aligned16 && sizeof(T) == 4 && a.inner % 4 == 0) {  // 16-byte lanes
    const unsigned grid = grid_for(cells / 4 * rd.S);
    if (has_status) hipLaunchKernelGGL((drillup_split4_kernel<T, METHOD, true, false>), grid, kBlock, 0, stream, in, st_in, a, rd);
    else if (kAdditive && fast) hipLaunchKernelGGL((drillup_split4_kernel<T, METHOD, false, kAdditive>), grid, kBlock, 0, stream, in, st_in, a, rd);
    else hipLaunchKernelGGL((drillup_split4_kernel<T, METHOD, false, false>), grid, kBlock, 0, stream, in, st_in, a, rd);
  } else {
    const unsigned grid = grid_for(cells * rd.S);
    if (has_status) hipLaunchKernelGGL((drillup_split_kernel<T, METHOD, true, false>), grid, kBlock, 0, stream, in, st_in, a, rd);
    else if (kAdditive && fast) hipLaunchKernelGGL((drillup_split_kernel<T, METHOD, false, kAdditive>), grid, kBlock, 0, stream, in, st_in, a, rd);
    else hipLaunchKernelGGL((drillup_split_kernel<T, METHOD, false, false>), grid, kBlock, 0, stream, in, st_in, a, rd);
  }
  if (rd.S <= 16) hipLaunchKernelGGL((drillup_merge_few_kernel<T, METHOD>), grid_for(cells), kBlock, 0, stream, out, st_out, a, rd);
  else if (rd.S >= 512) hipLaunchKernelGGL((drillup_merge_block_kernel<T, METHOD>), (unsigned)cells, kBlock, 0, stream, out, st_out, a, rd);
  else hipLaunchKernelGGL((drillup_merge_kernel<T, METHOD>), grid_for(cells * 64), kBlock, 0, stream, out, st_out, a, rd);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drillup_reduce(int method, bool has_status, const T *in, const int32_t *st_in, T *out,
                                     int32_t *st_out, const DrillUpAxis &a, const DrillUpReduce &rd, hipStream_t stream) {
  if (a.outer * a.G * a.inner == 0) return hipSuccess;
  switch (method) {
    case OLAP_SUM: return drillup_reduce_launch<T, OLAP_SUM>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_AVERAGE: return drillup_reduce_launch<T, OLAP_AVERAGE>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_HIGHEST: return drillup_reduce_launch<T, OLAP_HIGHEST>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_LOWEST: return drillup_reduce_launch<T, OLAP_LOWEST>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_FIRST: return drillup_reduce_launch<T, OLAP_FIRST>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_LAST: return drillup_reduce_launch<T, OLAP_LAST>(has_status, in, st_in, out, st_out, a, rd, stream);
    case OLAP_PARTIAL_AVERAGE: return drillup_reduce_launch<T, OLAP_PARTIAL_AVERAGE>(has_status, in, st_in, out, st_out, a, rd, stream);
    default: return drillup_reduce_launch<T, OLAP_PRODUCT>(has_status, in, st_in, out, st_out, a, rd, stream);
  }
}

template <typename T>
hipError_t Launch<T>::drillup_segmented(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                        const DrillUpAxis &a, const SegmentedRows &sg, hipStream_t stream) {
  if (a.outer * a.G * a.inner == 0) return hipSuccess;
  const bool additive = method == OLAP_SUM || method == OLAP_AVERAGE;
  const bool mask_primary = has_status;  // (the plan passes the mask only where it carries information)
  // stage 1: the row regime over the segments
  DrillUpAxis a1 = a;
  a1.G = sg.S_tot;
  a1.gstart = sg.gstart;
  a1.gtile = nullptr;
  a1.n_gtile = 0;
  a1.perm_cell = nullptr;
  a1.total = a1.outer * a1.G * a1.n_vec;
  a1.aligned16 = a.aligned16;  // (the partial buffers come from the pool: 256-byte aligned)
  a1.depth = 4;                // few, long workgroups (two per CU): four rows in flight per lane
  const bool want_aux = additive ? (method == OLAP_AVERAGE || a.def_nan || mask_primary) : mask_primary;
  int32_t *aux = want_aux ? sg.aux : nullptr;
  hipError_t e = drillup_axis(additive ? OLAP_PARTIAL_AVERAGE : method, has_status, vec, in, st_in, (T *)sg.partial, aux, a1, stream);
  if (e != hipSuccess) return e;
  // stage 2: a group's segments folded in order
  const uint64_t blocks = a.outer * a.G * ((a.inner + 63) / 64);
  if (blocks >= 0x7FFFFFFFull) return hipErrorInvalidValue;
#define OLAP_SEGC(M) hipLaunchKernelGGL((segments_combine_kernel<T, M>), (unsigned)blocks, kBlock, 0, stream, (const void *)sg.partial, (const int32_t *)aux, out, st_out, a, sg)
  switch (method) {
    case OLAP_SUM: OLAP_SEGC(OLAP_SUM); break;
    case OLAP_AVERAGE: OLAP_SEGC(OLAP_AVERAGE); break;
    case OLAP_HIGHEST: OLAP_SEGC(OLAP_HIGHEST); break;
    case OLAP_LOWEST: OLAP_SEGC(OLAP_LOWEST); break;
    case OLAP_FIRST: OLAP_SEGC(OLAP_FIRST); break;
    default: OLAP_SEGC(OLAP_LAST); break;
  }
#undef OLAP_SEGC
  return hipGetLastError();
}

template <typename T, bool HS>
static hipError_t drillup_generic_method(int method, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                         const DrillUpGeneric &a, hipStream_t stream) {
  const unsigned grid = grid_for(a.total);
#define OLAP_GEN(M) hipLaunchKernelGGL((drillup_generic_kernel<T, M, HS>), grid, kBlock, 0, stream, in, st_in, out, st_out, a)
  switch (method) {
    case OLAP_SUM: OLAP_GEN(OLAP_SUM); break;
    case OLAP_AVERAGE: OLAP_GEN(OLAP_AVERAGE); break;
    case OLAP_HIGHEST: OLAP_GEN(OLAP_HIGHEST); break;
    case OLAP_LOWEST: OLAP_GEN(OLAP_LOWEST); break;
    case OLAP_FIRST: OLAP_GEN(OLAP_FIRST); break;
    case OLAP_LAST: OLAP_GEN(OLAP_LAST); break;
    case OLAP_PARTIAL_AVERAGE: OLAP_GEN(OLAP_PARTIAL_AVERAGE); break;
    default: OLAP_GEN(OLAP_PRODUCT); break;
  }
#undef OLAP_GEN
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drillup_generic(int method, bool has_status, const T *in, const int32_t *st_in, T *out,
                                      int32_t *st_out, const DrillUpGeneric &a, hipStream_t stream) {
  if (a.total == 0) return hipSuccess;
  return has_status ? drillup_generic_method<T, true>(method, in, st_in, out, st_out, a, stream)
                    : drillup_generic_method<T, false>(method, in, st_in, out, st_out, a, stream);
}

template <typename T>
hipError_t Launch<T>::dice_direct(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out, const DiceRows &p,
                                  hipStream_t stream) {
  DiceDirect a;
  if (!dice_direct_fits<T>(p, &a)) return hipErrorInvalidValue;
  const int groups = dice_direct_groups();
  const uint64_t SPAN = (uint64_t)groups * kBlock * (16 / sizeof(T));
  const uint64_t n_out = p.outer * p.k_new * p.inner;
  if (n_out == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n_out + SPAN - 1) / SPAN);
  const size_t lds = (size_t)a.rows * sizeof(int64_t);  // <= (SPAN / V + 3) entries
#define OLAP_DD(HS, UU) hipLaunchKernelGGL((dice_direct_kernel<T, HS, UU>), blocks, kBlock, lds, stream, in, st_in, out, st_out, a)
  if (has_status) {
    if (groups == 1) OLAP_DD(true, 1); else if (groups == 2) OLAP_DD(true, 2); else if (groups == 8) OLAP_DD(true, 8); else OLAP_DD(true, 4);
  } else {
    if (groups == 1) OLAP_DD(false, 1); else if (groups == 2) OLAP_DD(false, 2); else if (groups == 8) OLAP_DD(false, 8); else OLAP_DD(false, 4);
  }
#undef OLAP_DD
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::gather(bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                             const Remap &r, hipStream_t stream) {
  if (r.total == 0) return hipSuccess;
  const unsigned grid = grid_for(r.total);
  const bool idx32 = r.total * (uint64_t)vec < 0xFFFFFFFFull;
#define OLAP_G(HS, V)                                                                                                        \
  do {                                                                                                                       \
    if (idx32) hipLaunchKernelGGL((gather_kernel<T, HS, V, uint32_t>), grid, kBlock, 0, stream, in, st_in, out, st_out, r);  \
    else hipLaunchKernelGGL((gather_kernel<T, HS, V, uint64_t>), grid, kBlock, 0, stream, in, st_in, out, st_out, r);        \
  } while (0)
  if (has_status) {
    if (vec == 4) OLAP_G(true, 4); else if (vec == 2) OLAP_G(true, 2); else OLAP_G(true, 1);
  } else {
    if (vec == 4) OLAP_G(false, 4); else if (vec == 2) OLAP_G(false, 2); else OLAP_G(false, 1);
  }
#undef OLAP_G
  return hipGetLastError();
}

template <typename T, int METHOD, bool HS>
static hipError_t gather_reduce_vec(int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                    const GatherReduce &a, hipStream_t stream) {
  constexpr bool kAdditive = (METHOD == OLAP_SUM || METHOD == OLAP_AVERAGE || METHOD == OLAP_PARTIAL_AVERAGE);
  const bool fast = kAdditive && !HS && !a.r.def_nan;
  const unsigned grid = grid_for(a.r.total);
#define OLAP_GR(V, F) hipLaunchKernelGGL((gather_reduce_kernel<T, METHOD, HS, V, F>), grid, kBlock, 0, stream, in, st_in, out, st_out, a)
  if constexpr (kAdditive && !HS) {
    if (fast) {
      if (vec == 4) OLAP_GR(4, true); else if (vec == 2) OLAP_GR(2, true); else OLAP_GR(1, true);
      return hipGetLastError();
    }
  }
  if (vec == 4) OLAP_GR(4, false); else if (vec == 2) OLAP_GR(2, false); else OLAP_GR(1, false);
#undef OLAP_GR
  return hipGetLastError();
}

template <typename T, bool HS>
static hipError_t gather_reduce_method(int method, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                       const GatherReduce &a, hipStream_t stream) {
  switch (method) {
    case OLAP_SUM: return gather_reduce_vec<T, OLAP_SUM, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_AVERAGE: return gather_reduce_vec<T, OLAP_AVERAGE, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_HIGHEST: return gather_reduce_vec<T, OLAP_HIGHEST, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_LOWEST: return gather_reduce_vec<T, OLAP_LOWEST, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_FIRST: return gather_reduce_vec<T, OLAP_FIRST, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_LAST: return gather_reduce_vec<T, OLAP_LAST, HS>(vec, in, st_in, out, st_out, a, stream);
    case OLAP_PARTIAL_AVERAGE: return gather_reduce_vec<T, OLAP_PARTIAL_AVERAGE, HS>(vec, in, st_in, out, st_out, a, stream);
    default: return gather_reduce_vec<T, OLAP_PRODUCT, HS>(vec, in, st_in, out, st_out, a, stream);
  }
}

template <typename T>
hipError_t Launch<T>::gather_reduce(int method, bool has_status, int vec, const T *in, const int32_t *st_in, T *out,
                                    int32_t *st_out, const GatherReduce &a, hipStream_t stream) {
  if (a.r.total == 0) return hipSuccess;
  return has_status ? gather_reduce_method<T, true>(method, vec, in, st_in, out, st_out, a, stream)
                    : gather_reduce_method<T, false>(method, vec, in, st_in, out, st_out, a, stream);
}

template <typename T>
hipError_t Launch<T>::reorder_brick(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                    const Brick &b, uint64_t n_bricks, hipStream_t stream) {
  if (n_bricks == 0) return hipSuccess;
  if constexpr (sizeof(T) == 4) {
    const bool aligned = (((uintptr_t)in | (uintptr_t)out | (uintptr_t)st_in | (uintptr_t)st_out) & 15u) == 0;
    if (b.quad && aligned) {
      const size_t padded = (size_t)(b.elems + ((b.elems >> 5) << 2)) * 4;
      const size_t lds4 = padded * (has_status ? 2 : 1);
      // above 64 KiB of dynamic LDS the kernel has to be told once (bricks of 10^4 cells with the mask: 90 KiB)
      static PerDeviceFlag raised;
      if (!raised.test_and_set()) {
        hipError_t e1 = hipFuncSetAttribute((const void *)reorder_brick4_kernel<T, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipError_t e2 = hipFuncSetAttribute((const void *)reorder_brick4_kernel<T, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
      }
      unsigned threads4 = kBlock;  // measured: 512 and 1024 lanes lose on the [10]^8 reversal, 512 gains 7 % on a 2-D transpose
      if (const char *e = getenv("OLAP_BRICK_THREADS")) threads4 = (unsigned)atoi(e);
      // groups in flight per lane: 2, 4 and 10 measure the same (the 400-byte runs, not latency, set the pace)
      if (has_status) hipLaunchKernelGGL((reorder_brick4_kernel<T, true, 4>), (unsigned)n_bricks, threads4, lds4, stream, in, st_in, out, st_out, b);
      else hipLaunchKernelGGL((reorder_brick4_kernel<T, false, 4>), (unsigned)n_bricks, threads4, lds4, stream, in, st_in, out, st_out, b);
      return hipGetLastError();
    }
  }
  const size_t cells = lds_pad_host(b.elems);
  const size_t lds = ((cells * sizeof(T) + 15) & ~(size_t)15) + (has_status ? cells * 4 : 0);
  unsigned threads = kBlock;  // larger workgroups for larger bricks bought nothing (tools/sweep.py)
  if (const char *e = getenv("OLAP_BRICK_THREADS")) threads = (unsigned)atoi(e);
  if (lds > 48 * 1024) {  // bricks sized for the 16-byte form, run here because a buffer is not 16 B aligned
    static PerDeviceFlag raised_scalar;
    if (!raised_scalar.test_and_set()) {
      hipError_t e1 = hipFuncSetAttribute((const void *)reorder_brick_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      hipError_t e2 = hipFuncSetAttribute((const void *)reorder_brick_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e1 != hipSuccess) return e1;
      if (e2 != hipSuccess) return e2;
    }
  }
  if (has_status) hipLaunchKernelGGL((reorder_brick_kernel<T, true>), (unsigned)n_bricks, threads, lds, stream, in, st_in, out, st_out, b);
  else hipLaunchKernelGGL((reorder_brick_kernel<T, false>), (unsigned)n_bricks, threads, lds, stream, in, st_in, out, st_out, b);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::load_scatter(bool has_status, int vec, const T *his, const int32_t *his_st, T *mine, int32_t *mine_st,
                                   const Remap &r, hipStream_t stream) {
  if (r.total == 0) return hipSuccess;
  const bool idx32 = r.total * (uint64_t)vec < 0xFFFFFFFFull;
  if (vec == 1 && r.nd >= 1 && (((uintptr_t)his | (uintptr_t)his_st) & 15u) == 0 && !getenv("OLAP_LOAD_NO_RUNS")) {
    // the innermost dimension is remapped: 16 bytes of HIS per lane, one decode per lane
    const uint64_t lanes = (r.total + 16 / sizeof(T) - 1) / (16 / sizeof(T));
    const unsigned g = grid_for(lanes);
    if (has_status) {
      if (idx32) hipLaunchKernelGGL((load_scatter_run_kernel<T, true, uint32_t>), g, kBlock, 0, stream, his, his_st, mine, mine_st, r, r.total);
      else hipLaunchKernelGGL((load_scatter_run_kernel<T, true, uint64_t>), g, kBlock, 0, stream, his, his_st, mine, mine_st, r, r.total);
    } else {
      if (idx32) hipLaunchKernelGGL((load_scatter_run_kernel<T, false, uint32_t>), g, kBlock, 0, stream, his, his_st, mine, mine_st, r, r.total);
      else hipLaunchKernelGGL((load_scatter_run_kernel<T, false, uint64_t>), g, kBlock, 0, stream, his, his_st, mine, mine_st, r, r.total);
    }
    return hipGetLastError();
  }
  const unsigned grid = grid_for(r.total);
#define OLAP_LD(HS, V)                                                                                                              \
  do {                                                                                                                              \
    if (idx32) hipLaunchKernelGGL((load_scatter_kernel<T, HS, V, uint32_t>), grid, kBlock, 0, stream, his, his_st, mine, mine_st, r); \
    else hipLaunchKernelGGL((load_scatter_kernel<T, HS, V, uint64_t>), grid, kBlock, 0, stream, his, his_st, mine, mine_st, r);       \
  } while (0)
  if (has_status) {
    if (vec == 4) OLAP_LD(true, 4); else if (vec == 2) OLAP_LD(true, 2); else OLAP_LD(true, 1);
  } else {
    if (vec == 4) OLAP_LD(false, 4); else if (vec == 2) OLAP_LD(false, 2); else OLAP_LD(false, 1);
  }
#undef OLAP_LD
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::load_permute(bool has_status, const T *his, const int32_t *his_st, T *mine, int32_t *mine_st, const LoadPermute &a,
                                   hipStream_t stream) {
  if (a.n_rows == 0 || a.len == 0) return hipSuccess;
  const uint64_t tiles = (a.n_rows + a.rows_per_tile - 1) / a.rows_per_tile;
  if (tiles >= 0x7FFFFFFFull) return hipErrorInvalidValue;
  const size_t lds = kTileBytes + (mine_st ? kTileBytes / sizeof(T) * 4 : 0) + (size_t)a.len * 4;
  if (has_status) hipLaunchKernelGGL((load_permute_rows_kernel<T, true>), (unsigned)tiles, kBlock, lds, stream, his, his_st, mine, mine_st, a);
  else hipLaunchKernelGGL((load_permute_rows_kernel<T, false>), (unsigned)tiles, kBlock, lds, stream, his, his_st, mine, mine_st, a);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drilldown(bool has_status, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                const DrillDown &a, hipStream_t stream) {
  if (a.total == 0) return hipSuccess;
  const unsigned grid = grid_for(a.total);
  if (has_status) hipLaunchKernelGGL((drilldown_kernel<T, true>), grid, kBlock, 0, stream, in, st_in, out, st_out, a);
  else hipLaunchKernelGGL((drilldown_kernel<T, false>), grid, kBlock, 0, stream, in, st_in, out, st_out, a);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drilldown_rows(bool has_status, int vec, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                     const DrillUpAxis &a, int divide, int use_rounding, uint32_t longest_group, hipStream_t stream) {
  const uint32_t segments = (longest_group + kChildrenPerBlock - 1) / kChildrenPerBlock;
  const uint64_t blocks = a.outer * a.G * a.blocks_per_row * segments;
  if (blocks == 0) return hipSuccess;
  if (blocks >= 0x7FFFFFFFull) return hipErrorInvalidValue;
#define OLAP_DD(HS, V) hipLaunchKernelGGL((drilldown_rows_kernel<T, HS, V>), (unsigned)blocks, kBlock, 0, stream, in, st_in, out, st_out, a, divide, use_rounding, segments)
  if (has_status) {
    if (vec == 4) OLAP_DD(true, 4); else if (vec == 2) OLAP_DD(true, 2); else OLAP_DD(true, 1);
  } else {
    if (vec == 4) OLAP_DD(false, 4); else if (vec == 2) OLAP_DD(false, 2); else OLAP_DD(false, 1);
  }
#undef OLAP_DD
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drilldown_rows_lines(bool has_status, bool any_shift, const T *in, const int32_t *st_in, T *out, int32_t *st_out,
                                           const DrillUpAxis &a, int divide, uint32_t longest_group, hipStream_t stream) {
  constexpr int V = 16 / (int)sizeof(T);
  constexpr uint32_t LINE = 128 / sizeof(T);
  const uint32_t segments = (longest_group + kChildrenPerBlock - 1) / kChildrenPerBlock;
  const uint64_t bpr = (a.inner + LINE + (uint64_t)kBlock * V - 1) / ((uint64_t)kBlock * V);
  const uint64_t blocks = a.outer * a.G * bpr * segments;
  if (blocks == 0) return hipSuccess;
  if (blocks >= 0x7FFFFFFFull) return hipErrorInvalidValue;
#define OLAP_DDL(HS, ANY) hipLaunchKernelGGL((drilldown_rows_lines_kernel<T, HS, V, ANY>), (unsigned)blocks, kBlock, 0, stream, in, st_in, out, st_out, a, divide, segments, (uint32_t)bpr)
  if (has_status) { if (any_shift) OLAP_DDL(true, true); else OLAP_DDL(true, false); }
  else { if (any_shift) OLAP_DDL(false, true); else OLAP_DDL(false, false); }
#undef OLAP_DDL
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::drilldown_scale(bool has_status, const T *in, const int32_t *st_in, T *q, const DrillDownScale &a,
                                      hipStream_t stream) {
  if (a.total == 0) return hipSuccess;
  const unsigned grid = grid_for(a.total);
  if (has_status) hipLaunchKernelGGL((drilldown_scale_kernel<T, true>), grid, kBlock, 0, stream, in, st_in, q, a);
  else hipLaunchKernelGGL((drilldown_scale_kernel<T, false>), grid, kBlock, 0, stream, in, st_in, q, a);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::canonicalize(T *values, int32_t *status, uint64_t n, int def_nan, int use_status,
                                   hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((canonicalize_kernel<T>), grid_stride_for(n), kBlock, 0, stream, values, status, n, def_nan, use_status);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::from_f64(const double *src, T *values, int32_t *status, uint64_t n, int def_nan,
                               hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((from_f64_kernel<T>), grid_stride_for(n), kBlock, 0, stream, src, values, status, n, def_nan);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::to_f64(const T *values, double *dst, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((to_f64_kernel<T>), grid_stride_for(n), kBlock, 0, stream, values, dst, n);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::fill_seeded(T *values, int32_t *status, uint64_t n, uint64_t first, uint32_t seed, double frac,
                                  hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((fill_seeded_kernel<T>), grid_stride_for(n), kBlock, 0, stream, values, status, n, first, seed, frac);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::average_finish(T *values, const int32_t *counts, int32_t *status, uint64_t n, int def_nan,
                                     hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((average_finish_kernel<T>), grid_stride_for(n), kBlock, 0, stream, values, counts, status, n, def_nan);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::total(const T *values, const int32_t *status, uint64_t n, int def_nan, void *workspace, double *total,
                            unsigned long long *count, hipStream_t stream) {
  // workspace: kTotalBlocks TotalPartial slots
  constexpr uint64_t per_block = (uint64_t)kBlock * (16 / sizeof(T)) * 4;  // cells one sweep of a workgroup covers
  // eight workgroups per CU, each streaming one contiguous range (tools/total_probe.py, 10^8 cells, per call with the
  // blocking read of the result: the grid-stride sweep of 4 096 workgroups 102 us; contiguous ranges with 512 / 1 024 /
  // 2 048 / 4 096 workgroups 111 / 94 / 96 / 95 us); more only when a lane's 32-bit count of set cells could overflow
  static const uint64_t units = getenv("OLAP_TOTAL_BLOCKS") ? std::max<uint64_t>(1, std::min<uint64_t>(kTotalBlocks, (uint64_t)atoll(getenv("OLAP_TOTAL_BLOCKS")))) : 2048;
  const uint64_t min_blocks = (n >> 40) + 1;
  const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(kTotalBlocks, std::max<uint64_t>(min_blocks, std::min<uint64_t>(units, (n + per_block - 1) / per_block))));
  const bool vector = (((uintptr_t)values | (uintptr_t)status) & 15u) == 0;
  TotalPartial *partial = (TotalPartial *)workspace;
  if (vector) hipLaunchKernelGGL((total_kernel<T, true>), blocks, kBlock, 0, stream, values, status, n, def_nan, partial);
  else hipLaunchKernelGGL((total_kernel<T, false>), blocks, kBlock, 0, stream, values, status, n, def_nan, partial);
  hipLaunchKernelGGL((total_finish_kernel<0>), 1, kBlock, 0, stream, partial, blocks, total, count);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::compact_count(const T *values, const int32_t *status, uint64_t n, uint64_t chunk, unsigned n_chunks,
                                    int def_nan, unsigned long long *counts, hipStream_t stream) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL((compact_count_kernel<T>), n_chunks, kBlock, 0, stream, values, status, n, chunk, def_nan, counts);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::compact_write(const T *values, const int32_t *status, uint64_t n, uint64_t chunk, unsigned n_chunks,
                                    int def_nan, const unsigned long long *offsets, uint32_t *idx_out, T *val_out,
                                    hipStream_t stream) {
  if (n_chunks == 0) return hipSuccess;
  hipLaunchKernelGGL((compact_write_kernel<T>), n_chunks, kBlock, 0, stream, values, status, n, chunk, def_nan, offsets, idx_out, val_out);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::scatter_sparse(T *values, const uint32_t *idx, const T *vals, uint64_t n_set, uint64_t size,
                                     hipStream_t stream) {
  if (n_set == 0) return hipSuccess;
  hipLaunchKernelGGL((scatter_sparse_kernel<T>), grid_stride_for(n_set), kBlock, 0, stream, values, idx, vals, n_set, size);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::get_cell(const T *values, const int32_t *status, uint64_t index, CellOut *out, hipStream_t stream) {
  hipLaunchKernelGGL((get_cell_kernel<T>), 1, 1, 0, stream, values, status, index, out);
  return hipGetLastError();
}

template <typename T>
hipError_t Launch<T>::set_cell(T *values, int32_t *status, uint64_t index, double value, int is_null, int def_nan,
                               hipStream_t stream) {
  hipLaunchKernelGGL((set_cell_kernel<T>), 1, 1, 0, stream, values, status, index, value, is_null, def_nan);
  return hipGetLastError();
}

#endif  // OLAP_KERNELS_IMPL

}  // namespace olap
