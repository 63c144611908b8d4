// olap_transpose.hip — reorder (axis permutation, /root/reference/src/store/in-memory.js:178-211) of
// 4-byte cells (and, one cell per lane, 8-byte cells: Float64 measures, integer measures of the Node host) as a two-axis
// LDS transpose.
//
// A permutation whose fastest dimension changes reads 4 bytes per cache line if it is written as a
// gather.  Here the dimensions are split into
//   X  the source's fastest dimensions (a contiguous chain in the SOURCE, merged into one linear axis),
//   Y  the destination's fastest dimensions (a contiguous chain in the DESTINATION, merged likewise),
//   B  everything else (batch: one coordinate per workgroup),
// and a workgroup moves one TX x TY tile of the (X, Y) plane: rows of the tile are read along X (TX
// contiguous source cells), parked in LDS, and written along Y (TY contiguous destination cells).
// Offsets are separable — source = base + x + inY[y], destination = base + outX[x] + y — so a tile needs
// two small tables (TX + TY entries, decoded by the workgroup itself from the tile's origin), not one
// entry per cell, tiles need not line up with dimension boundaries (512-byte runs on both sides whatever
// the extents are), and ragged edges are two wave-uniform bounds.
//
// LDS layout: cell (x, y) at x * (TY + 1) + y, every access 4 bytes wide.  The pitch TY + 1 = 1 (mod 32)
// makes the bank of a cell (x + y) mod 32, so
//   * 16-byte global accesses (lane = 4 adjacent cells) are conflict-free when a 32-lane group covers
//     8 quads x 4 lines (4q + j + line distinct), on the way in and on the way out alike;
//   * 4-byte global accesses (lane = one cell, 64 adjacent cells per wave) are conflict-free as they are.
// The 16-byte form of each side is used when every row of every tile starts 16-byte aligned on that
// side; otherwise that side moves 4 bytes per lane (cubes with odd extents).
//
// HBM-bound: 4 B read + 4 B written per cell.
//
// Measured and dropped in round 2 (profiles/transpose_phases_r02.txt, transpose_tiles_r02.txt, transpose_order_r02.txt):
// a variant that kept the tile in registers and passed it through LDS in halves or quarters (8 instead of ~3
// workgroups resident per CU) — every phase of a workgroup's life took twice as long and the kernel the same time,
// i.e. the memory system, not latency or residency, bounds this access pattern; 128 x 128, 64 x 256 and 128 x 256
// tiles with 512 / 1024 lanes (512-byte reads, 1 KB writes: within +-5 %); 16-byte accesses at cell-aligned addresses
// instead of the 4-byte lanes on odd extents (free in the streaming kernels, 25 % SLOWER here: 245 against 195 us).
//
// Round 3: WHAT bounds it.  tools/pattern_ceiling.hip moves the same tiles with the same addresses, run lengths and
// walk but no LDS, no barrier and full occupancy (registers to registers, the cells arrive scrambled): that pattern
// ceiling is 148-190 us on the shapes below (0.53-0.68 of 8 TB/s; its read side alone runs at streaming speed,
// 0.75-0.85, its write side — 512-byte runs a destination row apart — at 0.42-0.66), and this kernel reaches 0.86-1.0
// of it on the same box (profiles/pattern_ceiling_r03.txt, transpose_probe_r03.txt).  The counters
// (profiles/traffic_r03_kernels.json) rule the other suspects out: traffic 1.07-1.12 x algorithmic, no LDS bank
// conflicts, address translation misses <= 2.5 % of the requests ([10]^8 reversed; 0.0-0.4 % elsewhere), and NO
// back-pressure from the memory side of the L2 (write-request / read-credit stalls 0.00-0.03 of its busy cycles, against
// ~0.2 for the streaming kernels that saturate HBM).  Measured and dropped on that evidence: workgroups that walk a
// sequence of tiles with the NEXT tile's loads in flight (in registers) while the current tile leaves — twice the loads
// in flight per CU for the same LDS: equal on [10^4,10^4] (169 against 170 us), 15-40 % SLOWER everywhere else
// ([3652,27400] 213 against 177 us; profiles/transpose_stream_ab_r03.txt): once more many short-lived workgroups,
// dispatched in tile order, beat long-lived ones that drift apart.
#include <hip/hip_runtime.h>

#include "olap_device.hpp"
#include "olap_internal.hpp"

using namespace olap;

// tools/transpose_probe.py prints per-phase times when the library is built with OLAP_XY_PROBE: lane 0 of every
// workgroup records the 100 MHz wall clock at its phase boundaries (not compiled into the product)
#ifdef OLAP_XY_PROBE
__device__ unsigned long long g_xy_probe[1 << 20];
#define XY_PROBE(i)                                                                                  \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < (1u << 17)) g_xy_probe[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
extern "C" int olap_diag_xy_probe(unsigned long long *host, unsigned long long n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xy_probe), n * sizeof(unsigned long long));
}
#else
#define XY_PROBE(i) do { } while (0)
#endif

namespace {

// W: the cell as bits — uint32_t, or uint64_t for 8-byte cells (one cell per lane on both sides: 512-byte runs per
// wavefront; the 16-byte lane forms VIN / VOUT are the 4-byte cells')
template <typename W, int TX, int TY, bool VIN, bool VOUT>
__global__ __launch_bounds__(kBlock) void transpose_xy_kernel(const W *__restrict__ in, W *__restrict__ out,
                                                              int32_t *__restrict__ st_out, const TransposeXY t) {
  constexpr int P = TY + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  W *tile = reinterpret_cast<W *>(tile_raw);  // TX * P cells
  __shared__ uint64_t out_x[TX];  // destination offset of tile column x (relative to the tile's base)
  __shared__ uint64_t in_y[TY];   // source offset of tile row y

  // tile -> (batch coordinate, super-tile, tile inside it).  Workgroups that run at the same time (consecutive
  // logical ids, one XCD) cover a kSuper x kSuper block of tiles: their row pieces are neighbours in the source
  // (along X) AND in the destination (along Y), so DRAM pages opened for one tile serve its neighbours
  const uint32_t kSuper = (uint32_t)t.super;
  uint64_t c = xcd_contiguous(blockIdx.x, gridDim.x);
  const uint32_t in_super = (uint32_t)(c % (kSuper * kSuper));
  c /= kSuper * kSuper;
  const uint64_t sx_n = (t.tiles_x + kSuper - 1) / kSuper, sy_n = (t.tiles_y + kSuper - 1) / kSuper;
  uint32_t tx, ty;
  if (t.y_first) {  // neighbouring workgroups write neighbouring pieces of the same destination rows
    ty = (uint32_t)(c % sy_n) * kSuper + in_super / kSuper;
    c /= sy_n;
    tx = (uint32_t)(c % sx_n) * kSuper + in_super % kSuper;
    c /= sx_n;
  } else {  // ... read neighbouring pieces of the same source rows
    tx = (uint32_t)(c % sx_n) * kSuper + in_super % kSuper;
    c /= sx_n;
    ty = (uint32_t)(c % sy_n) * kSuper + in_super / kSuper;
    c /= sy_n;
  }
  if (tx >= t.tiles_x || ty >= t.tiles_y) return;  // the grid is padded to whole super-tiles
  XY_PROBE(0);
  uint64_t base_in = 0, base_out = 0;
#pragma unroll
  for (int d = 0; d < kTransposeMaxBatch; ++d) {
    if (d < t.nb) {
      const uint64_t digit = c % t.len_b[d];
      c /= t.len_b[d];
      base_in += digit * t.in_stride_b[d];
      base_out += digit * t.out_stride_b[d];
    }
  }
  const uint64_t x0 = (uint64_t)tx * TX, y0 = (uint64_t)ty * TY;
  const uint32_t nx = (uint32_t)((t.lx - x0) < (uint64_t)TX ? (t.lx - x0) : (uint64_t)TX);
  const uint32_t ny = (uint32_t)((t.ly - y0) < (uint64_t)TY ? (t.ly - y0) : (uint64_t)TY);
  base_in += x0;   // the X chain is contiguous in the source
  base_out += y0;  // the Y chain is contiguous in the destination

  // per-tile offset tables: digits of the merged coordinate, fastest dimension first (32-bit arithmetic: the
  // plan keeps both merged axes below 2^31).  Rows are needed at once; the column table is filled while the
  // tile's loads are in flight.
  auto decode_row = [&](uint32_t i) {
    uint32_t v = (uint32_t)y0 + i;
    uint64_t off = 0;
#pragma unroll
    for (int d = 0; d < kTransposeMaxAxis; ++d)
      if (d < t.ny) {
        const uint32_t qd = v / t.len_y[d];
        off += (uint64_t)(v - qd * t.len_y[d]) * t.in_stride_y[d];
        v = qd;
      }
    in_y[i] = off;
  };
  auto decode_col = [&](uint32_t i) {
    uint32_t v = (uint32_t)x0 + i;
    uint64_t off = 0;
#pragma unroll
    for (int d = 0; d < kTransposeMaxAxis; ++d)
      if (d < t.nx) {
        const uint32_t qd = v / t.len_x[d];
        off += (uint64_t)(v - qd * t.len_x[d]) * t.out_stride_x[d];
        v = qd;
      }
    out_x[i] = off;
  };
  for (uint32_t i = threadIdx.x; i < (uint32_t)TY; i += kBlock) decode_row(i);
  __syncthreads();
  XY_PROBE(1);

  const W *src = in + base_in;
  // ---- in: rows along X
  if constexpr (VIN && sizeof(W) == 4) {
    // a 32-lane group = 8 quads x 4 rows; the workgroup's 8 groups tile QX quad-blocks x RY row-blocks
    constexpr int QX = TX / 32;          // 32-cell blocks across a row
    constexpr int RY = 8 / QX;           // row blocks per pass
    constexpr int ROWS = RY * 4;         // rows per pass
    const uint32_t g = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t q = (g % QX) * 8 + (l & 7);          // quad index in the row
    const uint32_t r0 = (g / QX) * 4 + (l >> 3);        // row within the pass
    constexpr int PASSES = TY / ROWS;
    constexpr int UB = PASSES < 8 ? PASSES : 8;
    for (int p0 = 0; p0 < PASSES; p0 += UB) {
      Vec<uint32_t, 4> v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint32_t y = (p0 + u) * ROWS + r0;
        if (y < ny && 4 * q < nx) v[u] = load_stream<uint32_t, 4>(src + in_y[y] + 4 * q);
      }
      if (p0 == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)TX; i += kBlock) decode_col(i);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint32_t y = (p0 + u) * ROWS + r0;
        if (y < ny && 4 * q < nx) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tile[(4 * q + j) * P + y] = v[u].v[j];
        }
      }
    }
  } else {
    constexpr int ROWS = kBlock / TX > 0 ? kBlock / TX : 1;  // rows per pass (TX <= 256)
    const uint32_t x = threadIdx.x % TX, r0 = threadIdx.x / TX;
    constexpr int PASSES = TY / ROWS;
    constexpr int UB = 8;
    for (int p0 = 0; p0 < PASSES; p0 += UB) {
      W v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint32_t y = (p0 + u) * ROWS + r0;
        if (p0 + u < PASSES && y < ny && x < nx) v[u] = __builtin_nontemporal_load(src + in_y[y] + x);
      }
      if (p0 == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)TX; i += kBlock) decode_col(i);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint32_t y = (p0 + u) * ROWS + r0;
        if (p0 + u < PASSES && y < ny && x < nx) tile[x * P + y] = v[u];
      }
    }
  }
  __syncthreads();
  XY_PROBE(2);

  // ---- out: rows along Y
  W *dst = out + base_out;
  int32_t *sdst = st_out ? st_out + base_out : nullptr;
  auto status_of = [&](W bits) -> int32_t {
    bool is_default;
    constexpr W kAbs = (W)(~(W)0 >> 1);                                     // everything but the sign bit
    constexpr W kInf = sizeof(W) == 4 ? (W)0x7F800000u : (W)0x7FF0000000000000ull;
    switch (t.default_test) {
      case 0: is_default = bits == 0u; break;                               // integer cells, 0 default
      case 1: is_default = (bits & kAbs) == 0u; break;                      // float cells, 0 default (+0 and -0)
      case 2: is_default = (bits & kAbs) > kInf; break;                     // float cells, NaN default
      default: is_default = false; break;
    }
    return is_default ? 0 : OLAP_STATUS_SET;
  };
  if constexpr (VOUT && sizeof(W) == 4) {
    constexpr int QY = TY / 32;
    constexpr int RX = 8 / QY;
    constexpr int ROWS = RX * 4;
    const uint32_t g = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t q = (g % QY) * 8 + (l & 7);
    const uint32_t r0 = (g / QY) * 4 + (l >> 3);
    constexpr int PASSES = TX / ROWS;
    for (int p = 0; p < PASSES; ++p) {
      const uint32_t x = p * ROWS + r0;
      if (x < nx && 4 * q < ny) {
        Vec<uint32_t, 4> v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v.v[j] = tile[x * P + 4 * q + j];
        if (t.cached_stores) store_vec<uint32_t, 4>(dst + out_x[x] + 4 * q, v);
        else store_stream<uint32_t, 4>(dst + out_x[x] + 4 * q, v);
        if (sdst) {
          Vec<int32_t, 4> s;
#pragma unroll
          for (int j = 0; j < 4; ++j) s.v[j] = status_of(v.v[j]);
          store_stream<int32_t, 4>(sdst + out_x[x] + 4 * q, s);
        }
      }
    }
  } else {
    constexpr int ROWS = kBlock / TY > 0 ? kBlock / TY : 1;
    const uint32_t y = threadIdx.x % TY, r0 = threadIdx.x / TY;
    constexpr int PASSES = TX / ROWS;
    for (int p = 0; p < PASSES; ++p) {
      const uint32_t x = p * ROWS + r0;
      if (x < nx && y < ny) {
        const W v = tile[x * P + y];
        if (t.cached_stores) dst[out_x[x] + y] = v;
        else __builtin_nontemporal_store(v, dst + out_x[x] + y);
        if (sdst) __builtin_nontemporal_store(status_of(v), sdst + out_x[x] + y);
      }
    }
  }
  XY_PROBE(3);
}

template <typename W, int TX, int TY, bool VIN, bool VOUT>
hipError_t launch_one(const TransposeXY &t, const W *in, W *out, int32_t *st_out, unsigned grid, hipStream_t stream) {
  constexpr size_t lds = (size_t)TX * (TY + 1) * sizeof(W);
  if (lds > 48 * 1024) {
    static PerDeviceFlag raised;  // (per instantiation, per device)
    if (!raised.test_and_set()) {
      hipError_t e = hipFuncSetAttribute((const void *)transpose_xy_kernel<W, TX, TY, VIN, VOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL((transpose_xy_kernel<W, TX, TY, VIN, VOUT>), grid, kBlock, lds, stream, in, out, st_out, t);
  return hipGetLastError();
}

static bool grid_of(const TransposeXY &t, unsigned *grid, hipError_t *err) {
  const uint64_t kSuper = (uint64_t)t.super;
  const uint64_t sx = (t.tiles_x + kSuper - 1) / kSuper, sy = (t.tiles_y + kSuper - 1) / kSuper;
  const uint64_t tiles = sx * sy * kSuper * kSuper * t.batch;
  *err = hipSuccess;
  if (t.tiles_x * t.tiles_y * t.batch == 0) return false;
  if (tiles > 0x7FFFFFFFull) {
    *err = hipErrorInvalidValue;
    return false;
  }
  *grid = (unsigned)tiles;
  return true;
}

template <int TX, int TY>
hipError_t launch_tile(const TransposeXY &t, const uint32_t *in, uint32_t *out, int32_t *st_out, bool vin, bool vout, hipStream_t stream) {
  const uint64_t kSuper = (uint64_t)t.super;
  const uint64_t sx = (t.tiles_x + kSuper - 1) / kSuper, sy = (t.tiles_y + kSuper - 1) / kSuper;
  const uint64_t tiles = sx * sy * kSuper * kSuper * t.batch;
  if (t.tiles_x * t.tiles_y * t.batch == 0) return hipSuccess;
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)tiles;
  if (vin && vout) return launch_one<uint32_t, TX, TY, true, true>(t, in, out, st_out, grid, stream);
  if (vin) return launch_one<uint32_t, TX, TY, true, false>(t, in, out, st_out, grid, stream);
  if (vout) return launch_one<uint32_t, TX, TY, false, true>(t, in, out, st_out, grid, stream);
  return launch_one<uint32_t, TX, TY, false, false>(t, in, out, st_out, grid, stream);
}

}  // namespace

hipError_t launch_transpose_xy(const TransposeXY &t, int cell_bytes, const void *in, void *out, int32_t *st_out, bool aligned16, hipStream_t stream) {
  if (cell_bytes == 8) {  // one 8-byte cell per lane on both sides (the plan sets 64 x 64 tiles: 33 KB of LDS)
    unsigned grid = 0;
    hipError_t e;
    if (!grid_of(t, &grid, &e)) return e;
    if (t.tx == 64 && t.ty == 64) return launch_one<uint64_t, 64, 64, false, false>(t, (const uint64_t *)in, (uint64_t *)out, st_out, grid, stream);
    if (t.tx == 128 && t.ty == 64) return launch_one<uint64_t, 128, 64, false, false>(t, (const uint64_t *)in, (uint64_t *)out, st_out, grid, stream);
    if (t.tx == 64 && t.ty == 128) return launch_one<uint64_t, 64, 128, false, false>(t, (const uint64_t *)in, (uint64_t *)out, st_out, grid, stream);
    return hipErrorInvalidValue;
  }
  const bool vin = t.vec_in && aligned16, vout = t.vec_out && aligned16;
  const uint32_t *src = (const uint32_t *)in;
  uint32_t *dst = (uint32_t *)out;
  if (t.tx == 64 && t.ty == 64) return launch_tile<64, 64>(t, src, dst, st_out, vin, vout, stream);
  if (t.tx == 128 && t.ty == 64) return launch_tile<128, 64>(t, src, dst, st_out, vin, vout, stream);
  if (t.tx == 64 && t.ty == 128) return launch_tile<64, 128>(t, src, dst, st_out, vin, vout, stream);
  if (t.tx == 128 && t.ty == 128) return launch_tile<128, 128>(t, src, dst, st_out, vin, vout, stream);
  return hipErrorInvalidValue;
}
