// olap_transpose.hip — reorder (axis permutation, /root/reference/src/store/in-memory.js:178-211) of
// 4-byte cells as a two-axis LDS transpose.
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
// Both sides move 16 bytes per lane: at 16-byte-aligned addresses when every row of every tile starts aligned on
// that side, at cell-aligned addresses otherwise (cubes with odd extents).
//
// HBM-bound: 4 B read + 4 B written per cell.
#include <hip/hip_runtime.h>

#include "olap_device.hpp"
#include "olap_internal.hpp"

using namespace olap;

// tools/transpose_probe.py --phases builds this file with OLAP_XY_PROBE: lane 0 of every workgroup records the 100 MHz
// wall clock at its phase boundaries (not compiled into the product)
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

// AIN / AOUT: every row of every tile starts 16-byte aligned on that side (aligned 16-byte accesses); otherwise the
// same 16-byte accesses go to cell-aligned addresses (cubes with odd extents: gfx950 takes a dwordx4 at any 4-byte
// boundary, free on the load side and ~7 % on the store side, tools/unaligned_probe.hip) and only a quad that runs over
// the tile's ragged edge moves cell by cell.
// H: the tile leaves in H phases of TX / H source columns each, through an LDS buffer of that size.  The tile's cells
// wait in REGISTERS (32 per lane) from the moment their loads are issued; LDS is only the transposing step.  A
// workgroup lives ~10 us (tools/transpose_probe.py with an OLAP_XY_PROBE build: 4-9 us until its rows have arrived,
// 3-5 us issuing its stores), so what bounds the kernel is how many tiles a CU keeps in flight: a whole 64 x 128 tile
// in LDS (33 KB) allows 4 workgroups per CU, half of it 8 — twice the loads in flight.
template <int TX, int TY, bool AIN, bool AOUT, int H, int NT>
__global__ __launch_bounds__(NT) void transpose_xy_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              int32_t *__restrict__ st_out, const TransposeXY t) {
  constexpr int P = TY + 1;
  constexpr int TXH = TX / H;
  extern __shared__ __attribute__((aligned(16))) uint32_t tile[];  // TXH * P cells
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
  for (uint32_t i = threadIdx.x; i < (uint32_t)TY; i += NT) decode_row(i);
  __syncthreads();
  XY_PROBE(1);

  // ---- in: rows along X, the whole tile into registers.  A 32-lane group = 8 quads x 4 rows; the workgroup's
  // groups tile QX quad-blocks x RY row-blocks
  const uint32_t *src = in + base_in;
  constexpr int G32 = NT / 32;         // 32-lane groups of the workgroup
  constexpr int QX = TX / 32;          // 32-cell blocks across a row
  constexpr int RY = G32 / QX;         // row blocks per pass
  constexpr int ROWS = RY * 4;         // rows per pass
  constexpr int PASSES = TY / ROWS;    // 16-byte loads per lane (8 for 64 x 128)
  const uint32_t g = threadIdx.x >> 5, l = threadIdx.x & 31;
  const uint32_t q = (g % QX) * 8 + (l & 7);          // quad index in the row
  const uint32_t r0 = (g / QX) * 4 + (l >> 3);        // row within the pass
  const bool whole_in = 4 * q + 3 < nx;               // (a quad that runs over the ragged edge goes cell by cell)
  Vec<uint32_t, 4> v[PASSES];
#pragma unroll
  for (int u = 0; u < PASSES; ++u) {
    const uint32_t y = u * ROWS + r0;
    if (y < ny && whole_in) {
      const uint32_t *p = src + in_y[y] + 4 * q;
      v[u] = AIN ? load_stream<uint32_t, 4>(p) : load_stream_cell_aligned<uint32_t, 4>(p);
    }
  }
  if (!whole_in && 4 * q < nx) {
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const uint32_t y = u * ROWS + r0;
      if (y < ny) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * q + j < nx) v[u].v[j] = __builtin_nontemporal_load(src + in_y[y] + 4 * q + j);
      }
    }
  }
  for (uint32_t i = threadIdx.x; i < (uint32_t)TX; i += NT) decode_col(i);

  uint32_t *dst = out + base_out;
  int32_t *sdst = st_out ? st_out + base_out : nullptr;
  auto status_of = [&](uint32_t bits) -> int32_t {
    bool is_default;
    switch (t.default_test) {
      case 0: is_default = bits == 0u; break;                               // integer cells, 0 default
      case 1: is_default = (bits << 1) == 0u; break;                        // float cells, 0 default (+0 and -0)
      case 2: is_default = (bits & 0x7FFFFFFFu) > 0x7F800000u; break;       // float cells, NaN default
      default: is_default = false; break;
    }
    return is_default ? 0 : OLAP_STATUS_SET;
  };
  constexpr int QY = TY / 32;
  constexpr int RX = G32 / QY;
  constexpr int ROWS_O = RX * 4;
  constexpr int PASSES_O = TXH / ROWS_O;
  static_assert(TXH % ROWS_O == 0 && TXH % 4 == 0, "a phase is whole passes of whole quads");
  const uint32_t qo = (g % QY) * 8 + (l & 7);
  const uint32_t ro = (g / QY) * 4 + (l >> 3);
  const bool whole_out = 4 * qo + 3 < ny;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    if (h > 0) __syncthreads();  // the previous phase has left LDS
    // LDS: cell (x, y) at (x - h TXH) * P + y; P = 1 (mod 32) makes the bank (x + y) mod 32: a 32-lane group's 8 quads
    // x 4 rows are conflict-free on the way in, its 8 quads x 4 columns on the way out
    if ((4 * q) / TXH == (uint32_t)h && 4 * q < nx) {
#pragma unroll
      for (int u = 0; u < PASSES; ++u) {
        const uint32_t y = u * ROWS + r0;
        if (y < ny) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tile[(4 * q + j - h * TXH) * P + y] = v[u].v[j];
        }
      }
    }
    __syncthreads();
    if (h == 0) XY_PROBE(2);
    // ---- out: rows along Y
#pragma unroll
    for (int p = 0; p < PASSES_O; ++p) {
      const uint32_t xl = p * ROWS_O + ro, x = h * TXH + xl;
      if (x < nx && 4 * qo < ny) {
        Vec<uint32_t, 4> w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w.v[j] = tile[xl * P + 4 * qo + j];
        uint32_t *d = dst + out_x[x] + 4 * qo;
        if (whole_out) {
          if constexpr (AOUT) store_stream<uint32_t, 4>(d, w);
          else store_stream_cell_aligned<uint32_t, 4>(d, w);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * qo + j < ny) __builtin_nontemporal_store(w.v[j], d + j);
        }
        if (sdst) {
          Vec<int32_t, 4> sw;
#pragma unroll
          for (int j = 0; j < 4; ++j) sw.v[j] = status_of(w.v[j]);
          int32_t *sd = sdst + out_x[x] + 4 * qo;
          if (whole_out) {
            if constexpr (AOUT) store_stream<int32_t, 4>(sd, sw);
            else store_stream_cell_aligned<int32_t, 4>(sd, sw);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * qo + j < ny) __builtin_nontemporal_store(sw.v[j], sd + j);
          }
        }
      }
    }
  }
  XY_PROBE(3);
}

template <int TX, int TY, bool AIN, bool AOUT, int H, int NT>
hipError_t launch_one(const TransposeXY &t, const uint32_t *in, uint32_t *out, int32_t *st_out, unsigned grid, hipStream_t stream) {
  constexpr size_t lds = (size_t)(TX / H) * (TY + 1) * sizeof(uint32_t);
  hipLaunchKernelGGL((transpose_xy_kernel<TX, TY, AIN, AOUT, H, NT>), grid, NT, lds, stream, in, out, st_out, t);
  return hipGetLastError();
}

template <int TX, int TY, int H, int NT>
hipError_t launch_tile(const TransposeXY &t, const uint32_t *in, uint32_t *out, int32_t *st_out, bool ain, bool aout, hipStream_t stream) {
  const uint64_t kSuper = (uint64_t)t.super;
  const uint64_t sx = (t.tiles_x + kSuper - 1) / kSuper, sy = (t.tiles_y + kSuper - 1) / kSuper;
  const uint64_t tiles = sx * sy * kSuper * kSuper * t.batch;
  if (t.tiles_x * t.tiles_y * t.batch == 0) return hipSuccess;
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)tiles;
  if (ain && aout) return launch_one<TX, TY, true, true, H, NT>(t, in, out, st_out, grid, stream);
  if (ain) return launch_one<TX, TY, true, false, H, NT>(t, in, out, st_out, grid, stream);
  if (aout) return launch_one<TX, TY, false, true, H, NT>(t, in, out, st_out, grid, stream);
  return launch_one<TX, TY, false, false, H, NT>(t, in, out, st_out, grid, stream);
}

}  // namespace

// Tile shapes: 64 x 64 and 64 x 128 with 256 lanes (the whole tile or half of it in LDS at a time); 128 x 128 and
// 64 x 256 with 512 lanes, 128 x 256 with 1024 — longer runs per row, a quarter or an eighth of the tile in LDS at a time
hipError_t launch_transpose_xy(const TransposeXY &t, const void *in, void *out, int32_t *st_out, bool aligned16, hipStream_t stream) {
  const bool ain = t.vec_in && aligned16, aout = t.vec_out && aligned16;
  const uint32_t *src = (const uint32_t *)in;
  uint32_t *dst = (uint32_t *)out;
  if (t.tx == 64 && t.ty == 64) return t.phases == 1 ? launch_tile<64, 64, 1, 256>(t, src, dst, st_out, ain, aout, stream) : launch_tile<64, 64, 2, 256>(t, src, dst, st_out, ain, aout, stream);
  if (t.tx == 64 && t.ty == 128) return t.phases == 1 ? launch_tile<64, 128, 1, 256>(t, src, dst, st_out, ain, aout, stream) : launch_tile<64, 128, 2, 256>(t, src, dst, st_out, ain, aout, stream);
  if (t.tx == 128 && t.ty == 128) return launch_tile<128, 128, 4, 512>(t, src, dst, st_out, ain, aout, stream);
  if (t.tx == 64 && t.ty == 256) return launch_tile<64, 256, 4, 512>(t, src, dst, st_out, ain, aout, stream);
  if (t.tx == 128 && t.ty == 256) return launch_tile<128, 256, 8, 1024>(t, src, dst, st_out, ain, aout, stream);
  return hipErrorInvalidValue;
}
