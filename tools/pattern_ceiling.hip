// tools/pattern_ceiling.hip — developer tool: what the memory system gives for the ACCESS PATTERN of the two-axis
// transpose (csrc/olap_transpose.hip), without the transpose: a workgroup moves a tile of TY row pieces of TX cells
// (TX*4 bytes each, one source row apart) and writes TX row pieces of TY cells (one destination row apart) — the same
// addresses, run lengths and number of bytes in flight as transpose_xy_kernel, but the cells go from registers to
// registers' own slots (no LDS, no barrier; the data arrive scrambled, which a ceiling does not mind).
//   mode read   only the tile's loads (the read side of the pattern alone)
//   mode write  only the tile's stores
//   mode copy   loads + stores at the SAME tile of an identically shaped destination (strided on both sides, no transposition)
//   mode trans  loads + stores at the transposed tile (the transpose's pattern)
// Next to them: the plain streaming copy of the same bytes.  Usage: pattern_ceiling [R C]...
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
  const uint32_t q = n / 8, r = n % 8, x = b % 8, i = b / 8;
  return x * q + (x < r ? x : r) + i;
}

enum { READ = 0, WRITE = 1, COPY = 2, TRANS = 3 };

// source [R][C] cells, destination [C][R] (TRANS) or [R][C] (COPY).  Tile: TX cells along C, TY rows.
// OCC = waves per SIMD the compiler must leave room for (the product's transpose holds its tile in LDS: 33 KB per
// 64 x 128 tile = 4 workgroups = 4 waves per SIMD; without LDS up to 8 fit)
template <int TX, int TY, int MODE, int ORDER, int OCC>
__global__ __launch_bounds__(256, OCC) void pattern(const float *__restrict__ in, float *__restrict__ out, uint32_t R, uint32_t C, float *sink) {
  constexpr int F4 = TX * TY / 4;      // 16-byte groups per tile
  constexpr int U = F4 / 256;          // per lane
  const uint32_t tiles_x = (C + TX - 1) / TX, tiles_y = (R + TY - 1) / TY;
  uint32_t tx, ty;
  if (ORDER == 2) {  // the product's walk: XCD-contiguous ids, 4 x 4 blocks of tiles, Y fastest
    uint32_t c = xcd_contiguous(blockIdx.x, gridDim.x);
    const uint32_t in_super = c % 16;
    c /= 16;
    const uint32_t sy_n = (tiles_y + 3) / 4;
    ty = (c % sy_n) * 4 + in_super / 4;
    tx = (c / sy_n) * 4 + in_super % 4;
  } else if (ORDER == 1) {
    ty = blockIdx.x % tiles_y;
    tx = blockIdx.x / tiles_y;
  } else {
    tx = blockIdx.x % tiles_x;
    ty = blockIdx.x / tiles_x;
  }
  if (tx >= tiles_x || ty >= tiles_y) return;
  const uint64_t x0 = (uint64_t)tx * TX, y0 = (uint64_t)ty * TY;
  f4 v[U];
  float acc = 0.f;
  if (MODE != WRITE) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t idx = threadIdx.x + u * 256;
      const uint32_t piece = idx / (TX / 4), within = idx % (TX / 4);
      const uint64_t y = y0 + piece, x = x0 + within * 4;
      if (y < R && x + 4 <= C) v[u] = __builtin_nontemporal_load((const f4u *)(in + y * C + x));
      else v[u] = f4{0, 0, 0, 0};
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = f4{1.f, 2.f, 3.f, (float)u};
  }
  if (MODE == READ) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 12345.678f) sink[0] = acc;
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t idx = threadIdx.x + u * 256;
    if (MODE == TRANS || MODE == WRITE) {  // TX destination rows, TY cells each, one destination row (R cells) apart
      const uint32_t piece = idx / (TY / 4), within = idx % (TY / 4);
      const uint64_t x = x0 + piece, y = y0 + within * 4;
      if (x < C && y + 4 <= R) __builtin_nontemporal_store(v[u], (f4u *)(out + x * R + y));
    } else {  // COPY: the same tile of a [R][C] destination
      const uint32_t piece = idx / (TX / 4), within = idx % (TX / 4);
      const uint64_t y = y0 + piece, x = x0 + within * 4;
      if (y < R && x + 4 <= C) __builtin_nontemporal_store(v[u], (f4u *)(out + y * C + x));
    }
  }
}

__global__ __launch_bounds__(256) void stream_copy(const f4 *__restrict__ in, f4 *__restrict__ out, uint64_t n4) {
  for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < n4; base += (uint64_t)gridDim.x * 1024) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) v[u] = __builtin_nontemporal_load(in + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) __builtin_nontemporal_store(v[u], out + i);
    }
  }
}

static hipEvent_t e0, e1;
template <typename F>
static float timed(F fn, int iters = 10, int rounds = 5) {
  fn();
  fn();
  std::vector<float> ms;
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / iters);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2] * 1e3f;  // us
}

template <int TX, int TY, int ORDER, int OCC>
static void run_tile(const float *in, float *out, float *sink, uint32_t R, uint32_t C) {
  const uint32_t tiles_x = (C + TX - 1) / TX, tiles_y = (R + TY - 1) / TY;
  const unsigned grid = ORDER == 2 ? ((tiles_x + 3) / 4) * ((tiles_y + 3) / 4) * 16 : tiles_x * tiles_y;
  const double bytes = (double)R * C * 4;
  const float r = timed([&] { hipLaunchKernelGGL((pattern<TX, TY, READ, ORDER, OCC>), grid, 256, 0, 0, in, out, R, C, sink); });
  const float w = timed([&] { hipLaunchKernelGGL((pattern<TX, TY, WRITE, ORDER, OCC>), grid, 256, 0, 0, in, out, R, C, sink); });
  const float c = timed([&] { hipLaunchKernelGGL((pattern<TX, TY, COPY, ORDER, OCC>), grid, 256, 0, 0, in, out, R, C, sink); });
  const float t = timed([&] { hipLaunchKernelGGL((pattern<TX, TY, TRANS, ORDER, OCC>), grid, 256, 0, 0, in, out, R, C, sink); });
  printf("  tile %3d x %3d (reads %4d B, writes %4d B) order %d, %d waves/SIMD:  read %6.1f us %.3f | write %6.1f us %.3f | strided copy %6.1f us %.3f | transposed %6.1f us %.3f\n",
         TX, TY, TX * 4, TY * 4, ORDER, OCC, r, bytes / (r * 1e-6) / 8e12, w, bytes / (w * 1e-6) / 8e12, c, 2 * bytes / (c * 1e-6) / 8e12, t,
         2 * bytes / (t * 1e-6) / 8e12);
  fflush(stdout);
}

int main(int argc, char **argv) {
  std::vector<std::pair<uint32_t, uint32_t>> shapes;
  for (int i = 1; i + 1 < argc; i += 2) shapes.push_back({(uint32_t)atoi(argv[i]), (uint32_t)atoi(argv[i + 1])});
  if (shapes.empty()) shapes = {{10000, 10000}, {100000, 1000}, {1000, 100000}, {3652, 27400}, {27400, 3652}};
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float *sink;
  CK(hipMalloc(&sink, 4));
  for (auto [R, C] : shapes) {
    const uint64_t n = (uint64_t)R * C;
    float *in, *out;
    CK(hipMalloc(&in, n * 4 + 64));
    CK(hipMalloc(&out, n * 4 + 64));
    CK(hipMemset(in, 1, n * 4));
    const float s = timed([&] { hipLaunchKernelGGL(stream_copy, 8192, 256, 0, 0, (const f4 *)in, (f4 *)out, n / 4); });
    printf("[%u, %u] (%.0f MB each way): streaming copy %6.1f us %.3f of 8 TB/s\n", R, C, n * 4 / 1e6, s, 2.0 * n * 4 / (s * 1e-6) / 8e12);
    run_tile<64, 128, 2, 4>(in, out, sink, R, C);  // the product's tile, walk and residency
    run_tile<64, 128, 2, 8>(in, out, sink, R, C);
    run_tile<64, 128, 0, 8>(in, out, sink, R, C);
    run_tile<128, 64, 2, 8>(in, out, sink, R, C);
    run_tile<128, 128, 2, 4>(in, out, sink, R, C);
    run_tile<64, 256, 2, 4>(in, out, sink, R, C);
    run_tile<256, 64, 2, 4>(in, out, sink, R, C);
    run_tile<128, 256, 2, 2>(in, out, sink, R, C);
    run_tile<256, 128, 2, 2>(in, out, sink, R, C);
    run_tile<64, 512, 2, 2>(in, out, sink, R, C);
    run_tile<512, 64, 2, 2>(in, out, sink, R, C);
    CK(hipFree(in));
    CK(hipFree(out));
  }
  return 0;
}
