// tools/tile_probe.hip — developer tool (not part of the product): where a workgroup of drillup_tile_kernel spends
// its life.  Builds the kernel header with OLAP_TILE_PROBE (lane 0 of every workgroup records the 100 MHz wall clock
// at its phase boundaries) and prints, per table mode, the mean time from start to "loads arrived + tables in LDS"
// (P1, MODE 3 only), to "tile staged" (P2) and to "first output stored" (P3), the mean workgroup lifetime and the
// average number of resident workgroups per CU that these lifetimes imply.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOLAP_TILE_PROBE -I include -I olap-in-memory_amd/csrc tools/tile_probe.hip -o tools/tile_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "olap_kernels.hpp"

using namespace olap;

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__global__ void fill_kernel(float *p, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = 0.5f + (float)((i * 2654435761ull) & 1023) / 1024.f;
}

template <int MODE>
static void run(const char *name, uint64_t outer, uint32_t K, uint32_t inner, const std::vector<uint32_t> &map, uint32_t G,
                const float *in, float *out) {
  std::vector<uint32_t> gstart(G + 1, 0), order(K);
  for (uint32_t k = 0; k < K; ++k) gstart[map[k] + 1]++;
  for (uint32_t g = 0; g < G; ++g) gstart[g + 1] += gstart[g];
  std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
  for (uint32_t k = 0; k < K; ++k) order[cur[map[k]]++] = k;
  uint32_t *dev_tab;
  std::vector<uint32_t> tab(gstart);
  tab.insert(tab.end(), order.begin(), order.end());
  CK(hipMalloc(&dev_tab, tab.size() * 4));
  CK(hipMemcpy(dev_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  DrillUpAxis a{};
  a.outer = outer;
  a.K = K;
  a.inner = inner;
  a.G = G;
  a.n_vec = inner;
  a.total = outer * G * inner;
  a.order = MODE == 2 || MODE == 1 ? nullptr : dev_tab + G + 1;
  a.gstart = dev_tab;
  a.aligned16 = 1;
  a.xcd_order = 1;
  DrillUpTile tl{};
  const uint64_t row_elems = (uint64_t)K * inner, budget = kTileBytes / 4;
  uint64_t R = budget / row_elems;
  if (R >= 4) R &= ~3ull;
  while (R > 0 && (R * row_elems) % 4 != 0) --R;
  if (!R) {
    printf("%s: row does not fit\n", name);
    return;
  }
  tl.rows_per_tile = (uint32_t)R;
  tl.row_elems = (uint32_t)row_elems;
  tl.out_row = G * inner;
  tl.inner = inner;
  uint32_t *dev_perm = nullptr;
  if (MODE == 3) {
    TilePerm tp;
    tile_perm_build(gstart.data(), order.data(), K, G, inner, budget, 4, &tp);
    std::vector<uint32_t> both(tp.cell);
    both.insert(both.end(), tp.grp.begin(), tp.grp.end());
    CK(hipMalloc(&dev_perm, both.size() * 4));
    CK(hipMemcpy(dev_perm, both.data(), both.size() * 4, hipMemcpyHostToDevice));
    tl.perm_cell = dev_perm;
    tl.perm_grp = dev_perm + tp.cell.size();
    tl.pitch_cells = tp.pitch * inner;
    R = tile_rows_for(row_elems, tl.pitch_cells, budget, 4);
    tl.rows_per_tile = (uint32_t)R;
    if (!small_div_for(row_elems, budget, &tl.by_row)) exit(2);
  }
  const unsigned tiles = (unsigned)((outer + R - 1) / R);
  const size_t lds = kTileBytes + (MODE == 1 ? 0 : MODE == 3 ? 2 * G * 4 : (G + 1 + (MODE == 2 ? 0 : K)) * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it)
    hipLaunchKernelGGL((drillup_tile_kernel<float, OLAP_SUM, false, true, MODE>), tiles, kBlock, lds, 0, Batch<float>::one(in, nullptr, out, nullptr), a, tl);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((drillup_tile_kernel<float, OLAP_SUM, false, true, MODE>), tiles, kBlock, lds, 0, Batch<float>::one(in, nullptr, out, nullptr), a, tl);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const unsigned nb = std::min(tiles, 1u << 17);
  std::vector<unsigned long long> pr((size_t)nb * 8);
  CK(hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_tile_probe), pr.size() * 8));
  double p1 = 0, p2 = 0, p3 = 0;
  unsigned long long first = ~0ull, last = 0;
  for (unsigned b = 0; b < nb; ++b) {
    const unsigned long long *q = &pr[(size_t)b * 8];
    p1 += (double)(q[1] > q[0] ? q[1] - q[0] : 0);
    p2 += (double)(q[2] - q[0]);
    p3 += (double)(q[3] - q[0]);
    first = std::min(first, q[0]);
    last = std::max(last, q[3]);
  }
  const double tick = 0.01;  // us per 100 MHz tick
  const double span = (double)(last - first) * tick;
  (void)p1;
  printf("%-34s mode %d  %7.1f us  (%u tiles, R = %u, pitch %u, lds %zu)  start ->staged %5.2f us  ->stored %5.2f us   resident/CU %.1f\n",
         name, MODE, ms * 1e3, tiles, (unsigned)R, MODE == 3 ? tl.pitch_cells : tl.row_elems, lds, p2 / nb * tick, p3 / nb * tick, p3 * tick / span / 256.0);
  if (dev_perm) CK(hipFree(dev_perm));
  CK(hipFree(dev_tab));
}

// latency of the reduction's dependent chain, one wavefront alone on a CU: N float64 adds each waiting for the last
// (a), the same with the float32 -> float64 conversion of an independent operand in between (b), and the product's
// loop shape — eight LDS reads requested a batch ahead, conversions, dependent adds (c)
template <int KIND>
__global__ __launch_bounds__(64) void chain_kernel(const float *in, double *out, unsigned long long *ticks, int n) {
  __shared__ float cells[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) cells[i] = in[i];
  __syncthreads();
  double acc = 0.0;
  const float y = in[threadIdx.x];
  const double yd = (double)y;
  const unsigned long long t0 = wall_clock64();
  if (KIND == 0) {
#pragma unroll 8
    for (int i = 0; i < n; ++i) acc += yd;
  } else if (KIND == 1) {
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      float f = y;
      asm volatile("" : "+v"(f));
      acc += (double)f;
    }
  } else if (KIND == 6) {  // three stages: LDS reads two batches ahead, conversions one batch ahead, dependent adds
    const uint32_t base = threadIdx.x * 61u;
    float x[8];
    double d[8], e[8];
    for (int u = 0; u < 8; ++u) d[u] = (double)cells[(base + u) & 4095];
    for (int u = 0; u < 8; ++u) x[u] = cells[(base + 8 + u) & 4095];
    for (int i = 16; i < n; i += 8) {
      float z[8];
      for (int u = 0; u < 8; ++u) z[u] = cells[(base + i + u) & 4095];
      for (int u = 0; u < 8; ++u) e[u] = (double)x[u];
      for (int u = 0; u < 8; ++u) acc += d[u];
      for (int u = 0; u < 8; ++u) {
        d[u] = e[u];
        x[u] = z[u];
      }
    }
  } else if (KIND == 3) {  // conversions alone, independent of each other
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i += 8)
      for (int u = 0; u < 8; ++u) {
        float f = y;
        asm volatile("" : "+v"(f));
        a[u] = (double)f;
        asm volatile("" : "+v"(a[u]));
      }
    for (int u = 0; u < 8; ++u) acc += a[u];
  } else if (KIND == 4) {  // integer restatement of the conversion (normal numbers and zero) + dependent add
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      float f = y;
      asm volatile("" : "+v"(f));
      const uint32_t u = __float_as_uint(f), ab = u & 0x7FFFFFFFu;
      uint32_t hi = (u & 0x80000000u) | ((ab >> 3) + 0x38000000u);
      hi = ab == 0 ? (u & 0x80000000u) : hi;
      acc += __hiloint2double((int)hi, (int)(u << 29));
    }
  } else if (KIND == 5) {  // the loop shape of (c) with the integer conversion
    const uint32_t base = threadIdx.x * 61u;
    float x[8], z[8];
    for (int u = 0; u < 8; ++u) x[u] = cells[(base + u) & 4095];
    for (int i = 8; i < n; i += 8) {
      for (int u = 0; u < 8; ++u) z[u] = cells[(base + i + u) & 4095];
      for (int u = 0; u < 8; ++u) {
        const uint32_t w = __float_as_uint(x[u]), ab = w & 0x7FFFFFFFu;
        uint32_t hi = (w & 0x80000000u) | ((ab >> 3) + 0x38000000u);
        hi = ab == 0 ? (w & 0x80000000u) : hi;
        acc += __hiloint2double((int)hi, (int)(w << 29));
      }
      for (int u = 0; u < 8; ++u) x[u] = z[u];
    }
  } else {
    const uint32_t base = threadIdx.x * 61u;
    float x[8], z[8];
    for (int u = 0; u < 8; ++u) x[u] = cells[(base + u) & 4095];
    for (int i = 8; i < n; i += 8) {
      for (int u = 0; u < 8; ++u) z[u] = cells[(base + i + u) & 4095];
      for (int u = 0; u < 8; ++u) acc += (double)x[u];
      for (int u = 0; u < 8; ++u) x[u] = z[u];
    }
  }
  const unsigned long long t1 = wall_clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND>
static void chain(const char *name, const float *in) {
  double *out;
  unsigned long long *ticks, h = 0;
  CK(hipMalloc(&out, 64 * 8));
  CK(hipMalloc(&ticks, 8));
  const int n = 80000;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(chain_kernel<KIND>, 1, 64, 0, 0, in, out, ticks, n);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
  printf("%-60s %6.2f ns per member\n", name, (double)h * 10.0 / n);
  CK(hipFree(out));
  CK(hipFree(ticks));
}

int main(int argc, char **) {
  const uint64_t n = 100000000ull;
  float *in, *out;
  CK(hipMalloc(&in, n * 4));
  CK(hipMalloc(&out, n * 4));
  hipLaunchKernelGGL(fill_kernel, 4096, 256, 0, 0, in, n);
  CK(hipDeviceSynchronize());
  if (argc == 1) {
    std::vector<uint32_t> m(1000);
    for (uint32_t k = 0; k < 1000; ++k) m[k] = k % 10;
    run<3>("[1e5,1000] -> 10 interleaved", 100000, 1000, 1, m, 10, in, out);
    run<0>("[1e5,1000] -> 10 interleaved", 100000, 1000, 1, m, 10, in, out);
    for (uint32_t k = 0; k < 1000; ++k) m[k] = k / 100;
    run<2>("[1e5,1000] -> 10 contiguous", 100000, 1000, 1, m, 10, in, out);
    for (uint32_t k = 0; k < 1000; ++k) m[k] = k % 100;
    run<3>("[1e5,1000] -> 100 interleaved", 100000, 1000, 1, m, 100, in, out);
    run<0>("[1e5,1000] -> 100 interleaved", 100000, 1000, 1, m, 100, in, out);
  }
  if (argc == 1) {
    std::vector<uint32_t> m(100);
    for (uint32_t k = 0; k < 100; ++k) m[k] = k % 10;
    run<3>("[1e5,100,10] -> 10 interleaved", 100000, 100, 10, m, 10, in, out);
    run<0>("[1e5,100,10] -> 10 interleaved", 100000, 100, 10, m, 10, in, out);
  }
  if (argc == 1) {
    std::vector<uint32_t> m(3652);
    for (uint32_t k = 0; k < 3652; ++k) m[k] = (uint32_t)(k / 30.4375);
    run<2>("[27400,3652] -> month", 27400, 3652, 1, m, 120, in, out);
  }
  if (argc == 1) {
    std::vector<uint32_t> m(10, 0);
    run<1>("[1e7,10] -> all (inner 1)", 10000000, 10, 1, m, 1, in, out);
  }
  chain<0>("dependent float64 adds", in);
  chain<1>("conversion + dependent float64 add", in);
  chain<2>("LDS reads a batch ahead + conversion + dependent add", in);
  chain<6>("LDS reads two batches ahead, conversions one ahead, dependent add", in);
  chain<3>("conversions alone (independent)", in);
  chain<4>("integer conversion + dependent float64 add", in);
  chain<5>("LDS reads a batch ahead + integer conversion + dependent add", in);
  return 0;
}
