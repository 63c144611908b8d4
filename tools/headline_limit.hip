// tools/headline_limit.hip — developer tool (not part of the product): where the headline roll-up's time goes.
// [10]^8 float32, dimension 0 -> all: ten row streams of 40 MB read, one 40 MB stream written.  Same process,
// alternating rounds (guide §5.4 rule 24), medians:
//   linear read 400 MB / linear write 40 MB            the box's one-way ceilings
//   product kernel                                     drillup_rows_kernel<float, sum, 16-byte lanes, one row in flight>
//   stripped                                           the same loop with nothing else in it (no tables, no Batch)
//   stripped, nothing written                          the ten read streams alone: what the READ PATTERN costs
//   stripped, written to 1 MB                          ... plus the store instructions, without their HBM traffic
//   linear 10:1                                        every workgroup reads 40 KB contiguous and writes 4 KB: the same
//                                                      read/write mix without the ten-stream pattern
//   stripped, plain stores / plain loads               cache-policy variants of the stripped loop
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I olap-in-memory_amd/csrc tools/headline_limit.hip -o tools/headline_limit.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

template <int UNR>
__global__ __launch_bounds__(256) void read_linear_kernel(const float *__restrict__ in, uint64_t n4, float *sink) {
  float acc = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
    Vec<float, 4> v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = load_stream<float, 4>(in + (i + u * stride) * 4);
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].v[0] + v[u].v[1] + v[u].v[2] + v[u].v[3];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write_linear_kernel(float *__restrict__ out, uint64_t n4) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  Vec<float, 4> o;
  o.v[0] = o.v[1] = o.v[2] = o.v[3] = (float)threadIdx.x;
  store_stream<float, 4>(out + i * 4, o);
}

// MODE 0: full (nt loads, nt stores)  1: nothing written  2: nt stores into a 1 MB window  3: nt loads, plain stores
//      4: plain loads, nt stores      5: plain stores into a 1 MB window (the L2 absorbs them: store instructions without
//      their HBM traffic)
template <int MODE>
__global__ __launch_bounds__(256) void stripped_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K, float *sink) {
  const uint64_t inner4 = inner / 4;
  const uint64_t i = (uint64_t)xcd_contiguous(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (i >= inner4) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; ++k) {
    const float *p = in + (uint64_t)k * inner + i * 4;
    const Vec<float, 4> v = MODE == 4 ? load_vec<float, 4>(p) : load_stream<float, 4>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += (double)v.v[e];
  }
  Vec<float, 4> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
  if constexpr (MODE == 1) {
    if (o.v[0] + o.v[1] + o.v[2] + o.v[3] == 12345.678f) sink[0] = o.v[0];
  } else if constexpr (MODE == 2) {
    store_stream<float, 4>(out + (i & 0xFFFFull) * 4, o);
  } else if constexpr (MODE == 5) {
    store_vec<float, 4>(out + (i & 0xFFFFull) * 4, o);
  } else if constexpr (MODE == 3) {
    store_vec<float, 4>(out + i * 4, o);
  } else {
    store_stream<float, 4>(out + i * 4, o);
  }
}

// B column blocks per workgroup, one after the other: the store of a block is in flight while the next block's rows
// are read (a workgroup's last store is what it waits for before it ends)
template <int B>
__global__ __launch_bounds__(256) void stripped_long_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  const uint64_t inner4 = inner / 4;
  const uint64_t first = (uint64_t)xcd_contiguous(blockIdx.x, gridDim.x) * B;
  for (int bb = 0; bb < B; ++bb) {
    const uint64_t i = (first + bb) * 256 + threadIdx.x;
    if (i >= inner4) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < K; ++k) {
      const Vec<float, 4> v = load_stream<float, 4>(in + (uint64_t)k * inner + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (double)v.v[e];
    }
    Vec<float, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
    store_stream<float, 4>(out + i * 4, o);
  }
}

// persistent form: a fixed grid, every workgroup walks column blocks grid-stride; block b+1's rows are requested while block
// b's store is still on its way
template <bool XCD>
__global__ __launch_bounds__(256) void stripped_persistent_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  const uint64_t inner4 = inner / 4;
  const uint64_t blocks = (inner4 + 255) / 256;
  // XCD: a workgroup's blocks stay inside its XCD's contiguous range of the output
  const uint32_t x = blockIdx.x % 8, per = gridDim.x / 8, w = blockIdx.x / 8;
  const uint64_t range = (blocks + 7) / 8;
  const uint64_t b0 = XCD ? x * range + w : blockIdx.x;
  const uint64_t bend = XCD ? (x * range + range < blocks ? x * range + range : blocks) : blocks;
  const uint64_t step = XCD ? per : gridDim.x;
  for (uint64_t b = b0; b < bend; b += step) {
    const uint64_t i = b * 256 + threadIdx.x;
    if (i >= inner4) continue;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < K; ++k) {
      const Vec<float, 4> v = load_stream<float, 4>(in + (uint64_t)k * inner + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (double)v.v[e];
    }
    Vec<float, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
    store_stream<float, 4>(out + i * 4, o);
  }
}

// the workgroup's 4 KB of results leave through LDS from ONE wavefront (4 stores of 16 bytes per lane): three of four
// wavefronts end without a store of their own in flight
__global__ __launch_bounds__(256) void stripped_onewave_store_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  __shared__ __attribute__((aligned(16))) float res[1024];
  const uint64_t inner4 = inner / 4;
  const uint64_t b = xcd_contiguous(blockIdx.x, gridDim.x);
  const uint64_t i = b * 256 + threadIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < inner4) {
    for (int k = 0; k < K; ++k) {
      const Vec<float, 4> v = load_stream<float, 4>(in + (uint64_t)k * inner + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (double)v.v[e];
    }
  }
  Vec<float, 4> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
  *reinterpret_cast<Vec<float, 4> *>(res + threadIdx.x * 4) = o;
  __syncthreads();
  if (threadIdx.x < 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t j = b * 256 + u * 64 + threadIdx.x;
      if (j < inner4) store_stream<float, 4>(out + j * 4, *reinterpret_cast<const Vec<float, 4> *>(res + (u * 64 + threadIdx.x) * 4));
    }
  }
}

// what bench.py calls the read ceiling: a workgroup reads two 4 KB pieces and ends
__global__ __launch_bounds__(256) void read_short_kernel(const float *__restrict__ src, uint64_t n_vec, float *scratch) {
  const uint64_t base = (uint64_t)blockIdx.x * 512 + threadIdx.x;
  Vec<float, 4> a{}, b{};
  if (base < n_vec) a = load_stream<float, 4>(src + base * 4);
  if (base + 256 < n_vec) b = load_stream<float, 4>(src + (base + 256) * 4);
  const float acc = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
  if (acc == 123456.789f) scratch[0] = acc;
}

// the same 10 : 1 mix of bytes without the pattern: a workgroup reads K consecutive 4 KB pieces and writes one
__global__ __launch_bounds__(256) void linear_mix_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  const uint64_t inner4 = inner / 4;
  const uint64_t b = xcd_contiguous(blockIdx.x, gridDim.x);
  const uint64_t i = b * 256 + threadIdx.x;
  if (i >= inner4) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; ++k) {
    const uint64_t at = (b * K + k) * 256 + threadIdx.x;  // 16-byte slots
    if (at < inner4 * (uint64_t)K) {
      const Vec<float, 4> v = load_stream<float, 4>(in + at * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (double)v.v[e];
    }
  }
  Vec<float, 4> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
  store_stream<float, 4>(out + i * 4, o);
}

struct Variant {
  std::string name;
  std::function<void()> fn;
  double bytes;
  std::vector<float> ms;
};

int main() {
  const uint64_t N = 100000000ull, K = 10, inner = N / K;
  float *in, *out, *sink;
  CK(hipMalloc(&in, N * 4));
  CK(hipMalloc(&out, inner * 4));
  CK(hipMalloc(&sink, 4));
  {
    std::vector<float> h(N);
    uint32_t s = 12345;
    for (uint64_t i = 0; i < N; ++i) {
      s = s * 1664525u + 1013904223u;
      h[i] = 0.5f + (s >> 8) * (1.0f / 16777216.0f);
    }
    CK(hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice));
  }
  uint32_t h_gstart[2] = {0, (uint32_t)K};
  uint32_t *gstart;
  CK(hipMalloc(&gstart, 8));
  CK(hipMemcpy(gstart, h_gstart, 8, hipMemcpyHostToDevice));
  DrillUpAxis a{};
  a.outer = 1;
  a.K = K;
  a.inner = inner;
  a.G = 1;
  a.n_vec = inner / 4;
  a.total = a.n_vec;
  a.blocks_per_row = (a.n_vec + 255) / 256;
  a.gstart = gstart;
  a.aligned16 = 1;
  a.xcd_order = 1;
  a.lanes = 256;
  const unsigned grid = (unsigned)a.blocks_per_row;
  const double rd = N * 4.0, wr = inner * 4.0;
  std::vector<Variant> vs;
  vs.push_back({"linear read 400 MB (8 in flight)", [&] { hipLaunchKernelGGL(read_linear_kernel<8>, 2048, 256, 0, 0, in, N / 4, sink); }, rd, {}});
  vs.push_back({"linear read 400 MB (short workgroups)", [&] { hipLaunchKernelGGL(read_short_kernel, (unsigned)((N / 4 + 511) / 512), 256, 0, 0, in, N / 4, sink); }, rd, {}});
  vs.push_back({"linear write 40 MB", [&] { hipLaunchKernelGGL(write_linear_kernel, grid, 256, 0, 0, out, inner / 4); }, wr, {}});
  vs.push_back({"product kernel", [&] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_SUM, false, 4, 1, true, true, true>), grid, 256, 0, 0, Batch<float>::one(in, nullptr, out, nullptr), a); }, rd + wr, {}});
  vs.push_back({"stripped", [&] { hipLaunchKernelGGL(stripped_kernel<0>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd + wr, {}});
  vs.push_back({"stripped, nothing written", [&] { hipLaunchKernelGGL(stripped_kernel<1>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd, {}});
  vs.push_back({"stripped, written to 1 MB", [&] { hipLaunchKernelGGL(stripped_kernel<2>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd, {}});
  vs.push_back({"stripped, plain stores to 1 MB", [&] { hipLaunchKernelGGL(stripped_kernel<5>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd, {}});
  vs.push_back({"stripped, 2 blocks per workgroup", [&] { hipLaunchKernelGGL(stripped_long_kernel<2>, (grid + 1) / 2, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"stripped, 4 blocks per workgroup", [&] { hipLaunchKernelGGL(stripped_long_kernel<4>, (grid + 3) / 4, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"persistent 2048 workgroups, grid-stride", [&] { hipLaunchKernelGGL(stripped_persistent_kernel<false>, 2048, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"persistent 2048 workgroups, per XCD", [&] { hipLaunchKernelGGL(stripped_persistent_kernel<true>, 2048, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"persistent 4096 workgroups, per XCD", [&] { hipLaunchKernelGGL(stripped_persistent_kernel<true>, 4096, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"persistent 1024 workgroups, per XCD", [&] { hipLaunchKernelGGL(stripped_persistent_kernel<true>, 1024, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"stripped, one wavefront stores (via LDS)", [&] { hipLaunchKernelGGL(stripped_onewave_store_kernel, grid, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  vs.push_back({"stripped, plain stores", [&] { hipLaunchKernelGGL(stripped_kernel<3>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd + wr, {}});
  vs.push_back({"stripped, plain loads", [&] { hipLaunchKernelGGL(stripped_kernel<4>, grid, 256, 0, 0, in, out, inner, (int)K, sink); }, rd + wr, {}});
  vs.push_back({"linear 10:1 (40 KB read, 4 KB written)", [&] { hipLaunchKernelGGL(linear_mix_kernel, grid, 256, 0, 0, in, out, inner, (int)K); }, rd + wr, {}});
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](std::function<void()> &f, int iters) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
  };
  for (auto &v : vs) run(v.fn, 5);
  for (int r = 0; r < 9; ++r)
    for (auto &v : vs) v.ms.push_back(run(v.fn, 50));
  printf("%-44s %9s %9s %11s %9s\n", "variant", "med us", "min us", "GB/s (med)", "of 8 TB/s");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2];
    printf("%-44s %9.2f %9.2f %11.1f %9.3f\n", v.name.c_str(), med * 1e3, v.ms[0] * 1e3, v.bytes / (med * 1e-3) / 1e9, v.bytes / (med * 1e-3) / 8e12);
  }
  return 0;
}
