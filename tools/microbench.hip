// tools/microbench.hip — developer tool (not part of the product): HBM ceilings on this box and
// A/B of drillUp kernel variants, interleaved rounds in one process (guide §5.4 rule 24).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I olap-in-memory_amd/csrc tools/microbench.hip -o build/microbench
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

// read ceiling: grid-stride float4 loads, UNR independent loads in flight per lane
template <int UNR>
__global__ __launch_bounds__(256) void read_sum_kernel(const float4 *__restrict__ in, uint64_t n4, float *sink) {
  float acc = 0.f;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) {
    float4 v = in[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;  // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, uint64_t n4) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}

// the cheapest possible column sum: K rows of float4, f32 adds, value-only output
template <int K>
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, uint64_t inner4) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= inner4) return;
  float4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = in[i + (uint64_t)k * inner4];
  float4 a = v[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    a.x += v[k].x;
    a.y += v[k].y;
    a.z += v[k].z;
    a.w += v[k].w;
  }
  out[i] = a;
}

// K rows, one row in flight, C column slots (256 slots = 4 KiB apart) per lane: a workgroup sweeps
// C*4 KiB of contiguous memory per row, float64 accumulation, non-temporal loads
template <int C, bool XCD>
__global__ __launch_bounds__(256) void colrows_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  const uint32_t bid = XCD ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint64_t inner4 = inner / 4;
  const uint64_t base = (uint64_t)bid * 256 * C + threadIdx.x;
  double acc[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[c][e] = 0.0;
  for (int k = 0; k < K; ++k) {
    Vec<float, 4> v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const uint64_t i = base + (uint64_t)c * 256;
      if (i < inner4) v[c] = load_stream<float, 4>(in + (uint64_t)k * inner + i * 4);
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e] += (double)v[c].v[e];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const uint64_t i = base + (uint64_t)c * 256;
    if (i < inner4) {
      Vec<float, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[c][e];
      store_stream<float, 4>(out + i * 4, o);
    }
  }
}

// cache-policy variants of the same loop: POLICY 0 plain, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc1, 5 sc0 sc1
template <int POLICY>
__device__ __forceinline__ float4 load_policy(const float4 *p) {
  float4 v;
  if constexpr (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POLICY>
__global__ __launch_bounds__(256) void colpolicy_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t inner, int K) {
  const uint64_t inner4 = inner / 4;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= inner4) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < K; ++k) {
    float4 v = load_policy<POLICY>(reinterpret_cast<const float4 *>(in + (uint64_t)k * inner) + i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc[0] += (double)v.x;
    acc[1] += (double)v.y;
    acc[2] += (double)v.z;
    acc[3] += (double)v.w;
  }
  Vec<float, 4> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.v[e] = (float)acc[e];
  store_stream<float, 4>(out + i * 4, o);
}

struct Timer {
  hipEvent_t a, b;
  Timer() {
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
  }
  template <typename F>
  float run(F f, int iters) {
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
  }
};

struct Variant {
  std::string name;
  std::function<void()> fn;
  double bytes;
  std::vector<float> ms;
};

int main(int argc, char **argv) {
  const uint64_t N = 100000000ull, K = 10, inner = N / K;
  float *in, *out, *sink;
  int32_t *st_out;
  CK(hipMalloc(&in, N * 4));
  CK(hipMalloc(&out, N * 4));  // big enough for the copy test
  CK(hipMalloc(&st_out, inner * 4));
  int32_t *st_in;
  CK(hipMalloc(&st_in, N * 4));
  {
    std::vector<int32_t> hs(N, 2);
    CK(hipMemcpy(st_in, hs.data(), N * 4, hipMemcpyHostToDevice));
  }
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
  a.order = nullptr;
  a.gstart = gstart;
  a.def_nan = 0;
  a.aligned16 = 1;
  const unsigned rows_grid = (unsigned)a.blocks_per_row;

  std::vector<Variant> vs;
  const double alg = N * 4.0 + inner * 8.0;
  vs.push_back({"read_sum<4> 2048 blocks", [&] { hipLaunchKernelGGL(read_sum_kernel<4>, 2048, 256, 0, 0, (const float4 *)in, N / 4, sink); }, N * 4.0, {}});
  vs.push_back({"read_sum<8> 2048 blocks", [&] { hipLaunchKernelGGL(read_sum_kernel<8>, 2048, 256, 0, 0, (const float4 *)in, N / 4, sink); }, N * 4.0, {}});
  vs.push_back({"read_sum<8> 4096 blocks", [&] { hipLaunchKernelGGL(read_sum_kernel<8>, 4096, 256, 0, 0, (const float4 *)in, N / 4, sink); }, N * 4.0, {}});
  vs.push_back({"read_sum<2> 8192 blocks", [&] { hipLaunchKernelGGL(read_sum_kernel<2>, 8192, 256, 0, 0, (const float4 *)in, N / 4, sink); }, N * 4.0, {}});
  vs.push_back({"copy 2048 blocks (r+w)", [&] { hipLaunchKernelGGL(copy_kernel, 2048, 256, 0, 0, (const float4 *)in, (float4 *)out, N / 4); }, N * 8.0, {}});
  vs.push_back({"colsum_f32<10> value-only", [&] { hipLaunchKernelGGL(colsum_f32_kernel<10>, rows_grid, 256, 0, 0, (const float4 *)in, (float4 *)out, inner / 4); }, N * 4.0 + inner * 4.0, {}});
#define ROWS(U, FAST, NT, ST, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_SUM, false, 4, U, true, FAST, NT>), rows_grid, 256, 0, 0, Batch<float>::one(in, nullptr, out, ST), a); }, (ST) ? alg : N * 4.0 + inner * 4.0, {}})
  ROWS(8, true, true, (int32_t *)nullptr, "rows U=8 nt values only");
  ROWS(4, true, true, (int32_t *)nullptr, "rows U=4 nt values only (product)");
  ROWS(2, true, true, (int32_t *)nullptr, "rows U=2 nt values only");
  ROWS(1, true, true, (int32_t *)nullptr, "rows U=1 nt values only");
  ROWS(5, true, true, (int32_t *)nullptr, "rows U=5 nt values only");
  ROWS(10, true, true, (int32_t *)nullptr, "rows U=10 nt values only");
  {
    static DrillUpAxis a8 = a;
    a8.n_vec = inner / 8;
    a8.total = a8.n_vec;
    a8.blocks_per_row = (a8.n_vec + 255) / 256;
    const unsigned g8 = (unsigned)a8.blocks_per_row;
    vs.push_back({"rows VEC=8 U=4 nt values only", [&, g8] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_SUM, false, 8, 4, true, true, true>), g8, 256, 0, 0, Batch<float>::one(in, nullptr, out, nullptr), a8); }, N * 4.0 + inner * 4.0, {}});
    vs.push_back({"rows VEC=8 U=2 nt values only", [&, g8] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_SUM, false, 8, 2, true, true, true>), g8, 256, 0, 0, Batch<float>::one(in, nullptr, out, nullptr), a8); }, N * 4.0 + inner * 4.0, {}});
  }
#define COLS(C, X, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((colrows_kernel<C, X>), (unsigned)((inner / 4 + 256 * C - 1) / (256 * C)), 256, 0, 0, in, out, inner, (int)K); }, N * 4.0 + inner * 4.0, {}})
  COLS(1, false, "colrows C=1");
  COLS(2, false, "colrows C=2");
  COLS(4, false, "colrows C=4");
  COLS(8, false, "colrows C=8");
  COLS(2, true, "colrows C=2 xcd");
  COLS(4, true, "colrows C=4 xcd");
#define POL(P, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((colpolicy_kernel<P>), (unsigned)((inner / 4 + 255) / 256), 256, 0, 0, in, out, inner, (int)K); }, N * 4.0 + inner * 4.0, {}})
  POL(0, "policy plain");
  POL(1, "policy nt");
  POL(2, "policy sc1 nt");
  POL(3, "policy sc0 sc1 nt");
  POL(4, "policy sc1");
  POL(5, "policy sc0 sc1");
  ROWS(4, true, true, st_out, "rows U=4 nt + status out");
  ROWS(1, true, true, st_out, "rows U=1 nt + status out");
  ROWS(1, false, true, (int32_t *)nullptr, "rows U=1 exact");
  ROWS(2, false, true, (int32_t *)nullptr, "rows U=2 exact");
  ROWS(4, false, true, (int32_t *)nullptr, "rows U=4 exact");
#define ROWS_ST(U, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_SUM, true, 4, U, true, false, true>), rows_grid, 256, 0, 0, Batch<float>::one(in, st_in, out, st_out), a); }, 2 * alg, {}})
  ROWS_ST(1, "rows U=1 mask in+out");
  ROWS_ST(2, "rows U=2 mask in+out");
  ROWS_ST(4, "rows U=4 mask in+out");
#define ROWS_HI(U, NAME) vs.push_back({NAME, [&] { hipLaunchKernelGGL((drillup_rows_kernel<float, OLAP_HIGHEST, false, 4, U, true, false, true>), rows_grid, 256, 0, 0, Batch<float>::one(in, nullptr, out, nullptr), a); }, N * 4.0 + inner * 4.0, {}})
  ROWS_HI(1, "rows U=1 highest");
  ROWS_HI(4, "rows U=4 highest");

  Timer t;
  for (auto &v : vs) t.run(v.fn, 3);  // warm-up
  const int rounds = 7, iters = 20;
  for (int r = 0; r < rounds; ++r)
    for (auto &v : vs) v.ms.push_back(t.run(v.fn, iters));
  printf("%-36s %10s %10s %12s %12s\n", "variant", "med us", "min us", "GB/s(med)", "frac of 8TB/s");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-36s %10.2f %10.2f %12.1f %12.3f\n", v.name.c_str(), med * 1e3, mn * 1e3, v.bytes / (med * 1e-3) / 1e9, v.bytes / (med * 1e-3) / 1e9 / 8000.0);
  }
  return 0;
}
