// tools/ceilings.hip — developer tool: sweep of plain HBM read / copy kernels on this box, to know
// what "speed of light" is for a 400 MB stream before judging the drillUp kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

template <bool NT>
__device__ __forceinline__ f4 ld(const f4 *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// tile mapping: block b owns UNR*256 consecutive float4; TILES tiles per block (persistent-ish)
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void read_tile(const f4 *__restrict__ in, uint64_t n4, float *sink) {
  float acc = 0.f;
  const uint64_t tile = (uint64_t)UNR * 256;
  for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n4; base += (uint64_t)gridDim.x * tile) {
    f4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint64_t i = base + u * 256 + threadIdx.x;
      v[u] = i < n4 ? ld<NT>(in + i) : f4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_tile(const f4 *__restrict__ in, f4 *__restrict__ out, uint64_t n4) {
  const uint64_t tile = (uint64_t)UNR * 256;
  for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n4; base += (uint64_t)gridDim.x * tile) {
    f4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) v[u] = ld<NT>(in + i);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint64_t i = base + u * 256 + threadIdx.x;
      if (i < n4) {
        if constexpr (NT) __builtin_nontemporal_store(v[u], out + i);
        else out[i] = v[u];
      }
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void write_tile(f4 *__restrict__ out, uint64_t n4) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
    if constexpr (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

struct V {
  std::string name;
  std::function<void()> fn;
  double bytes;
  std::vector<float> ms;
};

int main() {
  const uint64_t N = 100000000ull, n4 = N / 4;
  float *in, *out, *sink;
  CK(hipMalloc(&in, N * 4));
  CK(hipMalloc(&out, N * 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<V> vs;
  const unsigned grids[] = {1024, 2048, 4096, 8192, 16384, 0};
#define ADD_READ(UNR, NT)                                                                                          \
  for (unsigned g : grids) {                                                                                       \
    unsigned gg = g ? g : (unsigned)((n4 + UNR * 256 - 1) / (UNR * 256));                                          \
    char nm[96];                                                                                                   \
    snprintf(nm, sizeof nm, "read  unr=%d nt=%d grid=%u%s", UNR, NT, gg, g ? "" : " (one tile per block)");       \
    vs.push_back({nm, [=] { hipLaunchKernelGGL((read_tile<UNR, NT>), gg, 256, 0, 0, (const f4 *)in, n4, sink); }, N * 4.0, {}}); \
  }
#define ADD_COPY(UNR, NT)                                                                                          \
  for (unsigned g : grids) {                                                                                       \
    unsigned gg = g ? g : (unsigned)((n4 + UNR * 256 - 1) / (UNR * 256));                                          \
    char nm[96];                                                                                                   \
    snprintf(nm, sizeof nm, "copy  unr=%d nt=%d grid=%u%s", UNR, NT, gg, g ? "" : " (one tile per block)");       \
    vs.push_back({nm, [=] { hipLaunchKernelGGL((copy_tile<UNR, NT>), gg, 256, 0, 0, (const f4 *)in, (f4 *)out, n4); }, N * 8.0, {}}); \
  }
  ADD_READ(1, false) ADD_READ(2, false) ADD_READ(4, false) ADD_READ(8, false)
  ADD_READ(2, true) ADD_READ(4, true)
  ADD_COPY(1, false) ADD_COPY(2, false) ADD_COPY(4, false) ADD_COPY(4, true)
  for (unsigned g : {2048u, 8192u, 97657u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "write nt=0 grid=%u", g);
    vs.push_back({nm, [=] { hipLaunchKernelGGL((write_tile<false>), g, 256, 0, 0, (f4 *)out, n4); }, N * 4.0, {}});
    snprintf(nm, sizeof nm, "write nt=1 grid=%u", g);
    vs.push_back({nm, [=] { hipLaunchKernelGGL((write_tile<true>), g, 256, 0, 0, (f4 *)out, n4); }, N * 4.0, {}});
  }
  auto run = [&](V &v, int iters) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) v.fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
  };
  for (auto &v : vs) run(v, 2);
  for (int r = 0; r < 5; ++r)
    for (auto &v : vs) v.ms.push_back(run(v, 10));
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2];
    printf("%-52s %8.2f us %8.1f GB/s  %.3f\n", v.name.c_str(), med * 1e3, v.bytes / (med * 1e-3) / 1e9, v.bytes / (med * 1e-3) / 1e9 / 8000.0);
  }
  return 0;
}
