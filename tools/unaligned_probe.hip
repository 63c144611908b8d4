// tools/unaligned_probe.hip — developer tool: what does a 16-byte access cost on gfx950 when its address is
// only 4-byte aligned?  Copies and reads of 10^8 float32 cells with the source / destination shifted by 0..3
// cells, next to the 4-byte-per-lane form the odd-extent paths fall back to.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/unaligned_probe tools/unaligned_probe.hip && /tmp/unaligned_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e = (x);                                                              \
    if (e != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

// a 16-byte group of cells whose address is a multiple of 4 only
typedef float Q4 __attribute__((ext_vector_type(4), aligned(4)));

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_q(const float *__restrict__ in, float *__restrict__ out, uint64_t nq) {
  const uint64_t base = (uint64_t)blockIdx.x * UNR * 256;
  Q4 v[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t i = base + u * 256 + threadIdx.x;
    if (i < nq) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const Q4 *>(in) + i) : reinterpret_cast<const Q4 *>(in)[i];
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t i = base + u * 256 + threadIdx.x;
    if (i < nq) {
      if (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<Q4 *>(out) + i);
      else reinterpret_cast<Q4 *>(out)[i] = v[u];
    }
  }
}

template <int UNR>
__global__ __launch_bounds__(256) void copy_1(const float *__restrict__ in, float *__restrict__ out, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * UNR * 256;
  float v[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t i = base + u * 256 + threadIdx.x;
    if (i < n) v[u] = __builtin_nontemporal_load(in + i);
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t i = base + u * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(v[u], out + i);
  }
}

template <int UNR>
__global__ __launch_bounds__(256) void read_q(const float *__restrict__ in, uint64_t nq, float *sink) {
  const uint64_t base = (uint64_t)blockIdx.x * UNR * 256;
  float acc = 0.f;
  Q4 v[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const uint64_t i = base + u * 256 + threadIdx.x;
    if (i < nq) v[u] = __builtin_nontemporal_load(reinterpret_cast<const Q4 *>(in) + i);
    else v[u] = Q4{0, 0, 0, 0};
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  if (acc == 12345.678f) sink[0] = acc;
}

template <typename F>
static float time_us(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / 20;
}

int main() {
  const uint64_t n = 100000000;
  float *in, *out, *sink;
  CK(hipMalloc(&in, (n + 64) * 4));
  CK(hipMalloc(&out, (n + 64) * 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, (n + 64) * 4));
  const uint64_t nq = n / 4;
  for (int si = 0; si < 4; ++si)
    for (int so = 0; so < 4; so += (si == 0 ? 1 : 3)) {
      {
        constexpr int U = 4;
        const unsigned grid = (unsigned)((nq + U * 256 - 1) / (U * 256));
        float us = time_us([&] { copy_q<U, true><<<grid, 256>>>(in + si, out + so, nq); });
        printf("copy 16B/lane U=4 nt   src+%d dst+%d  %7.1f us  %6.0f GB/s\n", si, so, us, 2.0 * n * 4 / us * 1e-3);
      }
      {
        constexpr int U = 8;
        const unsigned grid = (unsigned)((nq + U * 256 - 1) / (U * 256));
        float us = time_us([&] { copy_q<U, true><<<grid, 256>>>(in + si, out + so, nq); });
        printf("copy 16B/lane U=8 nt   src+%d dst+%d  %7.1f us  %6.0f GB/s\n", si, so, us, 2.0 * n * 4 / us * 1e-3);
      }
      {
        constexpr int U = 4;
        const unsigned grid = (unsigned)((nq + U * 256 - 1) / (U * 256));
        float us = time_us([&] { copy_q<U, false><<<grid, 256>>>(in + si, out + so, nq); });
        printf("copy 16B/lane U=4      src+%d dst+%d  %7.1f us  %6.0f GB/s\n", si, so, us, 2.0 * n * 4 / us * 1e-3);
      }
    }
  for (int si = 0; si < 4; ++si) {
    constexpr int U = 8;
    const unsigned grid = (unsigned)((nq + U * 256 - 1) / (U * 256));
    float us = time_us([&] { read_q<U><<<grid, 256>>>(in + si, nq, sink); });
    printf("read 16B/lane U=8 nt   src+%d        %7.1f us  %6.0f GB/s\n", si, us, 1.0 * n * 4 / us * 1e-3);
  }
  {
    constexpr int U = 8;
    const unsigned grid = (unsigned)((n + U * 256 - 1) / (U * 256));
    float us = time_us([&] { copy_1<U><<<grid, 256>>>(in, out, n); });
    printf("copy 4B/lane  U=8 nt                  %7.1f us  %6.0f GB/s\n", us, 2.0 * n * 4 / us * 1e-3);
  }
  {
    constexpr int U = 16;
    const unsigned grid = (unsigned)((n + U * 256 - 1) / (U * 256));
    float us = time_us([&] { copy_1<U><<<grid, 256>>>(in, out, n); });
    printf("copy 4B/lane  U=16 nt                 %7.1f us  %6.0f GB/s\n", us, 2.0 * n * 4 / us * 1e-3);
  }
  return 0;
}
