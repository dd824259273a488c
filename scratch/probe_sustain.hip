// Probe: does a pure-MFMA kernel (registers only, no memory traffic) hold the 2.4 GHz peak when it runs for 100 ms and more,
// or does it fall to the ~2.09 GHz the conv GEMM sees? f32 32x32x2 and 16x16x4, one and three waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate32(float* out, int iters) {
  f32x16 acc[NACC];
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void rate16(float* out, int iters) {
  f32x4 acc[NACC];
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same with operands that change every instruction: 16 pseudo-random values per lane held in registers (no memory traffic)
template <int NACC>
__global__ void rate32_toggle(float* out, int iters) {
  f32x16 acc[NACC];
  float av[16], bv[16];
  unsigned int s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 16; ++i) {
    s = s * 1664525u + 1013904223u; av[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
    s = s * 1664525u + 1013904223u; bv[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  }
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(4 * u + j) & 15], bv[(4 * u + 3 * j + 1) & 15], acc[j], 0, 0, 0);
  }
  float t = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) t += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
  float* sink; hipMalloc(&sink, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define WALL(kern, nacc, blocks, threads, iters, flop)                                                               \
  { hipEventRecord(e0); hipLaunchKernelGGL((kern<nacc>), dim3(blocks), dim3(threads), 0, 0, sink, iters); hipEventRecord(e1);   \
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                                             \
    printf("%-7s acc %d  %4d x %4d threads  %9d iters: %8.2f ms  %.1f TFLOP/s\n", #kern, nacc, blocks, threads, iters, ms,   \
           (double)(blocks) * ((threads) / 64) * (double)(iters) * nacc * flop / (ms * 1e-3) / 1e12); }
  for (int rep = 0; rep < 2; ++rep) {
    WALL(rate32, 4, 256, 256, 4000, 4096.)          // ~1 ms
    WALL(rate32, 4, 256, 256, 400000, 4096.)        // ~100 ms
    WALL(rate32, 4, 256, 768, 133000, 4096.)        // three waves per SIMD, ~100 ms
    WALL(rate16, 4, 256, 256, 800000, 2048.)        // ~100 ms
    WALL(rate32, 4, 256, 256, 2000000, 4096.)       // ~500 ms
    WALL(rate32_toggle, 4, 256, 256, 400000, 4096.)
    WALL(rate32_toggle, 4, 256, 768, 133000, 4096.)
    WALL(rate32_toggle, 4, 256, 256, 2000000, 4096.)
  }
  return 0;
}
