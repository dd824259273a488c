// Probe: how fast does one wave's VALU stream issue while another wave on the same SIMD streams MFMAs (and vice versa)?
// Block of 8 waves on one CU: waves 0-3 run role A, waves 4-7 role B (wave w and w + 4 share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// role: 0 idle, 1 = MFMA 32x32x2 (4 independent accumulators), 2 = MFMA 16x16x4 (2 acc), 3 = VALU v_fma (8 independent chains),
// 4 = MFMA 32x32x2 with ONE accumulator (dependent chain)
__global__ void k(int roleA, int roleB, int iters, long long* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
  f32x16 acc[4]; f32x4 acq[2]; float v[8];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int j = 0; j < 2; ++j) acq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < 8; ++j) v[j] = j;
  __syncthreads();
  const long long t0 = clock64();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acq[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acq[j & 1], 0, 0, 0);
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], b, a);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  s += acq[0][0] + acq[1][0];
  for (int j = 0; j < 8; ++j) s += v[j];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
int main() {
  long long* d; float* sink; long long h[8];
  hipMalloc(&d, 64); hipMalloc(&sink, 1 << 20);
  const int iters = 4000;
  const char* names[] = {"idle", "mfma32 x4 acc", "mfma16 x2 acc", "valu fma x8", "mfma32 dependent"};
  const int per[] = {0, 4, 8, 32, 4};
  const int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 3}, {2, 3}, {4, 3}, {3, 3}, {1, 1}, {1, 4}};
  for (auto& c : combos) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, c[0], c[1], iters, d, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, c[0], c[1], iters, d, sink); hipDeviceSynchronize();
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("A = %-17s B = %-17s : A %.1f cycles / instr", names[c[0]], names[c[1]], (double)h[0] / iters / per[c[0]]);
    if (c[1]) printf(", B %.1f cycles / instr", (double)h[4] / iters / per[c[1]]);
    printf("\n");
  }
  return 0;
}
