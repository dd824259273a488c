// Probe: v_mfma_f32_4x4x1_16B_f32 -- operand layout, is one step a single fmaf, sustained rate (cycles per instruction
// with 4 / 8 / 16 independent accumulators; alone on the SIMD and with two waves), against v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k4(const float* A, const float* B, float* D, int K) {
  // 16 blocks; block b: A_b [4][K], B_b [K][4], D_b [4][4].   A: [16][4][K], B: [16][K][4], D: [16][4][4]
  const int l = threadIdx.x, b = l >> 2, i = l & 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const float a = A[(b * 4 + i) * K + k];
    const float bb = B[(b * K + k) * 4 + i];
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(b * 4 + r) * 4 + i] = acc[r];     // guess: register r = row, lane & 3 = column
}

template <int NACC>
__global__ void rate4(float* out, int iters, long long* cycles) {
  f32x4 acc[NACC];
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int NACC>
__global__ void rate16(float* out, int iters, long long* cycles) {
  f32x4 acc[NACC];
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
// the mix a real kernel would issue: per 4 MFMAs one ds_read_b128-like LDS read and some VALU
template <int NACC>
__global__ void rate4_mix(float* out, int iters, long long* cycles) {
  __shared__ float lds[64 * 36];
  for (int i = threadIdx.x; i < 64 * 36; i += blockDim.x) lds[i] = i * 0.25f;
  __syncthreads();
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  const float4* p = reinterpret_cast<const float4*>(lds + lane * 36);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const float4 xv = p[it & 7];
    const float4 wv = p[(it + 3) & 7];
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const float a = j % 4 == 0 ? wv.x : j % 4 == 1 ? wv.y : j % 4 == 2 ? wv.z : wv.w;
      const float b = j % 4 == 0 ? xv.x : j % 4 == 1 ? xv.y : j % 4 == 2 ? xv.z : xv.w;
      acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * expf((float)(rand() % 8 - 4)); }
int main() {
  const int K = 64;
  std::vector<float> A(16 * 4 * K), B(16 * K * 4), D(16 * 16);
  srand(3);
  for (auto& v : A) v = frand();
  for (auto& v : B) v = frand();
  float *dA, *dB, *dD; long long* dc;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1 << 22); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k4<<<1, 64>>>(dA, dB, dD, K);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad_fma = 0, bad_mul = 0;
  for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    float f = 0.f, m = 0.f;
    for (int k = 0; k < K; ++k) {
      f = fmaf(A[(b * 4 + i) * K + k], B[(b * K + k) * 4 + j], f);
      volatile float pr = A[(b * 4 + i) * K + k] * B[(b * K + k) * 4 + j];
      m = m + pr;
    }
    if (f != D[(b * 4 + i) * 4 + j]) ++bad_fma;
    if (m != D[(b * 4 + i) * 4 + j]) ++bad_mul;
  }
  printf("4x4x1_16B: layout (reg = row, lane&3 = column, lane>>2 = block): %d of 256 differ from the fmaf chain, %d from mul+add\n", bad_fma, bad_mul);
  const int iters = 20000;
  long long c;
#define RUN(kern, nacc, blocks, threads, per)                                                                         \
  kern<nacc><<<blocks, threads>>>(dD, iters, dc); hipDeviceSynchronize();                                             \
  kern<nacc><<<blocks, threads>>>(dD, iters, dc); hipDeviceSynchronize();                                             \
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);                                                                        \
  printf("%-10s acc %2d  blocks %4d x %3d threads: %.2f ticks per MFMA per wave\n", #kern, nacc, blocks, threads, (double)c / iters / nacc);
  RUN(rate4, 4, 1, 64, 0) RUN(rate4, 8, 1, 64, 0) RUN(rate4, 16, 1, 64, 0)
  RUN(rate4, 16, 1, 256, 0)           // one wave per SIMD
  RUN(rate4, 16, 1, 512, 0)           // two waves per SIMD
  RUN(rate4, 16, 256 * 2, 256, 0)     // whole GPU, two waves per SIMD
  RUN(rate4, 16, 1, 768, 0) RUN(rate4, 16, 1, 1024, 0) RUN(rate4, 8, 1, 1024, 0) RUN(rate4, 4, 1, 1024, 0)
  RUN(rate4_mix, 16, 1, 768, 0) RUN(rate4_mix, 16, 1, 1024, 0)
  RUN(rate16, 2, 1, 64, 0) RUN(rate16, 4, 1, 64, 0) RUN(rate16, 4, 1, 512, 0) RUN(rate16, 4, 256 * 2, 256, 0)
  RUN(rate4_mix, 16, 1, 64, 0) RUN(rate4_mix, 16, 1, 256, 0) RUN(rate4_mix, 16, 1, 512, 0) RUN(rate4_mix, 16, 256 * 2, 256, 0)
  // wall-clock rate over the whole GPU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define WALL(kern, nacc, blocks, threads, flop_per_mfma)                                                              \
  kern<nacc><<<blocks, threads>>>(dD, iters, dc); hipDeviceSynchronize();                                             \
  hipEventRecord(e0); kern<nacc><<<blocks, threads>>>(dD, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);    \
  { float ms; hipEventElapsedTime(&ms, e0, e1);                                                                       \
    printf("%-10s acc %2d blocks %4d x %4d: %.3f ms, %.1f TFLOP/s, err %d\n", #kern, nacc, blocks, threads, ms,        \
           (double)(blocks) * ((threads) / 64) * iters * nacc * flop_per_mfma / (ms * 1e-3) / 1e12, (int)hipGetLastError()); }
  WALL(rate4, 16, 256, 256, 512.) WALL(rate4, 16, 256, 512, 512.) WALL(rate4, 16, 256, 1024, 512.) WALL(rate4, 16, 512, 1024, 512.)
  WALL(rate16, 4, 256, 256, 2048.) WALL(rate16, 4, 256, 512, 2048.)
  WALL(rate4_mix, 16, 256, 512, 512.) WALL(rate4_mix, 16, 256, 1024, 512.) WALL(rate4_mix, 16, 512, 768, 512.)
  return 0;
}
