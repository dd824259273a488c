// Probe: is v_mfma_f32_32x32x2_f32 / 16x16x4 a k-ascending fmaf chain, bitwise? Are sqrtf and '/' correctly rounded?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float* A, const float* B, float* D, int K) {
  // A: [32][K], B: [K][32], D: [32][32]
  int l = threadIdx.x;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) {
    float a = A[(l & 31) * K + k + (l >> 5)];
    float b = B[(k + (l >> 5)) * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = acc[r];
  }
}
__global__ void k16(const float* A, const float* B, float* D, int K) {
  // A: [16][K], B: [K][16], D: [16][16]
  int l = threadIdx.x;
  f32x4 acc = {0};
  for (int k = 0; k < K; k += 4) {
    float a = A[(l & 15) * K + k + (l >> 4)];
    float b = B[(k + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    D[row * 16 + (l & 15)] = acc[r];
  }
}
__global__ void kds(const float* x, const float* y, float* q, float* s, float* r, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { q[i] = x[i] / y[i]; s[i] = sqrtf(fabsf(y[i])); r[i] = rintf(x[i]); }
}
static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * expf((float)(rand() % 8 - 4)); }
int main() {
  system("nproc; free -g | head -2; lscpu | grep 'Model name'");
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("dev %s CUs %d clock %d MHz mem %.1f GB l2 %d\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.totalGlobalMem / 1e9, p.l2CacheSize);
  const int K = 256;
  std::vector<float> A(32 * K), B(K * 32), D(32 * 32), R(32 * 32);
  srand(1);
  for (auto& v : A) v = frand();
  for (auto& v : B) v = frand();
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k32<<<1, 64>>>(dA, dB, dD, K);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
    if (memcmp(&acc, &D[i * 32 + j], 4)) bad++;
  }
  printf("mfma32x32x2 vs k-ascending fmaf chain: %d / 1024 mismatches\n", bad);
  // 16x16x4: reuse A as [16][K], B as [K][16]
  std::vector<float> B16(K * 16);
  for (auto& v : B16) v = frand();
  hipMemcpy(dB, B16.data(), B16.size() * 4, hipMemcpyHostToDevice);
  k16<<<1, 64>>>(dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B16[k * 16 + j], acc);
    if (memcmp(&acc, &D[i * 16 + j], 4)) bad++;
  }
  printf("mfma16x16x4 vs k-ascending fmaf chain: %d / 256 mismatches\n", bad);
  const int n = 1 << 20;
  std::vector<float> x(n), y(n), q(n), s(n), r(n);
  for (int i = 0; i < n; ++i) { x[i] = frand() * 100.f; y[i] = frand() + (frand() == 0 ? 1 : 0); if (y[i] == 0) y[i] = 1; }
  for (int i = 0; i < 64; ++i) x[i] = (float)(i - 32) + 0.5f;
  float *dx, *dy, *dq, *ds, *dr;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dq, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
  kds<<<n / 256, 256>>>(dx, dy, dq, ds, dr, n);
  hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
  int bq = 0, bs = 0, br = 0;
  for (int i = 0; i < n; ++i) {
    float cq = x[i] / y[i], cs = sqrtf(fabsf(y[i])), cr = rintf(x[i]);
    bq += memcmp(&cq, &q[i], 4) != 0; bs += memcmp(&cs, &s[i], 4) != 0; br += memcmp(&cr, &r[i], 4) != 0;
  }
  printf("div mismatches %d, sqrt mismatches %d, rint mismatches %d of %d\n", bq, bs, br, n);
  return 0;
}
