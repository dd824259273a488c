// r04 copy of probe_mix.hip with one more mode (bit 5 = 32): the activation tile staged by LDS-DMA (4 x buffer_load_dwordx4 ... lds per
// K-step, no VGPR round trip, no ds_write) next to the 16 weight loads and the 4 ds_read_b128 -- what VERDICT round 3 item 7(b) asks about.
// Probe: which companion activity makes the chip leave the ~2.37 GHz it holds under pure f32 MFMA load? A wave runs "K-steps"
// of 64 v_mfma_f32_32x32x2_f32 (4 accumulators, operands from a 16-entry register ring that toggles every instruction), three
// waves per SIMD like the conv GEMM, and per step optionally: LDS traffic (8 ds_write_b64 + 4 ds_read_b128), global loads
// (20 x 16 B per lane from a 64 KB table: L1 / L2 hits), integer vector work (40 instructions), FP32 vector work (40 FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // bit 0: LDS, bit 1: global loads, bit 2: integer VALU, bit 3: FP VALU
__global__ __launch_bounds__(256, 3) void mix(const float* __restrict__ table, float* out, int steps) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 1152];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wl = lds + wave * 2 * 1152;
  f32x16 acc[4];
  float av[16], bv[16];
  unsigned int s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 16; ++i) {
    s = s * 1664525u + 1013904223u; av[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
    s = s * 1664525u + 1013904223u; bv[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  }
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, 65536, 0x00020000);
  float4 g[4] = {make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4), make_float4(1, 2, 3, 4)};
  int iv = lane;
  float fv = lane * 0.5f;
  for (int st = 0; st < steps; ++st) {
    if (MODE & 32) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wl + (st & 1) * 1152 + i * 256),
                                                 16, (lane * 16 + i * 2048 + st * 8192 + 512) & 65535, 0, 0, 0);
      const float4* rd = reinterpret_cast<const float4*>(wl + ((st + 1) & 1) * 1152 + (lane & 31) * 32 + (lane >> 5) * 16);
      const float4 a0 = rd[0], a1 = rd[1], a2 = rd[2], a3 = rd[3];
      bv[0] += a0.x * 1e-30f; bv[4] += a1.y * 1e-30f; bv[8] += a2.z * 1e-30f; bv[12] += a3.w * 1e-30f;
    }
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* d = wl + (st & 1) * 1152 + ((lane >> 3) + 8 * i) * 36 + 2 * (lane & 7);
        *reinterpret_cast<float2*>(d) = make_float2(g[i].x, g[i].z);
        *reinterpret_cast<float2*>(d + 16) = make_float2(g[i].y, g[i].w);
      }
      const float4* rd = reinterpret_cast<const float4*>(wl + ((st + 1) & 1) * 1152 + (lane & 31) * 36 + (lane >> 5) * 16);
      const float4 a0 = rd[0], a1 = rd[1], a2 = rd[2], a3 = rd[3];
      bv[0] += a0.x * 1e-30f; bv[4] += a1.y * 1e-30f; bv[8] += a2.z * 1e-30f; bv[12] += a3.w * 1e-30f;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE & 16) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(av[(kk + 5 * j) & 15]), "v"(bv[kk]));
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(kk + 5 * j) & 15], bv[kk], acc[j], 0, 0, 0);
      }
      if (MODE & 2) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lane * 16 + kk * 1024 + st * 4096) & 65535), 0, 0);
        av[kk] += __uint_as_float(v.x) * 1e-30f;
        if (kk < 4 && !(MODE & 32)) {
          const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lane * 16 + kk * 2048 + st * 8192 + 512) & 65535), 0, 0);
          g[kk] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
        }
      }
      if (MODE & 4) { iv = iv * 3 + kk; iv ^= iv >> 3; iv += st; }          // ~2.5 integer instructions per kk
      if (MODE & 8) { fv = __builtin_fmaf(fv, 1.0001f, 0.5f); fv = __builtin_fmaf(fv, 0.9999f, -0.5f); fv = __builtin_fmaf(fv, 1.0002f, 0.25f); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = (float)iv + fv;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) t += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
// The 64-position tile: 8 accumulators (128 registers), two waves per SIMD, 128 MFMAs per K-step with the SAME 16 weight loads
// and 8 activation loads, 16 ds_write_b64 + 8 ds_read_b128, ~74 integer instructions.
template <int MODE>
__global__ __launch_bounds__(256, 2) void mix64(const float* __restrict__ table, float* out, int steps) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 2304];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wl = lds + wave * 2 * 2304;
  f32x16 acc[8];
  float av[16], bv[32];
  unsigned int s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 16; ++i) { s = s * 1664525u + 1013904223u; av[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f; }
  for (int i = 0; i < 32; ++i) { s = s * 1664525u + 1013904223u; bv[i] = __uint_as_float((s & 0x007FFFFFu) | 0x3F800000u) - 1.5f; }
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, 65536, 0x00020000);
  float4 g[8];
  for (int i = 0; i < 8; ++i) g[i] = make_float4(1, 2, 3, 4);
  int iv = lane;
  for (int st = 0; st < steps; ++st) {
    if (MODE & 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* d = wl + (st & 1) * 2304 + ((lane >> 3) + 8 * i) * 36 + 2 * (lane & 7);
        *reinterpret_cast<float2*>(d) = make_float2(g[i].x, g[i].z);
        *reinterpret_cast<float2*>(d + 16) = make_float2(g[i].y, g[i].w);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4* rd = reinterpret_cast<const float4*>(wl + ((st + 1) & 1) * 2304 + (32 * h + (lane & 31)) * 36 + (lane >> 5) * 16);
        const float4 a0 = rd[0], a1 = rd[1], a2 = rd[2], a3 = rd[3];
        bv[16 * h + 0] += a0.x * 1e-30f; bv[16 * h + 4] += a1.y * 1e-30f; bv[16 * h + 8] += a2.z * 1e-30f; bv[16 * h + 12] += a3.w * 1e-30f;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(kk + 5 * j) & 15], bv[kk], acc[j], 0, 0, 0);
        acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(kk + 5 * j) & 15], bv[16 + kk], acc[4 + j], 0, 0, 0);
      }
      if (MODE & 2) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lane * 16 + kk * 1024 + st * 4096) & 65535), 0, 0);
        av[kk] += __uint_as_float(v.x) * 1e-30f;
        if (kk < 8) {
          const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lane * 16 + kk * 2048 + st * 8192 + 512) & 65535), 0, 0);
          g[kk] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
        }
      }
      if (MODE & 4) { iv = iv * 3 + kk; iv ^= iv >> 3; iv += st; if (kk & 1) { iv = iv * 5 + 1; } }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = (float)iv;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) t += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
  float *sink, *table; hipMalloc(&sink, 1 << 22); hipMalloc(&table, 65536); hipMemset(table, 0, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 6000;                       // 6000 x 64 MFMAs x 64 cycles x 3 waves = 74 M cycles = ~31 ms at 2.4 GHz
  const char* names[40] = {"MFMA only", "+LDS", "+global", "+LDS+global", "+int", "+LDS+int", "+global+int", "+LDS+global+int",
                           "+fp", "+LDS+fp", "+global+fp", "+LDS+global+fp", "+int+fp", "+LDS+int+fp", "+global+int+fp", "+all",
                           "AGPR MFMA only", "AGPR +LDS", "AGPR +global", "AGPR +LDS+global", "AGPR +int", "", "", "AGPR +LDS+global+int", "AGPR +fp", "", "", "", "", "", "", "AGPR +all"};
#define RUN(M)                                                                                                       \
  { for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((mix<M>), dim3(768), dim3(256), 0, 0, table, sink, steps);   \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                       \
      if (rep) printf("%-18s %8.2f ms  %.1f TFLOP/s (%.3f of 157.3)\n", names[M], ms, 768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 1e12,   \
                      768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 157.3e12); } }
  names[3] = "+LDS+global (the kernel's mix)"; names[7] = "+LDS+global+int";
  RUN(0) RUN(3) RUN(7)
  { const char* n34 = "LDS-DMA staging + 16 weight loads"; const char* n38 = "LDS-DMA staging + weights + int";
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((mix<34>), dim3(768), dim3(256), 0, 0, table, sink, steps);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-34s %8.2f ms  %.1f TFLOP/s (%.3f of 157.3)\n", n34, ms, 768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 1e12, 768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 157.3e12); }
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((mix<38>), dim3(768), dim3(256), 0, 0, table, sink, steps);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-34s %8.2f ms  %.1f TFLOP/s (%.3f of 157.3)\n", n38, ms, 768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 1e12, 768. * 4 * steps * 64 * 4096. / (ms * 1e-3) / 157.3e12); } }
  return 0;
}
