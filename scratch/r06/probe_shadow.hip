// What one wave per SIMD can issue in the shadow of its own f32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each): cycles per MFMA of a
// loop of 64 MFMAs (two alternating accumulators) with k other instructions of one kind behind every MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_shadow probe_shadow.hip ; run: ./probe_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define MF "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\t"
#define MG "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n\t"

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void probe(long long* out, float* sink, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    float w = threadIdx.x * 0.001f, x = threadIdx.x * 0.002f;
    float v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f, v4 = 5.f, v5 = 6.f;
    int s0 = 1, s1 = 2;
    unsigned la = (threadIdx.x & 63) * 4;
    float d0, d1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OTHER                                                                                                    \
        if (KIND == 1) { if (K >= 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0)); if (K >= 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1)); \
                         if (K >= 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v2)); if (K >= 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v3)); \
                         if (K >= 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v4)); if (K >= 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v5)); } \
        if (KIND == 2) { if (K >= 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0)); if (K >= 2) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s1)); \
                         if (K >= 3) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0)); if (K >= 4) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s1)); \
                         if (K >= 5) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0)); if (K >= 6) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s1)); } \
        if (KIND == 3) { if (K >= 1) asm volatile("ds_read_b32 %0, %1" : "=v"(d0) : "v"(la)); if (K >= 2) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(d1) : "v"(la)); \
                         if (K >= 3) asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(d0) : "v"(la)); if (K >= 4) asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(d1) : "v"(la)); } \
        if (KIND == 4) { if (K >= 1) asm volatile("ds_write_b32 %1, %0" :: "v"(v0), "v"(la)); if (K >= 2) asm volatile("ds_write_b32 %1, %0 offset:256" :: "v"(v1), "v"(la)); } \
        if (KIND == 5) { if (K >= 1) asm volatile("v_and_b32 %0, %0, %0" : "+v"(s0)); if (K >= 2) asm volatile("v_and_b32 %0, %0, %0" : "+v"(s1)); \
                         if (K >= 3) asm volatile("v_and_b32 %0, %0, %0" : "+v"(s0)); if (K >= 4) asm volatile("v_and_b32 %0, %0, %0" : "+v"(s1)); }
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(a) : "v"(w), "v"(x));
            OTHER
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(b) : "v"(w), "v"(x));
            OTHER
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(a), "+a"(b));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a[0] + b[1] + v0 + v1 + v2 + v3 + v4 + v5 + s0 + s1 + d0 + d1;
}

template <int KIND, int K>
void run(const char* name) {
    long long* out; float* sink;
    hipMalloc(&out, 256 * 4 * 8); hipMalloc(&sink, 256 * 256 * 4);
    const int iters = 200;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 1024; ++i) sum += h[i];
    printf("%-28s k = %d: %.1f ticks per MFMA\n", name, K, sum / 1024 / iters / 64);
    hipFree(out); hipFree(sink);
}
int main() {
    run<0, 0>("MFMA alone");
    run<1, 1>("+ k v_add_f32 each"); run<1, 2>("+ k v_add_f32 each"); run<1, 3>("+ k v_add_f32 each"); run<1, 4>("+ k v_add_f32 each"); run<1, 6>("+ k v_add_f32 each");
    run<5, 2>("+ k v_and_b32 each"); run<5, 4>("+ k v_and_b32 each");
    run<2, 2>("+ k s_add_i32 each"); run<2, 4>("+ k s_add_i32 each"); run<2, 6>("+ k s_add_i32 each");
    run<3, 1>("+ k ds_read_b32 each"); run<3, 2>("+ k ds_read_b32 each"); run<3, 4>("+ k ds_read_b32 each");
    run<4, 1>("+ k ds_write_b32 each"); run<4, 2>("+ k ds_write_b32 each");
    return 0;
}
