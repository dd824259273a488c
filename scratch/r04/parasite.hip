// r04: does a resident wave that does NOTHING slow the conv GEMM down just by holding registers? One-wave blocks that sleep for `ticks`
// of s_memtime, with a VGPR allocation of 24 (fits next to three 160-register GEMM waves on a SIMD: 3 x 160 + 24 <= 512), 48 or 64
// (does not: the SIMD then holds two GEMM waves, and since a GEMM block needs a slot on every SIMD, the CU holds two blocks instead
// of three for as long as the sleeper stays).   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libparasite.so parasite.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int TOP> __global__ __launch_bounds__(64) void sleeper(unsigned long long ticks, float* sink);
#define SLEEPER(TOP_, REG_)                                                                                           \
    template <> __global__ __launch_bounds__(64) void sleeper<TOP_>(unsigned long long ticks, float* sink) {          \
        asm volatile("; hold " REG_ ::: REG_);                                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);                               \
        if (ticks == 12345) sink[0] = 1.f;                                                                            \
    }
SLEEPER(24, "v23")
SLEEPER(32, "v31")
SLEEPER(48, "v47")
SLEEPER(64, "v63")
extern "C" int parasite_launch(int vgprs, int blocks, unsigned long long ticks, float* sink, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (vgprs) {
    case 24: hipLaunchKernelGGL(sleeper<24>, dim3(blocks), dim3(64), 0, s, ticks, sink); break;
    case 32: hipLaunchKernelGGL(sleeper<32>, dim3(blocks), dim3(64), 0, s, ticks, sink); break;
    case 48: hipLaunchKernelGGL(sleeper<48>, dim3(blocks), dim3(64), 0, s, ticks, sink); break;
    case 64: hipLaunchKernelGGL(sleeper<64>, dim3(blocks), dim3(64), 0, s, ticks, sink); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
