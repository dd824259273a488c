// r06: shorter correctly rounded forms for the GDN epilogue's  y = x / sqrt(a)  (sqrt and division each correctly rounded, as the oracle's C).
//   mode 0: s = sqrt_ms(a) (v_rsq_f32 + Markstein's iteration, 8 instructions) against sqrtf(a) on EVERY float a in [2^-40, 2^80]
//   mode 1: q = div_fused(x, a) = x / RN(sqrt(a)) with the division's reciprocal taken from the square root's by-product (no v_rcp_f32)
//           against  x / sqrtf(a)  on mantissa-exhaustive slices: every a-mantissa x both exponent parities for a block of x-mantissas
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o probe_gdn_forms probe_gdn_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__device__ __forceinline__ float sqrt_ms(float a, float& h_out) {
    const float y = __builtin_amdgcn_rsqf(a);
    float g = a * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, a);
    h_out = h;
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float div_fused(float x, float a) {
    float h;
    const float s = sqrt_ms(a, h);
    float y = h + h;                                     // ~ 1 / sqrt(a)
#ifdef SECOND_ORDER
    const float e = __builtin_fmaf(-s, y, 1.f);
    y = __builtin_fmaf(__builtin_fmaf(e, e, e), y, y);
#elif defined(BIASED)
    y = __builtin_fmaf(__builtin_fmaf(-s, y, BIASED), y, y);       // the reciprocal pushed up by an ulp or so: breaks the one tie of the plain form
#else
    y = __builtin_fmaf(__builtin_fmaf(-s, y, 1.f), y, y);
#endif
    float q = x * y;
    q = __builtin_fmaf(__builtin_fmaf(-s, q, x), y, q);
    return __builtin_fmaf(__builtin_fmaf(-s, q, x), y, q);
}
__global__ __launch_bounds__(256) void sqrt_check(unsigned long long first, unsigned long long count, unsigned long long* out) {
    unsigned long long bad = 0, example = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const float a = __uint_as_float((unsigned int)(first + i));
        float h;
        const unsigned int got = __float_as_uint(sqrt_ms(a, h)), want = __float_as_uint(sqrtf(a));
        if (got != want) { bad++; example = first + i; }
    }
    if (bad) { atomicAdd(&out[0], bad); out[1] = example; }
}
// a = (1 + ma / 2^23) * 2^(ea) for ea in {0, 1}; x = (1 + mx / 2^23): all ma, both parities, mx in [mx0, mx0 + nmx)
__global__ __launch_bounds__(256) void div_check(unsigned int mx0, unsigned int nmx, unsigned long long* out) {
    unsigned long long bad = 0, example = 0;
    const unsigned long long total = (unsigned long long)nmx << 24;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256) {
        const unsigned int ma = (unsigned int)(i & 0x7FFFFFu), parity = (unsigned int)((i >> 23) & 1u), mx = mx0 + (unsigned int)(i >> 24);
        const float a = __uint_as_float(((127u + parity) << 23) | ma), x = __uint_as_float((127u << 23) | (mx & 0x7FFFFFu));
        const unsigned int got = __float_as_uint(div_fused(x, a)), want = __float_as_uint(x / sqrtf(a));
        if (got != want) { bad++; example = ((unsigned long long)__float_as_uint(x) << 32) | __float_as_uint(a); }
    }
    if (bad) { atomicAdd(&out[0], bad); out[1] = example; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    unsigned long long* dev; unsigned long long host[2];
    CK(hipMalloc(&dev, 16));
    CK(hipMemset(dev, 0, 16));
    const unsigned int lo = (127u - 40u) << 23, hi = (127u + 80u) << 23;
    hipLaunchKernelGGL(sqrt_check, dim3(8192), dim3(256), 0, 0, (unsigned long long)lo, (unsigned long long)(hi - lo), dev);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(host, dev, 16, hipMemcpyDeviceToHost));
    std::printf("sqrt_ms against sqrtf on every float of [2^-40, 2^80): %llu differ (example bits 0x%llx)\n", host[0], host[1]);
    const unsigned int blocks = argc > 1 ? (unsigned int)std::atoi(argv[1]) : 64u, per = 4096u;      // x-mantissas: blocks x 4096, spread over the range
    unsigned long long total_bad = 0;
    for (unsigned int b = 0; b < blocks; b++) {
        CK(hipMemset(dev, 0, 16));
        const unsigned int mx0 = (unsigned int)(((unsigned long long)b * 0x800000ull) / blocks);
        hipLaunchKernelGGL(div_check, dim3(16384), dim3(256), 0, 0, mx0, per, dev);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(host, dev, 16, hipMemcpyDeviceToHost));
        total_bad += host[0];
        if (host[0]) std::printf("  x-mantissas from 0x%x: %llu differ (x bits | a bits = 0x%llx)\n", mx0, host[0], host[1]);
    }
    std::printf("div_fused against x / sqrtf(a): %u x 4096 x-mantissas x 2^24 (a-mantissa, parity): %llu differ\n", blocks, total_bad);
    return 0;
}
