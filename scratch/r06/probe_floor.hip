// r06, VERDICT item 3 priced on the chip: cycles per binary decision of the encoder core's dependent chain (coder/lean_step.h: encode_step) with
// floor(p * range) in FP64 (the product's form) against an integer form, t = (P * R) >> 64 with P = floor(p * 2^48), R = range << 16:
//   x = mul_hi_u32(P_lo, R); t = (P_hi16 * (R >> 16) + (x >> 16)) >> 16.
// One wavefront alone on a SIMD, like a chain wave of the coder; 64 lanes with different probabilities and bits.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../autoencoder_based_image_compression_amd/csrc/coder -I../../include -o probe_floor probe_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "lean_step.h"
using namespace eae_lean;
__device__ __forceinline__ uint32_t middle32_int(const Interval& s, uint32_t p_lo, uint32_t p_hi) {
    const uint32_t range32 = 0xFFFF0000u - s.hc - s.lo;
    const uint32_t x = __umulhi(p_lo, range32);
    const uint32_t y = __umul24(p_hi, range32 >> 16) + (x >> 16);
    return s.lo + (y & 0xFFFF0000u);
}
template <int FORM>
__global__ __launch_bounds__(64) void chain_kernel(const double* probs, const uint32_t* bits, int steps, uint32_t* out, long long* cycles) {
    __shared__ double table[8][64];
    __shared__ uint2 itable[8][64];
    const int lane = threadIdx.x;
    for (int c = 0; c < 8; c++) {
        const double p = probs[c * 64 + lane];
        table[c][lane] = scale_probability(p);
        const unsigned long long P = (unsigned long long)floor(p * 281474976710656.0);
        itable[c][lane] = make_uint2((uint32_t)P, (uint32_t)(P >> 32));
    }
    __syncthreads();
    Interval s = interval_init();
    uint32_t acc = 0, word = bits[lane];
    const long long t0 = clock64();
    for (int i = 0; i < steps; i++) {
        const int c = i & 7;
        const bool one = (word >> (i & 31)) & 1u;
        uint32_t mid;
        if (FORM == 0) mid = middle32(s, table[c][lane]);
        else { const uint2 P = itable[c][lane]; mid = middle32_int(s, P.x, P.y); }
        narrow(s, mid, one);
        const Renorm r = renormalise(s);
        acc += (r.leaving & 0xFFFF0000u) | (r.n << 8) | r.k;
        if ((i & 31) == 31) word = word * 1664525u + 1013904223u;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc ^ s.lo ^ s.hc;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int steps = 20000;
    std::vector<double> probs(512);
    std::vector<uint32_t> bits(64);
    srand(3);
    for (auto& p : probs) p = 0.02 + 0.96 * (rand() / (double)RAND_MAX);
    for (auto& b : bits) b = (uint32_t)rand() * 2654435761u;
    double* dp; uint32_t* db; uint32_t* dout; long long* dcy;
    CK(hipMalloc(&dp, 512 * 8)); CK(hipMalloc(&db, 64 * 4)); CK(hipMalloc(&dout, 1024 * 64 * 4)); CK(hipMalloc(&dcy, 1024 * 8));
    CK(hipMemcpy(dp, probs.data(), 512 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bits.data(), 64 * 4, hipMemcpyHostToDevice));
    for (int blocks : {1, 256, 1024}) {
        std::vector<uint32_t> o0(blocks * 64), o1(blocks * 64);
        std::vector<long long> c0(blocks), c1(blocks);
        float ms0 = 0.f;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(64), 0, 0, dp, db, steps, dout, dcy);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms0, e0, e1));
            CK(hipMemcpy(o0.data(), dout, blocks * 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(c0.data(), dcy, blocks * 8, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(64), 0, 0, dp, db, steps, dout, dcy);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o1.data(), dout, blocks * 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), dcy, blocks * 8, hipMemcpyDeviceToHost));
        }
        int differ = 0;
        for (int i = 0; i < 64; i++) differ += o0[i] != o1[i];
        std::printf("%4d wave(s): FP64 form %.1f cycles per decision, integer form %.1f (clock64 ticks / %d steps; lanes whose results differ: %d of 64 -- 48-bit P unverified here); "
                    "FP64 launch %.3f ms by HIP events = %.1f ns per decision = %.2f GHz if a tick is a cycle\n",
                    blocks, c0[0] / (double)steps, c1[0] / (double)steps, steps, differ, ms0, ms0 * 1e6 / steps, (c0[0] / (double)steps) / (ms0 * 1e6 / steps));
    }
    return 0;
}
