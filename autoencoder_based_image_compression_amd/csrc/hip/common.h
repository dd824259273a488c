// common.h -- shared device helpers for the gfx950 kernels (wave64, f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eae_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define EAE_C 128  // channels of every hidden layer (eae/graph/constants.py:42-44)

#define EAE_HIP_CHECK_LAUNCH()                         \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact f32 FMA chain, k ascending (probed on MI355X:
// scratch/probe_mfma.hip). Lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D reg r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D reg r: row 4*(l>>4) + r, col l&15.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row32(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of the logical grid so that
// neighbouring tiles (shared input halos, same image) hit the same L2. Bijective for any grid size. Speed only.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// numpy.round == round-half-to-even == v_rndne_f32
__device__ __forceinline__ float round_half_even(float x) { return rintf(x); }

// ---- packed channel order -----------------------------------------------------------------------------------------
// A wave owns NT 32-wide output-channel tiles; lane j = lane & 31 needs channels {32 t + j}. Weight rows and gamma rows
// are stored with their 128 output channels permuted so that those NT values are contiguous:
//     packed[j * 4 + t] = row[32 * t + j]
// -> one 16-byte load per lane per k instead of four 4-byte loads (eae_hip_pack_* do this once per model).
__host__ __device__ __forceinline__ int packed_channel(int c) { return (c & 31) * 4 + (c >> 5); }

// ---- shared GDN / IGDN tile epilogue (tfutils.py:393-397, 505-509) ------------------------------------------------
// Xs: LDS tile [rows][XS_STRIDE] holding x (after bias) for 128 channels. This wave owns rows wm*32..wm*32+31 and the
// NT channel tiles t0..t0+NT-1. Computes, for lane column j = lane & 31,
//   d[t][r] = sum_{k ascending} Xs[row(r)][k]^2 * gamma[k][32 (t0 + t) + j]        (one f32 FMA chain per element)
// gamma_packed is read straight from global memory (64 KB, L2-resident): one 16-/8-byte load per lane per k.
#define EAE_XS_STRIDE 129
template <int NT>
__device__ __forceinline__ void gdn_denominator(const float* Xs, int wm, int lane, const float* __restrict__ gamma_packed,
                                                int t0, f32x16 (&d)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] = 0.f;
    const float* x_rd = Xs + (wm * 32 + (lane & 31)) * EAE_XS_STRIDE + (lane >> 5);
    const float* g_rd = gamma_packed + (size_t)(lane >> 5) * EAE_C + (lane & 31) * 4 + t0;
#pragma unroll 8
    for (int kk = 0; kk < EAE_C / 2; ++kk) {
        const float xv = x_rd[2 * kk];
        const float x2 = xv * xv;
        float g[NT];
        if constexpr (NT == 4) {
            const float4 gv = *reinterpret_cast<const float4*>(g_rd + (size_t)2 * kk * EAE_C);
            g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
        } else {
            const float2 gv = *reinterpret_cast<const float2*>(g_rd + (size_t)2 * kk * EAE_C);
            g[0] = gv.x; g[1] = gv.y;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) d[t] = mfma32(x2, g[t], d[t]);
    }
}
__device__ __forceinline__ float gdn_apply(float x, float d, float beta, bool inverse) {
    const float s = sqrtf(d + beta);   // `+ beta` after the matmul, then sqrt, then divide / multiply (tfutils.py:396)
    return inverse ? x * s : x / s;
}
