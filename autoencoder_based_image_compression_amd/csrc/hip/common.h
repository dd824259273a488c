// common.h -- shared device helpers for the gfx950 kernels (wave64, f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eae_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define EAE_C 128  // channels of every hidden layer (eae/graph/constants.py:42-44)

// Kernel-form overrides (experiments and the parity tests of every form): read from the environment ONCE, when the library is
// loaded, never on the launch path; tests that switch forms inside one process call eae_hip_debug_reload_launch_options(). The
// hand-off fault injection is not reachable from the environment at all (eae_hip_debug_set_split_mute). misc.hip owns the object.
struct EaeLaunchOptions {
    char gemm;          // EAE_HIP_GEMM: 0 (by shape) | 's' cut forced | 'u' whole tiles | 'w' one-tile-per-wave kernel | 'l' LDS slabs
    int split_waves;    // EAE_HIP_SPLIT_WAVES: 1..3 (with 's'), default 3
    int force_tile;     // EAE_HIP_FORCE_TILE: 0 (by shape) | 32 | 64 | 128
    int force_nt;       // EAE_HIP_FORCE_NT: 0 (by shape) | 1 | 2 | 4
    char latent;        // EAE_HIP_LATENT: 'q' (default) | 'w' | 'l'
    int split_mute;     // test hook, debug entry point only: heads of cut tiles never publish, tails give up after ~1 ms
};
extern EaeLaunchOptions g_eae_launch_options;

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

// Phase timestamps for experiments (scratch/variant.sh builds ONE kernel file with -DEAE_TRACE and reads them back through
// eae_hip_trace_read): mark i of wave w = clock64() at that point, in eae_trace_buf[8 w + i]. Not part of the product build.
#ifdef EAE_TRACE
static __device__ long long eae_trace_buf[65536 * 8];
#define EAE_TRACE_MARK(i_)                                                                                           \
    {                                                                                                                \
        const unsigned int wid_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                               \
        if ((threadIdx.x & 63) == 0 && wid_ < 65536u) eae_trace_buf[(size_t)wid_ * 8 + (i_)] = clock64();            \
    }
extern "C" int eae_hip_trace_read(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(eae_trace_buf), (size_t)n * sizeof(long long));
}
#else
#define EAE_TRACE_MARK(i_)
#endif

// Compute units of the current device (cached per device; 0 when the runtime cannot tell).
inline int eae_compute_units() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cached[dev] = prop.multiProcessorCount;
    }
    return cached[dev];
}

// True on the one architecture the hand-written hand-off protocols of this library were validated on (gfx950: in-order
// workgroup dispatch, sc1 write-through stores / L1-bypassing loads inside an XCD's L2). Cached per device.
inline bool eae_is_gfx950() {
    static int cached[16] = {0};       // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        const char* a = prop.gcnArchName;
        cached[dev] = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 2;
    }
    return cached[dev] == 1;
}

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
        } else if constexpr (NT == 2) {
            const float2 gv = *reinterpret_cast<const float2*>(g_rd + (size_t)2 * kk * EAE_C);
            g[0] = gv.x; g[1] = gv.y;
        } else {
            g[0] = g_rd[(size_t)2 * kk * EAE_C];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) d[t] = mfma32(x2, g[t], d[t]);
    }
}
__device__ __forceinline__ float gdn_apply(float x, float d, float beta, bool inverse) {
    const float s = sqrtf(d + beta);   // `+ beta` after the matmul, then sqrt, then divide / multiply (tfutils.py:396)
    return inverse ? x * s : x / s;
}


// ---- register-resident epilogue of the TRANSPOSED wave tile (conv_gemm.hip, conv1.hip) --------------------------------
// acc[t][r] at lane (hi = lane >> 5, lj = lane & 31) holds channel 32 t + (r & 3) + 8 (r >> 2) + 4 hi of position lj
// (the C/D layout of v_mfma_f32_32x32x2_f32 when the weights are the A operand). Steps:
//   bias_add (vec_lds[0..127]);  d^T[c][pos] = sum_k gamma[k][c] * x^2[pos][k], k ascending;  x (/ or *) sqrt(d + beta)
//   (vec_lds[128..255]);  16-byte stores of 4 consecutive channels.
// x^2 is fed to the MFMA straight from the accumulator registers: one v_permlane32_swap per register pair turns the
// (k | k+4), (k+1 | k+5) half-wave contents into the natural k pairs (k | k+1), (k+4 | k+5). gamma rows (packed channel
// order, eae_hip_pack_gamma) stream through a register ring of 16-byte buffer loads.
__device__ __forceinline__ void swap_halves(float& a, float& b) {
    // lanes 32-63 of a <-> lanes 0-31 of b (v_permlane32_swap_b32)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// RING: gamma k-pairs (16 bytes per lane each) in flight ahead of the MFMAs that use them
template <int NORM, int RING = 8>
__device__ __forceinline__ void wave_epilogue(f32x16 (&acc)[4], const float* vec_lds, bool has_bias,
                                              const float* __restrict__ gamma_packed, float* o, bool valid, int lane) {
    const int hi = lane >> 5, lj = lane & 31;
    const int cbase = 4 * hi;                 // channel of (t, g, q) = 32 t + 8 g + cbase + q
    o += cbase;
    if (has_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(vec_lds + 32 * t + 8 * g + cbase);
                acc[t][4 * g + 0] = acc[t][4 * g + 0] + bv.x;
                acc[t][4 * g + 1] = acc[t][4 * g + 1] + bv.y;
                acc[t][4 * g + 2] = acc[t][4 * g + 2] + bv.z;
                acc[t][4 * g + 3] = acc[t][4 * g + 3] + bv.w;
            }
    }
    if constexpr (NORM == EAE_NORM_NONE) {
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(o + 32 * t + 8 * g) =
                        make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        }
        return;
    } else {
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(gamma_packed), 0, (int)(EAE_C * EAE_C * sizeof(float)), 0x00020000);
        const int g_lane = (hi * EAE_C + lj * 4) * 4;        // byte offset inside a k-pair of rows
        f32x16 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[t][r] = 0.f;
        float4 ring[RING];
#define EAE_G_LOAD(dst_, kk_)                                                                                        \
        {                                                                                                            \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_lane + (kk_) * 2 * EAE_C * 4, 0, 0);    \
            dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),                  \
                               __uint_as_float(v_.w));                                                               \
        }
#pragma unroll
        for (int i = 0; i < RING; ++i) EAE_G_LOAD(ring[i], i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // registers 4g..4g+3 hold channels (8g + q | 8g + 4 + q) in the (low | high) half-waves
                float s0 = acc[t][4 * g + 0], s1 = acc[t][4 * g + 1], s2 = acc[t][4 * g + 2], s3 = acc[t][4 * g + 3];
                swap_halves(s0, s1);     // s0 = (8g+0 | 8g+1), s1 = (8g+4 | 8g+5)
                swap_halves(s2, s3);     // s2 = (8g+2 | 8g+3), s3 = (8g+6 | 8g+7)
                const float xs[4] = {s0, s2, s1, s3};            // k pairs in ascending order
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = 16 * t + 4 * g + e;            // k = 2 kk + hi
                    const float x2 = xs[e] * xs[e];
                    const float4 gq = ring[kk % RING];
                    d[0] = mfma32(gq.x, x2, d[0]);
                    d[1] = mfma32(gq.y, x2, d[1]);
                    d[2] = mfma32(gq.z, x2, d[2]);
                    d[3] = mfma32(gq.w, x2, d[3]);
                    if (kk + RING < EAE_C / 2) { EAE_G_LOAD(ring[kk % RING], kk + RING) }
                    // keep this load HERE: left alone, the scheduler sinks every load to one step before its use (a two-deep
                    // ring whatever RING says), and each K-step then waits for an L2 round trip
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#undef EAE_G_LOAD
        EAE_TRACE_MARK(3)
        constexpr bool inverse = NORM == EAE_NORM_IGDN;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bt = *reinterpret_cast<const float4*>(vec_lds + EAE_C + 32 * t + 8 * g + cbase);
                const float4 y = make_float4(gdn_apply(acc[t][4 * g + 0], d[t][4 * g + 0], bt.x, inverse),
                                             gdn_apply(acc[t][4 * g + 1], d[t][4 * g + 1], bt.y, inverse),
                                             gdn_apply(acc[t][4 * g + 2], d[t][4 * g + 2], bt.z, inverse),
                                             gdn_apply(acc[t][4 * g + 3], d[t][4 * g + 3], bt.w, inverse));
                if (valid) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = y;
            }
    }
}
