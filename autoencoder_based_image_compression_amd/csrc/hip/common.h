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
    int pack;           // EAE_HIP_PACK: -1 (by shape) | 0 | 1: partial channel tiles of small layers as four-wave blocks (conv_gemm.hip: launch)
    int split_wpb;      // EAE_HIP_SPLIT_WPB: 1 = one-wave blocks in the split conv GEMM (default: 4 waves per block)
    int split_mute;     // test hook, debug entry point only: heads of cut tiles never publish, tails give up after ~1 ms
    int assume_partitioned;   // EAE_HIP_ASSUME_PARTITIONED=1: behave as on a device that is not one whole MI355X (tests of the de-tuned path)
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

// The tile order of the conv launches (`blockIdx.x & 7` = the XCD, conv_gemm_split.hip) and the sizing of the cut launches (waves
// an XCD holds at once = CUs / 8 x 4 x k) are built on ONE logical device = one whole MI355X: 8 XCDs x 32 CUs, workgroups handed
// to the XCDs round-robin (compute partition SPX). In DPX / QPX / CPX a logical device is 4 / 2 / 1 XCDs with 128 / 64 / 32 CUs:
// blocks b and b + 8 still share an XCD (the XCD count divides 8), so every hand-off stays inside one L2 and the bits are the
// same, but the per-XCD locality and the cut sizing are off. The CU count of the logical device tells the mode apart; on anything
// but 256 CUs of gfx950 the launches are never cut of their own accord (eae_hip_partition_info says what was found).
inline bool eae_is_whole_mi355x() {
    return eae_is_gfx950() && eae_compute_units() == 256 && !g_eae_launch_options.assume_partitioned;
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

// ---- the same normalisation without the parts of `sqrtf` and `/` that only extreme operands need (round 4) ---------------
// hipcc's correctly rounded sqrtf is: [a < 2^-96: scale by 2^32] v_sqrt_f32, the two neighbours of the result, two FMA residuals,
// two selects, [unscale] [class check: +-0 and +inf return themselves] -- 16 vector instructions; its correctly rounded a / b is
// v_div_scale x2, v_rcp_f32, a Newton step, the quotient with two FMA corrections, v_div_fmas, v_div_fixup -- 11. The bracketed
// parts do nothing for operands in the middle of the range: `sqrt_mid` and `div_mid` are the SAME instruction sequences without them
// (9 and 8 instructions), hence the same bits, for
//     sqrt_mid(a):    a >= 2^-96 (also +inf and NaN: the selects leave v_sqrt_f32's own result alone; NOT negative denormals,
//                     which v_sqrt_f32 flushes to -0: the guard's minimum keeps every negative a out)
//     div_mid(x, s):  2^-60 <= |x| <= 2^60 and 2^-20 <= s <= 2^40: v_div_scale_f32 returns its operand unscaled (exponent
//                     difference below 96, no denormal operand, reciprocal or quotient, numerator exponent above 23), so
//                     v_div_fmas is a plain FMA and v_div_fixup the identity.
// Proven on the GPU besides (tests/test_gpu_kernels.py through eae_hip_debug_check_mid_forms): sqrt_mid against sqrtf on EVERY float
// from 2^-96 up and every NaN, div_mid against `/` on 2^33 operand pairs of the guarded range including its corners. `gdn_tile` checks the range
// per wavefront (a min / max over the tile: half an instruction per element and bound) and falls back to sqrtf and `/` otherwise.
__device__ __forceinline__ float sqrt_mid(float a) {
    float s = __builtin_amdgcn_sqrtf(a);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, a), r_up = __builtin_fmaf(-s_up, s, a);
    s = 0.f >= r_dn ? s_dn : s;
    s = 0.f < r_up ? s_up : s;
    return s;
}
__device__ __forceinline__ float div_mid(float x, float s) {
    float y = __builtin_amdgcn_rcpf(s);
    y = __builtin_fmaf(__builtin_fmaf(-s, y, 1.f), y, y);
    float q = x * y;
    q = __builtin_fmaf(__builtin_fmaf(-s, q, x), y, q);
    return __builtin_fmaf(__builtin_fmaf(-s, q, x), y, q);
}
#define EAE_MID_A_MIN_SQRT 0x1p-96f
#define EAE_MID_A_MIN 0x1p-40f        // s = sqrt(a) in [2^-20, 2^40]
#define EAE_MID_A_MAX 0x1p80f
#define EAE_MID_X_MIN 0x1p-60f
#define EAE_MID_X_MAX 0x1p60f

// One wavefront's register tile: out(t, g, y) receives, group of four registers by group, y = x / sqrt(a) (GDN) or x * sqrt(a)
// (IGDN) with a[t][r] = d + beta formed by the caller. The range check runs over the whole tile first (a min / max chain per bound);
// the groups are then normalised one after the other (a scheduling fence per group keeps the register pressure where it was
// when the epilogues normalised and stored element by element). MID = false keeps hipcc's full sequences for a caller the check
// does not pay for: the latent stage's quarter tiles (16 values per lane: 0.056 ms with the mid forms against 0.051 without,
// profiles/r04_latent_mid_forms.log -- the two code paths cost it a wave per SIMD).
template <int NT, bool INVERSE, bool MID = true, typename Out>
__device__ __forceinline__ void gdn_tile(const f32x16 (&x)[NT], const f32x16 (&a)[NT], Out out) {
#ifndef EAE_NO_MID_FORMS
  if constexpr (MID) {
    float a_min = a[0][0], a_max = a[0][0], x_min = __builtin_fabsf(x[0][0]), x_max = x_min;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = (t == 0 ? 1 : 0); r < 16; ++r) {
            a_min = __builtin_fminf(a_min, a[t][r]);
            if (!INVERSE) {
                a_max = __builtin_fmaxf(a_max, a[t][r]);
                x_min = __builtin_fminf(x_min, __builtin_fabsf(x[t][r]));
                x_max = __builtin_fmaxf(x_max, __builtin_fabsf(x[t][r]));
            }
        }
    const bool mid = INVERSE ? a_min >= EAE_MID_A_MIN_SQRT
                             : (a_min >= EAE_MID_A_MIN && a_max <= EAE_MID_A_MAX && x_min >= EAE_MID_X_MIN && x_max <= EAE_MID_X_MAX);
    if (__builtin_amdgcn_ballot_w64(!mid) == 0ull) {        // every lane of the wavefront in range: one path for all
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float s = sqrt_mid(a[t][4 * g + q]);
                    y[q] = INVERSE ? x[t][4 * g + q] * s : div_mid(x[t][4 * g + q], s);
                }
                out(t, g, make_float4(y[0], y[1], y[2], y[3]));
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
  }
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s = sqrtf(a[t][4 * g + q]);
                y[q] = INVERSE ? x[t][4 * g + q] * s : x[t][4 * g + q] / s;
            }
            out(t, g, make_float4(y[0], y[1], y[2], y[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
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
        // `+ beta` after the matmul, then sqrt, then divide / multiply (tfutils.py:396); d becomes d + beta in place
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bt = *reinterpret_cast<const float4*>(vec_lds + EAE_C + 32 * t + 8 * g + cbase);
                d[t][4 * g + 0] = d[t][4 * g + 0] + bt.x;
                d[t][4 * g + 1] = d[t][4 * g + 1] + bt.y;
                d[t][4 * g + 2] = d[t][4 * g + 2] + bt.z;
                d[t][4 * g + 3] = d[t][4 * g + 3] + bt.w;
            }
        gdn_tile<4, inverse>(acc, d, [&](int t, int g, float4 y) {
            if (valid) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = y;
        });
    }
}
