// latent_body.h -- the latent stage on one wavefront's register tile (32 positions x 128 channels in the accumulator layout
// of the conv GEMM epilogue), shared by latent_wave_kernel (latent.hip: the stage as its own launch) and by the conv_3
// epilogue of conv_gemm_split_kernel (conv_gemm_split.hip: the stage fused behind the convolution). See latent.hip for
// what the stage computes and which reference lines it replaces.
#pragma once
#include "common.h"

#ifndef EAE_LATENT_RING
#define EAE_LATENT_RING 8
#endif
#ifdef EAE_LATENT_TRACE               // scratch/variant.sh: clock ticks per phase, summed behind the three data checks
#define LAT_MARK(i_) { const long long t_ = clock64(); if (o.checks && lane == 0) atomicAdd(&o.checks[16 + (i_)], (unsigned int)((t_ - lat_last) >> 4)); lat_last = t_; }
#else
#define LAT_MARK(i_)
#endif

// The layout is the one the conv GEMM epilogue uses (common.h wave_epilogue): lane (hi = lane >> 5, lj = lane & 31) holds, for
// position lj, the 64 channels 32 t + 8 g + 4 hi + q in x[t][4 g + q]. x^2 goes from those registers straight into the MFMA
// (B operand) through one v_permlane32_swap per register pair, gamma rows (packed channel order) stream through a register
// ring as the A operand: no LDS tile, no barrier, and the same k-ascending FMA chain per element as gdn_denominator.
template <bool INVERSE>
__device__ __forceinline__ void wave_gdn_inplace(f32x16 (&x)[4], const float* beta_lds, const float* __restrict__ gamma_packed,
                                                 int lane) {
    constexpr int RING = EAE_LATENT_RING;
    const int hi = lane >> 5, lj = lane & 31;
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gamma_packed), 0, (int)(EAE_C * EAE_C * sizeof(float)), 0x00020000);
    const int g_lane = (hi * EAE_C + lj * 4) * 4;
    f32x16 d[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] = 0.f;
    float4 ring[RING];
#define EAE_L_LOAD(dst_, kk_)                                                                                            \
    {                                                                                                                    \
        const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_lane + (kk_) * 2 * EAE_C * 4, 0, 0);            \
        dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w));  \
    }
#pragma unroll
    for (int i = 0; i < RING; ++i) EAE_L_LOAD(ring[i], i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s0 = x[t][4 * g + 0], s1 = x[t][4 * g + 1], s2 = x[t][4 * g + 2], s3 = x[t][4 * g + 3];
            swap_halves(s0, s1);     // s0 = channels (8g+0 | 8g+1), s1 = (8g+4 | 8g+5) in the (low | high) half-waves
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
                if (kk + RING < EAE_C / 2) { EAE_L_LOAD(ring[kk % RING], kk + RING) }
            }
        }
    }
#undef EAE_L_LOAD
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bt = *reinterpret_cast<const float4*>(beta_lds + 32 * t + 8 * g + 4 * hi);
            d[t][4 * g + 0] = d[t][4 * g + 0] + bt.x;
            d[t][4 * g + 1] = d[t][4 * g + 1] + bt.y;
            d[t][4 * g + 2] = d[t][4 * g + 2] + bt.z;
            d[t][4 * g + 3] = d[t][4 * g + 3] + bt.w;
        }
    gdn_tile<4, INVERSE>(x, d, [&](int t, int g, float4 y) {
        x[t][4 * g + 0] = y.x; x[t][4 * g + 1] = y.y; x[t][4 * g + 2] = y.z; x[t][4 * g + 3] = y.w;
    });
}


// Pointers of the stage's outputs and per-map vectors (device memory; each output nullable except as eae_hip_latent_stage says)
struct LatentOut {
    float* y_out;                 // [N][hw][128] latents after gdn_3
    float* shifted_out;           // [N][hw][128] quantised + mean
    float* t_out;                 // [N][hw][128] after inverse_gdn_4 (IGDN_OUT only)
    int16_t* symbols;             // [N][128][hw]
    unsigned int* nonzero;        // [N][128]
    unsigned int* checks;         // [3]
};

// v: the tile (conv_3 + bias); vec: LDS, beta_in | beta_out | map_mean | bin_widths (128 floats each). This lane's position:
// image `img`, pixel `pix` of `hw`, `valid` when inside the image. Arithmetic: quantize.hip statement for statement, the two
// normalisations as gdn.hip (same bits as the separate kernels, tests/test_gpu_latent.py).
template <bool GDN_IN, bool IGDN_OUT>
__device__ __forceinline__ void wave_latent_body(f32x16 (&v)[4], const float* vec, const float* __restrict__ gamma_in,
                                                 const float* __restrict__ gamma_out, const LatentOut& o, bool valid, long img,
                                                 int pix, int hw, int lane) {
    const int hi = lane >> 5;
#ifdef EAE_LATENT_TRACE
    long long lat_last = clock64();
#endif
    if (GDN_IN) wave_gdn_inplace<false>(v, vec, gamma_in, lane);
    LAT_MARK(0)
    unsigned int bad = 0, not_quantized = 0, altered = 0;
    const size_t obase = ((size_t)(valid ? img : 0) * hw + (valid ? pix : 0)) * EAE_C + 4 * hi;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * t + 8 * g + 4 * hi;
            const float4 m4 = *reinterpret_cast<const float4*>(vec + 2 * EAE_C + c0);
            const float4 b4 = *reinterpret_cast<const float4*>(vec + 3 * EAE_C + c0);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
            float sh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float yv = v[t][4 * g + q];
                // quantize.hip, statement for statement
                const float centered = yv - mm[q];
                const float rr = round_half_even(centered / bw[q]);
                const float cq = bw[q] * rr;
                const float rs = round_half_even(cq / bw[q]);
                sh[q] = cq + mm[q];
                if (valid) {
                    if (!(fabsf(rs) < 32768.f)) bad++;
                    if (!(fabs((double)cq - (double)centered) < 1.5e-10)) not_quantized++;
                    if (!((float)(int16_t)(int)rs * bw[q] == centered)) altered++;
                    if (cq != 0.f && o.nonzero) o.nonzero[img * EAE_C + c0 + q] = 1u;      // benign race: every writer stores 1
                    // planar symbols: the lanes of a half-wave write neighbouring pixels of one map
                    if (o.symbols) o.symbols[((size_t)img * EAE_C + c0 + q) * hw + pix] = (int16_t)(int)rs;
                }
            }
            if (valid && o.y_out)
                *reinterpret_cast<float4*>(o.y_out + obase + 32 * t + 8 * g) =
                    make_float4(v[t][4 * g], v[t][4 * g + 1], v[t][4 * g + 2], v[t][4 * g + 3]);
            if (valid && o.shifted_out) *reinterpret_cast<float4*>(o.shifted_out + obase + 32 * t + 8 * g) = make_float4(sh[0], sh[1], sh[2], sh[3]);
            v[t][4 * g + 0] = sh[0]; v[t][4 * g + 1] = sh[1]; v[t][4 * g + 2] = sh[2]; v[t][4 * g + 3] = sh[3];
        }
    LAT_MARK(1)
    if (o.checks) {
        if (bad) atomicAdd(&o.checks[0], bad);
        if (not_quantized) atomicAdd(&o.checks[1], not_quantized);
        if (altered) atomicAdd(&o.checks[2], altered);
    }
    if (IGDN_OUT) {
        wave_gdn_inplace<true>(v, vec + EAE_C, gamma_out, lane);
        LAT_MARK(2)
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(o.t_out + obase + 32 * t + 8 * g) =
                        make_float4(v[t][4 * g], v[t][4 * g + 1], v[t][4 * g + 2], v[t][4 * g + 3]);
        }
        LAT_MARK(3)
    }
}
