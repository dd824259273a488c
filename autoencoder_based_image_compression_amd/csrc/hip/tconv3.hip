// tconv3.hip -- transpose_conv_3 (eae/graph/components.py:79-83: tf.nn.conv2d_transpose 9x9, 128 -> 1 channel,
// stride 4, 'SAME', no bias) fused with what follows it on the path: tls.cast_bt601 (tools/tools.py:93) and the
// squared error of tls.psnr_2d (tools.py:873-875).
//
// Formulation: output pixel (4p'+a, 4q'+b) receives kernel taps u = a + 2 - 4*dr,
// v = b + 2 - 4*dc from the input sites (p'+dr, q'+dc), dr, dc in {-1, 0, +1}. So for a tile of input sites
//     D[site][phase = 4a+b] = sum of X[site + (dr,dc)][ci] * Wp[(dr,dc)][ci][phase]
// is a GEMM with N = 16 phases and K = 9 * 128, Wp holding zeros where 0 <= u,v <= 8 fails (x * 0 adds +0: exact).
// K order = the oracle's: 32-channel block (outer), then (dr, dc) descending == (u, v) ascending, then the channel inside
// the block. v_mfma_f32_16x16x4_f32 (A = sites x k, B = k x phases).
//
// One block = 2 waves = 4 x 16 sites -> 16 x 64 output pixels; wave w owns tile rows 2w, 2w+1 (two independent
// accumulator chains, enough to keep the 32-cycle MFMA issue rate alone on its SIMD). The input patch (6 x 18 sites x 128
// channels) is staged ONCE in LDS with each site's channels permuted to [ci mod 4][ci / 4] (region stride 36, site
// stride 152 floats): the 8 k-values a lane needs per (channel block, neighbour) are then two conflict-free
// ds_read_b128 per tile row. Weights are pre-packed per lane ([block][neighbour][lane][8]) and stream from L1/L2 through
// a register ring four steps (1024 MFMA cycles) ahead: no barrier inside the K loop. Epilogue: the 16 x 64 pixel tile goes
// through LDS, 8 pixels per thread, BT.601 cast, exact integer squared error (wave shuffle -> one u64 atomic per block).
// Bound: MFMA for the contraction (2,304 issued / 1,296 algorithmic FLOP per pixel); 32 B/px read is the HBM term.
#include "common.h"

namespace {
constexpr int TH = 4, TW = 16;
#ifndef EAE_T3_PASSES
#define EAE_T3_PASSES 2
#endif
constexpr int PASSES = EAE_T3_PASSES;         // the input patch is staged PASSES times, 128 / PASSES channels at a time
constexpr int HALF_C = EAE_C / PASSES;
constexpr int REGION = HALF_C / 4 + 4;        // floats per (site, ci mod 4): HALF_C / 4 used + 4 pad (20 or 12)
constexpr int PS = PASSES == 2 ? 88 : 56;     // floats per site: 4 regions, rounded so that 8 consecutive sites hit 8 distinct
constexpr int PATCH_R = TH + 2, PATCH_C = TW + 2;     //   16-byte bank groups (PS/4 = 22 or 14) -> conflict-free b128 reads
constexpr int PATCH_FLOATS = PATCH_R * PATCH_C * PS;   // 38,016 B (4 blocks = 8 waves per CU) or 24,192 B (6 blocks)
constexpr int Q = HALF_C / 4;                 // float4 per site per pass
constexpr int LOADS = (PATCH_R * PATCH_C * Q + 127) / 128;   // per thread per pass (14 or 7)
constexpr int BATCHES = LOADS / 7;
constexpr int STEPS = 4 * 9;                  // (channel block, neighbour)
constexpr int RING = 4;                       // steps of weights in flight (2 float4 each)

__global__ __launch_bounds__(128, 2) void tconv3_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                        float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                        const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                        int h, int win, int tiles_r, int tiles_c) {
    __shared__ __attribute__((aligned(16))) float patch[PATCH_FLOATS];
    __shared__ unsigned int red[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tiles_c; b /= tiles_c;
    const int tr = b % tiles_r;
    const int img = b / tiles_r;
    const float* x_img = x + (size_t)img * h * win * EAE_C;
    const int r0 = tr * TH - 1, c0 = tc * TW - 1;
    // weights: lane-private 32 bytes per step; start the ring before touching the patch
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(wq), 0, (int)(STEPS * 64 * 8 * sizeof(float)), 0x00020000);
    float4 ring[RING][2];
#define EAE_T3_LOAD(slot_, step_)                                                                                    \
    {                                                                                                                \
        const u32x4 v0_ = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, lane * 32, (step_) * 64 * 32, 0);            \
        const u32x4 v1_ = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, lane * 32 + 16, (step_) * 64 * 32, 0);       \
        ring[slot_][0] = make_float4(__uint_as_float(v0_.x), __uint_as_float(v0_.y), __uint_as_float(v0_.z),         \
                                     __uint_as_float(v0_.w));                                                        \
        ring[slot_][1] = make_float4(__uint_as_float(v1_.x), __uint_as_float(v1_.y), __uint_as_float(v1_.z),         \
                                     __uint_as_float(v1_.w));                                                        \
    }
#pragma unroll
    for (int i = 0; i < RING; ++i) EAE_T3_LOAD(i, i)
    // The patch is staged in two passes of 64 channels (the K order is channel-block outer anyway): 38 KB of LDS per block
    // instead of 66 KB, so four blocks = eight waves share a CU and cover each other's staging and MFMA latencies (with the
    // whole patch resident only two blocks fitted: one wave per SIMD, MFMA issue stalls fully exposed).
    // Pass p: 108 sites x 16 float4, zero outside the image (zero-fill at THIS layer, appendix C.3); channel ci of the
    // pass lands at (ci & 3) * REGION + (ci >> 2). 1728 float4 = 13.5 per thread, two batches of 7 loads in flight.
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, kq = lane >> 4;
    const float* a_base = patch + ((2 * wave + 1) * PATCH_C + (i16 + 1)) * PS + kq * REGION;   // site (row 2w, col i16)
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass) __syncthreads();                     // both waves are done with the previous 64 channels
#pragma unroll
        for (int batch = 0; batch < BATCHES; ++batch) {
            float4 v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid + 128 * (7 * batch + j);
                const int site = i / Q, q = i % Q;
                const int r = r0 + site / PATCH_C, c = c0 + site % PATCH_C;
                const bool ok = i < PATCH_R * PATCH_C * Q && (unsigned)r < (unsigned)h && (unsigned)c < (unsigned)win;
                const float4* src = reinterpret_cast<const float4*>(x_img + ((size_t)(ok ? r : 0) * win + (ok ? c : 0)) * EAE_C +
                                                                    HALF_C * pass + 4 * q);
                const float4 t = *src;
                v[j] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int i = tid + 128 * (7 * batch + j);
                if (i < PATCH_R * PATCH_C * Q) {
                    float* dst = patch + (i / Q) * PS + (i % Q);
                    dst[0] = v[j].x; dst[REGION] = v[j].y; dst[2 * REGION] = v[j].z; dst[3 * REGION] = v[j].w;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ls = 0; ls < STEPS / PASSES; ++ls) {
            const int step = pass * (STEPS / PASSES) + ls;
            const int cbl = ls / 9, nb = ls % 9;            // channel block inside the pass, neighbour
            const int dr = 1 - nb / 3, dc = 1 - nb % 3;     // (+1,+1), (+1,0), ... (-1,-1)
            const float* a0p = a_base + (dr * PATCH_C + dc) * PS + 8 * cbl;
            const float4 a00 = *reinterpret_cast<const float4*>(a0p), a01 = *reinterpret_cast<const float4*>(a0p + 4);
            const float4 a10 = *reinterpret_cast<const float4*>(a0p + PATCH_C * PS), a11 = *reinterpret_cast<const float4*>(a0p + PATCH_C * PS + 4);
            const float4 w0 = ring[step % RING][0], w1 = ring[step % RING][1];
            acc0 = mfma16(a00.x, w0.x, acc0); acc1 = mfma16(a10.x, w0.x, acc1);
            acc0 = mfma16(a00.y, w0.y, acc0); acc1 = mfma16(a10.y, w0.y, acc1);
            acc0 = mfma16(a00.z, w0.z, acc0); acc1 = mfma16(a10.z, w0.z, acc1);
            acc0 = mfma16(a00.w, w0.w, acc0); acc1 = mfma16(a10.w, w0.w, acc1);
            acc0 = mfma16(a01.x, w1.x, acc0); acc1 = mfma16(a11.x, w1.x, acc1);
            acc0 = mfma16(a01.y, w1.y, acc0); acc1 = mfma16(a11.y, w1.y, acc1);
            acc0 = mfma16(a01.z, w1.z, acc0); acc1 = mfma16(a11.z, w1.z, acc1);
            acc0 = mfma16(a01.w, w1.w, acc0); acc1 = mfma16(a11.w, w1.w, acc1);
            if (step + RING < STEPS) EAE_T3_LOAD(step % RING, step + RING)
        }
    }
    __syncthreads();                                   // both waves are done reading the patch
    // ---- epilogue: 16 x 64 pixel tile through LDS, then 8 consecutive pixels per thread ----------------------------
    float* ot = patch;                                 // [16][64]
    {
        const int a = i16 >> 2, bq = i16 & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ot[(8 * wave + a) * 64 + 4 * (4 * kq + r) + bq] = acc0[r];
            ot[(8 * wave + 4 + a) * 64 + 4 * (4 * kq + r) + bq] = acc1[r];
        }
    }
    __syncthreads();
    const int ho = 4 * h, wo = 4 * win;
    const int prow = tid >> 3, pcol = (tid & 7) * 8;
    const int gr = tr * TH * 4 + prow, gc = tc * TW * 4 + pcol;
    unsigned int se = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int gcc = gc + 4 * half;
        if (gr < ho && gcc < wo) {     // wo is a multiple of 4, so the 4 pixels are inside together
            const float4 v = *reinterpret_cast<const float4*>(ot + prow * 64 + pcol + 4 * half);
            const size_t o = ((size_t)img * ho + gr) * wo + gcc;
            if (out_f32) *reinterpret_cast<float4*>(out_f32 + o) = v;
            if (out_u8 || ref) {
                // tls.cast_bt601: clip to [16, 235], round half to even, uint8
                const unsigned int q0 = (unsigned int)round_half_even(fminf(fmaxf(v.x, 16.f), 235.f));
                const unsigned int q1 = (unsigned int)round_half_even(fminf(fmaxf(v.y, 16.f), 235.f));
                const unsigned int q2 = (unsigned int)round_half_even(fminf(fmaxf(v.z, 16.f), 235.f));
                const unsigned int q3 = (unsigned int)round_half_even(fminf(fmaxf(v.w, 16.f), 235.f));
                if (out_u8) *reinterpret_cast<unsigned int*>(out_u8 + o) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
                if (ref) {
                    const unsigned int rv = *reinterpret_cast<const unsigned int*>(ref + o);
                    const int d0 = (int)(rv & 0xFF) - (int)q0, d1 = (int)((rv >> 8) & 0xFF) - (int)q1;
                    const int d2 = (int)((rv >> 16) & 0xFF) - (int)q2, d3 = (int)(rv >> 24) - (int)q3;
                    se += (unsigned int)(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
                }
            }
        }
    }
    if (ref && sse) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) se += __shfl_down(se, off, 64);
        if (lane == 0) red[wave] = se;
        __syncthreads();
        if (tid == 0) atomicAdd(&sse[img], (unsigned long long)red[0] + red[1]);
    }
}

// TF filter [9][9][1][128] -> per-lane fragments [4 channel blocks][9 neighbours (dr,dc) descending][64 lanes][8]:
// lane = kq * 16 + phase holds W[u][v][ci = 32 cb + 4 kk + kq] for kk = 0..7, zero where the tap falls outside 9x9.
__global__ void pack_tconv3_kernel(const float* __restrict__ w_tf, float* __restrict__ wq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= STEPS * 64 * 8) return;
    const int kk = i & 7, lane = (i >> 3) & 63, step = i >> 9;
    const int cb = step / 9, nb = step % 9;
    const int phase = lane & 15, kq = lane >> 4;
    const int ci = 32 * cb + 4 * kk + kq;
    const int dr = 1 - nb / 3, dc = 1 - nb % 3;
    const int u = (phase >> 2) + 2 - 4 * dr, v = (phase & 3) + 2 - 4 * dc;
    wq[i] = (u >= 0 && u < 9 && v >= 0 && v < 9) ? w_tf[(u * 9 + v) * EAE_C + ci] : 0.f;
}
}  // namespace

extern "C" int eae_hip_pack_tconv9x9s4_weights(const float* w_tf, float* w_phase, void* stream) {
    if (!w_tf || !w_phase) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_tconv3_kernel, dim3((STEPS * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_tf, w_phase);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_tconv9x9s4_luma(const float* x, const float* w_phase, float* out_f32, uint8_t* out_u8,
                                       const uint8_t* ref_u8, uint64_t* sse, int n, int h, int w_in, void* stream) {
    if (!x || !w_phase || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (!out_f32 && !out_u8 && !ref_u8) return EAE_HIP_BAD_ARGUMENT;
    if ((ref_u8 != nullptr) != (sse != nullptr)) return EAE_HIP_BAD_ARGUMENT;
    const int tiles_r = (h + TH - 1) / TH, tiles_c = (w_in + TW - 1) / TW;
    hipLaunchKernelGGL(tconv3_kernel, dim3(n * tiles_r * tiles_c), dim3(128), 0, (hipStream_t)stream, x, w_phase, out_f32,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), h, w_in, tiles_r, tiles_c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
