// tconv3.hip -- transpose_conv_3 (eae/graph/components.py:79-83: tf.nn.conv2d_transpose 9x9, 128 -> 1 channel,
// stride 4, 'SAME', no bias) fused with what follows it on the path: tls.cast_bt601 (tools/tools.py:93) and the
// squared error of tls.psnr_2d (tools.py:873-875).
//
// Formulation: output pixel (4p'+a, 4q'+b) receives kernel taps u = a + 2 - 4*dr,
// v = b + 2 - 4*dc from the input sites (p'+dr, q'+dc), dr, dc in {-1, 0, +1}. So for a tile of input sites
//     D[site][phase = 4a+b] = sum over (dr, dc) descending, ci ascending of X[site + (dr,dc)][ci] * Wp[(dr,dc)][ci][phase]
// is a GEMM with N = 16 phases and K = 9 * 128, Wp holding zeros where 0 <= u,v <= 8 fails (x * 0 adds +0: exact).
// K order = the oracle's: 32-channel block (outer), then (dr, dc) descending == (u, v) ascending, then the channel inside
// the block. v_mfma_f32_16x16x4_f32.
//
// One block = 4 x 16 sites -> 16 x 64 output pixels; the input patch 6 x 18 sites x 128 channels sits in LDS for the
// whole block (site stride 130 floats: 16 sites x 2 k read 32 distinct banks); the weights of one 32-channel block
// ([9 neighbours][32][16] = 18 KB) are staged in LDS per block of channels (4 stages per tile). Bound: MFMA for the contraction; 32 B/pixel read + 1 B/pixel write is the HBM term.
#include "common.h"

namespace {
constexpr int TH = 4, TW = 16;
constexpr int PS = 130;                       // floats per site in the LDS patch
constexpr int PATCH_R = TH + 2, PATCH_C = TW + 2;
constexpr int PATCH_FLOATS = PATCH_R * PATCH_C * PS;   // 14040
constexpr int WSLAB = 9 * 32 * 16;            // one channel block: [9 neighbours][32 ci][16 phases]
constexpr int LDS_FLOATS = PATCH_FLOATS + WSLAB;       // 18648 floats = 74,592 B -> 2 blocks / CU

__global__ __launch_bounds__(256, 2) void tconv3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                        const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                        int h, int win, int tiles_r, int tiles_c) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* patch = lds;
    float* Wl = lds + PATCH_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tiles_c; b /= tiles_c;
    const int tr = b % tiles_r;
    const int img = b / tiles_r;
    const float* x_img = x + (size_t)img * h * win * EAE_C;
    const int r0 = tr * TH - 1, c0 = tc * TW - 1;
    // patch: 108 sites x 32 float4; zero outside the image (zero-fill at THIS layer, appendix C.3)
    for (int i = tid; i < PATCH_R * PATCH_C * 32; i += 256) {
        const int site = i >> 5, q = i & 31;
        const int r = r0 + site / PATCH_C, c = c0 + site % PATCH_C;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0 && r < h && c >= 0 && c < win) v = *reinterpret_cast<const float4*>(x_img + ((size_t)r * win + c) * EAE_C + 4 * q);
        float2* dst = reinterpret_cast<float2*>(patch + site * PS + 4 * q);   // 8-byte aligned (PS even)
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, kq = lane >> 4;
    for (int cb = 0; cb < EAE_C / 32; ++cb) {
        __syncthreads();                       // patch staged (cb == 0) / previous channel block's weights consumed
        for (int i = tid; i < WSLAB / 4; i += 256)
            reinterpret_cast<float4*>(Wl)[i] = reinterpret_cast<const float4*>(wp + (size_t)cb * WSLAB)[i];
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 9; ++nb) {
            const int dr = 1 - nb / 3, dc = 1 - nb % 3;     // (+1,+1), (+1,0), ... (-1,-1)
            const float* a_rd = patch + ((wave + dr + 1) * PATCH_C + (i16 + dc + 1)) * PS + cb * 32 + kq;
            const float* b_rd = Wl + nb * 32 * 16 + kq * 16 + i16;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc = mfma16(a_rd[4 * kk], b_rd[4 * kk * 16], acc);
        }
    }
    __syncthreads();
    // ---- epilogue: 16 x 64 pixel tile through LDS, then 4 consecutive pixels per thread ----------------------------
    float* ot = lds;                                   // [16][64]; the patch is dead after the last barrier
    {
        const int a = i16 >> 2, bq = i16 & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(4 * wave + a) * 64 + 4 * (4 * kq + r) + bq] = acc[r];
    }
    __syncthreads();
    const int ho = 4 * h, wo = 4 * win;
    const int prow = tid >> 4, pcol = (tid & 15) * 4;
    const int gr = tr * TH * 4 + prow, gc = tc * TW * 4 + pcol;
    unsigned int se = 0;
    if (gr < ho && gc < wo) {      // wo is a multiple of 4, so the 4 pixels are inside together
        const float4 v = *reinterpret_cast<const float4*>(ot + prow * 64 + pcol);
        const size_t o = ((size_t)img * ho + gr) * wo + gc;
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
                se = (unsigned int)(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
            }
        }
    }
    if (ref && sse) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) se += __shfl_down(se, off, 64);
        __shared__ unsigned int red[4];
        if (lane == 0) red[wave] = se;
        __syncthreads();
        if (tid == 0) atomicAdd(&sse[img], (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
    }
}

// TF filter [9][9][1][128] -> phase-packed [4 channel blocks][9 neighbours (dr,dc) descending][32 ci][16 phases], zeros
// where the tap falls outside the 9x9 kernel.
__global__ void pack_tconv3_kernel(const float* __restrict__ w_tf, float* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * EAE_C * 16) return;
    const int phase = i & 15, cin = (i >> 4) & 31, nb = (i >> 9) % 9, cb = i / (9 * 32 * 16);
    const int ci = cb * 32 + cin;
    const int dr = 1 - nb / 3, dc = 1 - nb % 3;
    const int u = (phase >> 2) + 2 - 4 * dr, v = (phase & 3) + 2 - 4 * dc;
    wp[i] = (u >= 0 && u < 9 && v >= 0 && v < 9) ? w_tf[(u * 9 + v) * EAE_C + ci] : 0.f;
}
}  // namespace

extern "C" int eae_hip_pack_tconv9x9s4_weights(const float* w_tf, float* w_phase, void* stream) {
    if (!w_tf || !w_phase) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_tconv3_kernel, dim3((9 * EAE_C * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_tf, w_phase);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_tconv9x9s4_luma(const float* x, const float* w_phase, float* out_f32, uint8_t* out_u8,
                                       const uint8_t* ref_u8, uint64_t* sse, int n, int h, int w_in, void* stream) {
    if (!x || !w_phase || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (!out_f32 && !out_u8 && !ref_u8) return EAE_HIP_BAD_ARGUMENT;
    if ((ref_u8 != nullptr) != (sse != nullptr)) return EAE_HIP_BAD_ARGUMENT;
    const int tiles_r = (h + TH - 1) / TH, tiles_c = (w_in + TW - 1) / TW;
    hipLaunchKernelGGL(tconv3_kernel, dim3(n * tiles_r * tiles_c), dim3(256), 0, (hipStream_t)stream, x, w_phase, out_f32,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), h, w_in, tiles_r, tiles_c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
