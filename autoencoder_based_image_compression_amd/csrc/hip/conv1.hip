// conv1.hip -- conv_1 + bias_add + gdn_1 (eae/graph/components.py:119-125): tf.nn.conv2d 9x9, 1 -> 128 channels,
// stride 4, 'SAME' (pad 2 before / 3 after), fed by the raw uint8 luminance (eae/batching.py:95: astype(float32), no
// offset, no scale), then GDN over the 128 output channels.
//
// One block = 8 x 16 output positions x 128 channels. The uint8 input patch (37 x 69 pixels) is converted to f32 and
// staged in LDS split by column phase (col mod 4), so that the 32 positions of a wave read consecutive words;
// w1 (81 x 128, padded to 82 rows) is staged once per block. The contraction (K = 81) and the GDN matmul (K = 128)
// both run on v_mfma_f32_32x32x2_f32. K order: kernel row, kernel column -- the oracle's order.
// Bound: MFMA (GDN1 is 61 % of this stage's flops); output write 32 B per input pixel is the HBM term.
#include "common.h"

namespace {
constexpr int TH = 8, TW = 16, TM = TH * TW;
constexpr int K9 = 9, S4 = 4, KTAPS = 81, KPAD = 82;
constexpr int PR = TH * S4 + K9 - S4;    // 37 patch rows
constexpr int PCOLS = TW * S4 + K9 - S4; // 69 patch columns
constexpr int PW = 20;                   // words per (phase, row): 18 used; 4*PW = 80 = 16 (mod 32) -> two rows, 32 banks
constexpr int PATCH_FLOATS = 4 * PR * PW;                 // 2960
constexpr int W_FLOATS = KPAD * EAE_C;                    // 10496
constexpr int LDS_MAIN = PATCH_FLOATS + W_FLOATS;
constexpr int LDS_EPI = TM * EAE_XS_STRIDE;
constexpr int LDS_FLOATS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;

__global__ __launch_bounds__(256, 2) void conv1_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out,
                                                       int h, int win, int ho, int wo, int tiles_r, int tiles_c) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* patch = lds;                 // [4 phases][37][20]
    float* Ws = lds + PATCH_FLOATS;     // [82][128]
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tiles_c; b /= tiles_c;
    const int tr = b % tiles_r;
    const int img = b / tiles_r;
    const uint8_t* x_img = x + (size_t)img * h * win;
    const int r0 = tr * TH * S4 - 2, c0 = tc * TW * S4 - 2;   // SAME: pad_before = 2 (appendix A.2)
    for (int i = tid; i < PR * PCOLS; i += 256) {
        const int pr = i / PCOLS, pc = i % PCOLS;
        const int r = r0 + pr, c = c0 + pc;
        float v = 0.f;
        if (r >= 0 && r < h && c >= 0 && c < win) v = (float)x_img[(size_t)r * win + c];
        patch[((pc & 3) * PR + pr) * PW + (pc >> 2)] = v;
    }
    for (int i = tid; i < KTAPS * EAE_C / 4; i += 256)
        reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(w)[i];
    if (tid < EAE_C) Ws[KTAPS * EAE_C + tid] = 0.f;          // k = 81: zero row (A is zero there too)
    __syncthreads();

    // this lane's position inside the tile: wave wm owns tile rows 2wm, 2wm+1
    const int li = lane & 31;
    const int lr = 2 * wm + (li >> 4), lc = li & 15;
    const int hi = lane >> 5;                                 // which k of the pair
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* prow = patch + (4 * lr) * PW + lc;
    const float* brow = Ws + hi * EAE_C + li;
#pragma unroll
    for (int kk = 0; kk < KPAD / 2; ++kk) {
        const int k0 = 2 * kk, k1 = 2 * kk + 1;
        constexpr int dummy = 0; (void)dummy;
        const int u0 = k0 / K9, v0 = k0 % K9;
        const int u1 = k1 < KTAPS ? k1 / K9 : 0, v1 = k1 < KTAPS ? k1 % K9 : 0;
        const int off0 = ((v0 & 3) * PR + u0) * PW + (v0 >> 2);
        const int off1 = ((v1 & 3) * PR + u1) * PW + (v1 >> 2);
        float a = prow[hi ? off1 : off0];
        if (k1 >= KTAPS && hi) a = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma32(a, brow[k0 * EAE_C + 32 * t], acc[t]);
    }
    const int col0 = li;
    if (bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bv = bias[col0 + 32 * t];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] + bv;
        }
    }
    float* out_img = out + (size_t)img * ho * wo * EAE_C;
    if (!gamma) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            const int pr = tr * TH + (m >> 4), pc = tc * TW + (m & 15);
            if (pr < ho && pc < wo)
#pragma unroll
                for (int t = 0; t < 4; ++t) out_img[((size_t)pr * wo + pc) * EAE_C + col0 + 32 * t] = acc[t][r];
        }
        return;
    }
    __syncthreads();
    float* Xs = lds;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(wm * 32 + acc_row32(r, lane)) * EAE_XS_STRIDE + col0 + 32 * t] = acc[t][r];
    __syncthreads();
    f32x16 d[4];
    gdn_denominator<4>(Xs, wm, lane, gamma, 0, d);   // gamma is packed (eae_hip_pack_gamma)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bt = beta[col0 + 32 * t];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            const int pr = tr * TH + (m >> 4), pc = tc * TW + (m & 15);
            if (pr < ho && pc < wo)
                out_img[((size_t)pr * wo + pc) * EAE_C + col0 + 32 * t] = gdn_apply(acc[t][r], d[t][r], bt, false);
        }
    }
}
}  // namespace

extern "C" int eae_hip_conv9x9s4_u8(const uint8_t* x, const float* w, const float* bias, const float* gamma,
                                    const float* beta, float* out, int n, int h, int w_in, void* stream) {
    if (!x || !w || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (gamma && !beta) return EAE_HIP_BAD_ARGUMENT;
    if ((h & 3) || (w_in & 3)) return EAE_HIP_BAD_SHAPE;
    const int ho = h / 4, wo = w_in / 4;
    const int tiles_r = (ho + TH - 1) / TH, tiles_c = (wo + TW - 1) / TW;
    hipLaunchKernelGGL(conv1_kernel, dim3(n * tiles_r * tiles_c), dim3(256), 0, (hipStream_t)stream, x, w, bias, gamma,
                       beta, out, h, w_in, ho, wo, tiles_r, tiles_c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
