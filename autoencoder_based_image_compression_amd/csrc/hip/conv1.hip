// conv1.hip -- conv_1 + bias_add + gdn_1 (eae/graph/components.py:119-125): tf.nn.conv2d 9x9, 1 -> 128 channels,
// stride 4, 'SAME' (pad 2 before / 3 after), fed by the raw uint8 luminance (eae/batching.py:95: astype(float32), no
// offset, no scale), then GDN over the 128 output channels.
//
// One block = 8 x 16 output positions; each of its 4 waves owns 2 tile rows = 32 positions x 128 channels and, after one
// barrier for the shared input patch, runs on its own (same structure as the conv GEMM wave kernel):
//   * the uint8 patch (37 rows x 69 columns, read as 18 aligned 32-bit words per row) is converted to f32 and staged in LDS
//     split by column phase (col mod 4), so the 32 positions of a wave read consecutive words (conflict-free);
//   * the product is transposed (weights = MFMA A operand): w1 rows (packed channel order, 82 rows, the last one zero)
//     stream from L1/L2 through a register ring of 16-byte loads, K = 81 taps in (kernel row, kernel column) order;
//   * bias, GDN (x^2 from accumulator registers, gamma through the ring) and 16-byte stores: common.h wave_epilogue.
// Bound: MFMA (GDN1 is 61 % of this stage's flops); the 32 B / input pixel output write is the HBM term.
#include "common.h"

namespace {
constexpr int TH = 8, TW = 16;
constexpr int K9 = 9, S4 = 4, KTAPS = 81, KPAD = 82;
constexpr int PR = TH * S4 + K9 - S4;    // 37 patch rows
constexpr int PW = 20;                   // words per (phase, row): 18 used; 4*PW = 80 = 16 (mod 32) -> two rows, 32 banks
constexpr int PATCH_FLOATS = 4 * PR * PW;                 // 2960
constexpr int RING = 6;                  // weight rows in flight (with 8 the kernel needs 172 registers: two waves per SIMD)

template <int NORM>
__global__ __launch_bounds__(256, 2) void conv1_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w_packed,
                                                       const float* __restrict__ bias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out,
                                                       int h, int win, int ho, int wo, int tiles_r, int tiles_c) {
    __shared__ __attribute__((aligned(16))) float lds[PATCH_FLOATS + 2 * EAE_C];
    float* patch = lds;                 // [4 phases][37][20]
    float* vec_lds = lds + PATCH_FLOATS; // bias[128], beta[128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    EAE_TRACE_MARK(0)
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tiles_c; b /= tiles_c;
    const int tr = b % tiles_r;
    const int img = b / tiles_r;
    const uint8_t* x_img = x + (size_t)img * h * win;
    const int r0 = tr * TH * S4 - 2, c0 = tc * TW * S4 - 2;   // SAME: pad_before = 2 (appendix A.2)
    // The patch as aligned 32-bit words: columns c0 - 2 + 4 D ... + 3 for D = 0..17 (c0 = 64 tc - 2, so the words are aligned
    // and -- the width being a multiple of 4 -- each lies entirely inside or outside the image). 37 x 18 words, at most three per
    // thread, all in flight together (one byte per load in a loop with its bounds check cost ten memory latencies per tile:
    // 33,000 of a wave's 107,000 cycles). Byte b of word D is patch column 4 D + b - 2: phase (b + 2) & 3, word D - 1 or D.
    constexpr int DW = 18, NDW = (PR * DW + 255) / 256;
    unsigned int pw[NDW];
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int i = tid + 256 * j;
        const int pr = i / DW, dq = i % DW;
        const int r = r0 + pr, c = c0 - 2 + 4 * dq;
        const bool ok = i < PR * DW && r >= 0 && r < h && c >= 0 && c < win;
        const unsigned int* src = reinterpret_cast<const unsigned int*>(x_img + (size_t)(ok ? r : 0) * win + (ok ? c : 0));
        const unsigned int t = *src;
        pw[j] = ok ? t : 0u;
    }
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int i = tid + 256 * j;
        const int pr = i / DW, dq = i % DW;
        if (i < PR * DW) {
            float* row = patch + pr * PW + dq;
            if (dq > 0) {
                row[2 * PR * PW - 1] = (float)(pw[j] & 0xFFu);             // phase 2, word D - 1
                row[3 * PR * PW - 1] = (float)((pw[j] >> 8) & 0xFFu);      // phase 3, word D - 1
            }
            row[0] = (float)((pw[j] >> 16) & 0xFFu);                       // phase 0, word D
            row[PR * PW] = (float)(pw[j] >> 24);                           // phase 1, word D
        }
    }
    if (tid < EAE_C) {
        vec_lds[tid] = bias ? bias[tid] : 0.f;
        vec_lds[EAE_C + tid] = NORM != EAE_NORM_NONE ? beta[tid] : 0.f;
    }
    __syncthreads();
    EAE_TRACE_MARK(1)

    const int hi = lane >> 5, lj = lane & 31;
    const int lr = 2 * wave + (lj >> 4), lc = lj & 15;        // this lane's position inside the tile
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(w_packed), 0, (int)(KPAD * EAE_C * sizeof(float)), 0x00020000);
    const int w_lane = (hi * EAE_C + lj * 4) * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 ring[RING];
#define EAE_C1_LOAD(dst_, kk_)                                                                                       \
    {                                                                                                                \
        const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane + (kk_) * 2 * EAE_C * 4, 0, 0);        \
        dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),                      \
                           __uint_as_float(v_.w));                                                                   \
    }
#pragma unroll
    for (int i = 0; i < RING; ++i) EAE_C1_LOAD(ring[i], i)
    const float* prow = patch + (4 * lr) * PW + lc;
    // the patch value of K-step kk: taps k = 2 kk (low half-wave) and 2 kk + 1 (high); k = 81 does not exist (zero weight row too)
#define EAE_C1_A(dst_, kk_)                                                                                          \
    {                                                                                                                \
        const int k0_ = 2 * (kk_), k1_ = 2 * (kk_) + 1;                                                              \
        const int u0_ = k0_ / K9, v0_ = k0_ % K9;                                                                    \
        const int u1_ = k1_ < KTAPS ? k1_ / K9 : 0, v1_ = k1_ < KTAPS ? k1_ % K9 : 0;                                \
        const int off0_ = ((v0_ & 3) * PR + u0_) * PW + (v0_ >> 2);                                                  \
        const int off1_ = ((v1_ & 3) * PR + u1_) * PW + (v1_ >> 2);                                                  \
        dst_ = prow[hi ? off1_ : off0_];                                                                             \
        if (k1_ >= KTAPS && hi) dst_ = 0.f;                                                                          \
    }
    float a_pipe[2];
    EAE_C1_A(a_pipe[0], 0)
#pragma unroll
    for (int kk = 0; kk < KPAD / 2; ++kk) {
        if (kk + 1 < KPAD / 2) EAE_C1_A(a_pipe[(kk + 1) & 1], kk + 1)      // read from LDS a step before its MFMAs
        const float a = a_pipe[kk & 1];
        const float4 wq = ring[kk % RING];
        acc[0] = mfma32(wq.x, a, acc[0]);                      // A = w1^T[co][k], B = patch^T[k][pos]
        acc[1] = mfma32(wq.y, a, acc[1]);
        acc[2] = mfma32(wq.z, a, acc[2]);
        acc[3] = mfma32(wq.w, a, acc[3]);
        if (kk + RING < KPAD / 2) { EAE_C1_LOAD(ring[kk % RING], kk + RING) }
        __builtin_amdgcn_sched_barrier(0);                     // the load stays RING steps ahead of its use (see wave_epilogue)
    }
#undef EAE_C1_A
    EAE_TRACE_MARK(2)
    const int pr = tr * TH + lr, pc = tc * TW + lc;
    const bool valid = pr < ho && pc < wo;
    float* o = out + (((size_t)img * ho + pr) * wo + pc) * EAE_C;
    wave_epilogue<NORM, RING>(acc, vec_lds, bias != nullptr, gamma, o, valid, lane);
    EAE_TRACE_MARK(4)
}

// TF filter [9][9][1][128] -> [82][128 packed co], row 81 = 0
__global__ void pack_conv1_kernel(const float* __restrict__ w, float* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KPAD * EAE_C) return;
    const int k = i / EAE_C, c = i % EAE_C;
    wp[k * EAE_C + packed_channel(c)] = k < KTAPS ? w[k * EAE_C + c] : 0.f;
}
}  // namespace

extern "C" int eae_hip_pack_conv9x9s4_weights(const float* w_tf, float* w_packed, void* stream) {
    if (!w_tf || !w_packed) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_conv1_kernel, dim3((KPAD * EAE_C + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_conv9x9s4_u8(const uint8_t* x, const float* w_packed, const float* bias, const float* gamma_packed,
                                    const float* beta, float* out, int n, int h, int w_in, void* stream) {
    if (!x || !w_packed || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (gamma_packed && !beta) return EAE_HIP_BAD_ARGUMENT;
    if ((h & 3) || (w_in & 3)) return EAE_HIP_BAD_SHAPE;
    if (reinterpret_cast<uintptr_t>(x) & 3u) return EAE_HIP_BAD_ARGUMENT;      // the kernel reads the image as aligned 32-bit words
    const int ho = h / 4, wo = w_in / 4;
    const int tiles_r = (ho + TH - 1) / TH, tiles_c = (wo + TW - 1) / TW;
    const dim3 grid(n * tiles_r * tiles_c);
    if (gamma_packed)
        hipLaunchKernelGGL((conv1_kernel<EAE_NORM_GDN>), grid, dim3(256), 0, (hipStream_t)stream, x, w_packed, bias,
                           gamma_packed, beta, out, h, w_in, ho, wo, tiles_r, tiles_c);
    else
        hipLaunchKernelGGL((conv1_kernel<EAE_NORM_NONE>), grid, dim3(256), 0, (hipStream_t)stream, x, w_packed, bias,
                           gamma_packed, beta, out, h, w_in, ho, wo, tiles_r, tiles_c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
