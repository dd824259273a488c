// conv_gemm.h -- definitions shared by the conv GEMM kernels (conv_gemm.hip: the block-cooperative form and the forms for
// layers too small to fill the machine; conv_gemm_split.hip: one item per wave, the launch's last tiles cut at a K-step boundary).
#pragma once
#include "common.h"
#include "latent_body.h"

namespace eae_conv_gemm {

constexpr int KC = 32;           // K-step (input channels per LDS slab)
constexpr int AS_STRIDE = 36;    // floats per position: [2 parities][16] + 4 pad (16-byte aligned, conflict-free b128)
constexpr int XS_STRIDE = EAE_XS_STRIDE;   // epilogue tile [TM][128] (+1)
constexpr int MAX_TAPS = 25;
constexpr int TILE_H = 8;

struct PhaseDesc {
    int out_a, out_b;            // output pixel = position * out_stride + (out_a, out_b)
    int ntaps;
    // per tap, packed in one dword (sub-dword kernarg arrays get copied to scratch by the compiler):
    //   bits 0-7 off_r + 8, bits 8-15 off_c + 8 (input pixel = position * in_stride + (off_r, off_c)),
    //   bits 16-23 slab index into the packed weights [T][128][128]
    int tap[MAX_TAPS];
};
inline int pack_tap(int off_r, int off_c, int widx) { return (off_r + 8) | ((off_c + 8) << 8) | (widx << 16); }

struct ConvGemmParams {
    const float* in;     // [N][Hin][Win][128]
    float* out;          // [N][Hout][Wout][128]
    const float* w;      // packed [T][128 ci][128 co permuted]
    const float* bias;   // [128] or nullptr
    const float* gamma;  // packed [128 k][128 c permuted] or nullptr
    const float* beta;   // [128]
    int norm;            // EAE_NORM_*
    int n, hin, win, hp, wp, hout, wout;
    int in_stride, out_stride;
    int tiles_r, tiles_c, n_phases;
    unsigned long long* stamps;   // diagnostic only (eae_hip_debug_set_stamp_buffer): 8 x u64 per wave, else nullptr
    unsigned int* split_ws;             // conv_gemm_split.hip: zeroed workspace of SPLIT_WORDS words (left zeroed), or nullptr
    int split;                          // conv_gemm_split.hip: 1 = the last tiles of each XCD's share are cut in two
    int split_resident_waves_per_xcd;   // conv_gemm_split.hip: how many tiles get cut (the waves an XCD holds at once)
    int split_spin_limit;               // conv_gemm_split.hip: polls of a tail for its head before it gives up (error word)
    int split_mute_heads;               // conv_gemm_split.hip: TEST HOOK (eae_hip_debug_set_split_mute): heads never publish
    // conv_gemm_split.hip, norm == NORM_LATENT / NORM_LATENT_PLAIN: the latent stage behind conv_3 (latent_body.h)
    const float* map_mean;              // [128] or nullptr
    const float* bin_widths;            // [128]
    const float* gamma_out;             // packed, inverse_gdn_4 (NORM_LATENT)
    const float* beta_out;
    LatentOut latent;
    PhaseDesc phase[4];
};
// epilogues beyond EAE_NORM_*: conv_3 + bias -> gdn_3 -> quantiser -> inverse_gdn_4 (fixed bin widths), or -> quantiser only
constexpr int NORM_LATENT = 3, NORM_LATENT_PLAIN = 4;

// conv_gemm_split.hip
constexpr int SPLIT_WORDS = 256 + 8 * 1024;     // [255] error word (tails that gave up); 8 x 1024 "head published" flags
constexpr int SPLIT_ERROR_WORD = 255;
// cut: -1 = decide from the shape, 0 = whole tiles only, 1..3 = cut (sized for that many resident waves per SIMD).
// Returns EAE_HIP_OK, or 1 when the device does not suit the kernel (nothing launched).
int launch_split(ConvGemmParams& p, hipStream_t stream, int cut);

}  // namespace eae_conv_gemm
