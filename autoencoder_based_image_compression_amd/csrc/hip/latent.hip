// latent.hip -- everything between the last analysis convolution and the first synthesis convolution in ONE pass over the
// latent variables (fixed-bin-width model; the learned-bin-width model has neither normalisation):
//   gdn_3                                           eae/graph/components.py:139-142, tfutils.py:393-397
//   y - map_mean, quantize_per_map, int16 symbols   reconstructing_eae_kodak.py:170-178, tools.py:883-929, compression.py:142
//   count_nb_deads flags, the three data checks     tools.py:294-320, 130-132, 372-375, compression.py:149-153
//   + map_mean, inverse_gdn_4                       reconstructing_eae_kodak.py:192, components.py:53-58, tfutils.py:505-509
// The separate kernels (conv epilogue / gdn_kernel, quantize_kernel, gdn_kernel) each stream the 128-channel latents
// through HBM; here the 32 positions x 128 channels of a tile stay in registers -- of four waves, 32 channels each
// (latent_quarter_kernel, the default), or of one wave (latent_wave_kernel); latent_kernel keeps 64 positions per block in
// LDS -- and both 128 x 128 contractions run on the MFMA.
// Arithmetic, operation by operation, is that of gdn.hip and quantize.hip (same helpers), so the results are the same bits.
#include "latent_body.h"

#include <cstdlib>

namespace {
// Block-cooperative form (EAE_HIP_LATENT_LDS=1, kept for comparison): 64 positions per block (2 waves x 32 positions x 128
// channels) in 50 KB of LDS. Measured at Kodak batch 24 (36,864 positions): 127-131 us in the bench, against 188 us for
// 128-position blocks and 162 us for 32 positions with one channel tile per wave; the register-resident wave form below
// (round 1's default) takes 121 us; the three separate kernels 25 + 75 + 45 us plus two launch gaps. With about one wave per SIMD
// every form is bound by exposed latencies (the gamma rows come from L2 behind an 8-deep ring), not by MFMA or HBM rates.
constexpr int WAVES = 2;
constexpr int TM = WAVES * 32;
constexpr int SYM_STRIDE = TM + 2;

template <bool GDN_IN, bool IGDN_OUT>
// x, y_out, shifted_out and t_out carry no __restrict__: eae_hip_conv5x5s2_latent runs the stage IN PLACE on the convolution's
// output for small layers (x == t_out or shifted_out); every lane loads its elements before it stores them.
__global__ __launch_bounds__(WAVES * 64) void latent_kernel(const float* x, const float* __restrict__ gamma_in,
                                                     const float* __restrict__ beta_in, const float* __restrict__ map_mean,
                                                     const float* __restrict__ bin_widths, const float* __restrict__ gamma_out,
                                                     const float* __restrict__ beta_out, float* y_out,
                                                     float* shifted_out, float* t_out,
                                                     int16_t* __restrict__ symbols, unsigned int* __restrict__ nonzero,
                                                     unsigned int* __restrict__ checks, long rows, int hw) {
    __shared__ __attribute__((aligned(16))) float Xs[TM * EAE_XS_STRIDE];   // [TM][129]: x, then y / shifted in place
    __shared__ int16_t Sy[EAE_C * SYM_STRIDE];                              // [128 maps][TM + 2]: symbols, for the planar write
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const long row0 = (long)blockIdx.x * TM;
    for (int i = tid; i < TM * (EAE_C / 4); i += WAVES * 64) {
        const int r = i >> 5, q = i & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < rows) v = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * EAE_C + 4 * q);
        float* dst = Xs + r * EAE_XS_STRIDE + 4 * q;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    const int col0 = lane & 31;
    // image and pixel of a row of this tile without a division per element: one division per block when the tile spans at
    // most two images (hw >= TM), the general formula otherwise
    const long img0 = row0 / hw;
    const int pix0 = (int)(row0 - img0 * hw);
    const bool simple = hw >= TM;
    f32x16 d[4];
    if (GDN_IN) {
        gdn_denominator<4>(Xs, wm, lane, gamma_in, 0, d);
        __syncthreads();                    // every wave is done reading x of every row before y overwrites it
    }
    unsigned int bad = 0, not_quantized = 0, altered = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = col0 + 32 * t;
        const float m = map_mean ? map_mean[c] : 0.f;
        const float bw = bin_widths[c];
        const float bt = GDN_IN ? beta_in[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mrow = wm * 32 + acc_row32(r, lane);
            const long row = row0 + mrow;
            float yv = Xs[mrow * EAE_XS_STRIDE + c];
            if (GDN_IN) yv = gdn_apply(yv, d[t][r], bt, false);
            // quantize.hip, statement for statement
            const float centered = yv - m;
            const float rr = round_half_even(centered / bw);
            const float cq = bw * rr;
            const float rs = round_half_even(cq / bw);
            const float shifted = cq + m;
            if (row < rows) {
                if (!(fabsf(rs) < 32768.f)) bad++;
                if (!(fabs((double)cq - (double)centered) < 1.5e-10)) not_quantized++;
                if (!((float)(int16_t)(int)rs * bw == centered)) altered++;
                if (y_out) y_out[(size_t)row * EAE_C + c] = yv;
                if (shifted_out) shifted_out[(size_t)row * EAE_C + c] = shifted;
                if (cq != 0.f && nonzero) {
                    const long img = simple ? img0 + (pix0 + mrow >= hw ? 1 : 0) : row / hw;
                    nonzero[img * EAE_C + c] = 1u;                               // benign race: every writer stores 1
                }
            }
            Xs[mrow * EAE_XS_STRIDE + c] = shifted;
            Sy[c * SYM_STRIDE + mrow] = (int16_t)(int)rs;
        }
    }
    if (checks) {
        if (bad) atomicAdd(&checks[0], bad);
        if (not_quantized) atomicAdd(&checks[1], not_quantized);
        if (altered) atomicAdd(&checks[2], altered);
    }
    __syncthreads();
    if (symbols) {
        // planar write: consecutive threads cover consecutive positions of one map
        for (int i = tid; i < EAE_C * TM; i += WAVES * 64) {
            const int ch = i / TM, px = i % TM;
            const long row = row0 + px;
            if (row < rows) {
                const bool next = pix0 + px >= hw;
                const long img = simple ? img0 + (next ? 1 : 0) : row / hw;
                const int pix = simple ? pix0 + px - (next ? hw : 0) : (int)(row % hw);
                symbols[((size_t)img * EAE_C + ch) * hw + pix] = Sy[ch * SYM_STRIDE + px];
            }
        }
    }
    if (IGDN_OUT) {
        gdn_denominator<4>(Xs, wm, lane, gamma_out, 0, d);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = col0 + 32 * t;
            const float bt = beta_out[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = wm * 32 + acc_row32(r, lane);
                if (row0 + mrow < rows)
                    t_out[(size_t)(row0 + mrow) * EAE_C + c] = gdn_apply(Xs[mrow * EAE_XS_STRIDE + c], d[t][r], bt, true);
            }
        }
    }
}

// ---- one wavefront per 32 positions, everything in registers: latent_body.h -------------------------------------------------
template <bool GDN_IN, bool IGDN_OUT>
__global__ __launch_bounds__(64) void latent_wave_kernel(const float* x, const float* __restrict__ gamma_in,
                                                         const float* __restrict__ beta_in, const float* __restrict__ map_mean,
                                                         const float* __restrict__ bin_widths, const float* __restrict__ gamma_out,
                                                         const float* __restrict__ beta_out, float* y_out,
                                                         float* shifted_out, float* t_out,
                                                         int16_t* __restrict__ symbols, unsigned int* __restrict__ nonzero,
                                                         unsigned int* __restrict__ checks, long rows, int hw) {
    __shared__ __attribute__((aligned(16))) float vec[4 * EAE_C];      // beta_in | beta_out | map_mean | bin_widths
    const int lane = threadIdx.x, hi = lane >> 5, lj = lane & 31;
    {
        const float2 z = make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(vec + 2 * lane) = GDN_IN ? *reinterpret_cast<const float2*>(beta_in + 2 * lane) : z;
        *reinterpret_cast<float2*>(vec + EAE_C + 2 * lane) = IGDN_OUT ? *reinterpret_cast<const float2*>(beta_out + 2 * lane) : z;
        *reinterpret_cast<float2*>(vec + 2 * EAE_C + 2 * lane) = map_mean ? *reinterpret_cast<const float2*>(map_mean + 2 * lane) : z;
        *reinterpret_cast<float2*>(vec + 3 * EAE_C + 2 * lane) = *reinterpret_cast<const float2*>(bin_widths + 2 * lane);
    }
    const long row = (long)blockIdx.x * 32 + lj;
    const bool valid = row < rows;
    const long img = row / hw;
    const int pix = (int)(row - img * hw);
    const float* xrow = x + (size_t)(valid ? row : 0) * EAE_C + 4 * hi;
    f32x16 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 q = valid ? *reinterpret_cast<const float4*>(xrow + 32 * t + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[t][4 * g + 0] = q.x; v[t][4 * g + 1] = q.y; v[t][4 * g + 2] = q.z; v[t][4 * g + 3] = q.w;
        }
    __syncthreads();
    const LatentOut o{y_out, shifted_out, t_out, symbols, nonzero, checks};
    wave_latent_body<GDN_IN, IGDN_OUT>(v, vec, gamma_in, gamma_out, o, valid, img, pix, hw, lane);
}

// ---- four wavefronts per 32 positions, one 32-channel tile each (the default) ------------------------------------------------
// The register-resident form above puts 1,152 wavefronts on 1,024 SIMDs at Kodak batch 24: most SIMDs hold ONE wave whose
// ~60,000 cycles are a quarter MFMA and the rest vector work (64 sqrt, 192 divisions, 64 double-precision checks per lane)
// with nothing to overlap it, and an eighth of them hold two. Here a block of four waves shares the tile: every wave keeps
// the squares of all 128 channels of the 32 positions in registers as the B operand (each wave squares and orders its own 32
// channels, then they exchange through LDS, lane-private slots) but
// accumulates, normalises, quantises and stores only its own 32 channels. 4.5 waves per SIMD, each a quarter as long: the
// MFMAs of one overlap the vector work of the others. Same FMA chain per element, same statements: same bits.
// x2[t][4 g + e]: the B operand of K-step kk = 16 t + 4 g + e, i.e. the squares of the tile in the order and lane placement the
// MFMA wants them (squares_for_mfma below), all 128 channels; the wave accumulates its own channel tile w.
template <int RING>
__device__ __forceinline__ f32x16 quarter_denominator(const f32x16 (&x2)[4], const float* __restrict__ gamma_packed, int w, int lane) {
    const int hi = lane >> 5, lj = lane & 31;
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(gamma_packed), 0, (int)(EAE_C * EAE_C * sizeof(float)), 0x00020000);
    const int g_lane = (hi * EAE_C + lj * 4 + w) * 4;      // packed row k = 2 kk + hi, column 4 lj + w = channel 32 w + lj
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    float ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(g_rsrc, g_lane, i * 2 * EAE_C * 4, 0));
#pragma unroll
    for (int kk = 0; kk < EAE_C / 2; ++kk) {                // k = 2 kk + hi
        d = mfma32(ring[kk % RING], x2[kk >> 4][kk & 15], d);
        if (kk + RING < EAE_C / 2)
            ring[kk % RING] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(g_rsrc, g_lane, (kk + RING) * 2 * EAE_C * 4, 0));
    }
    return d;
}

// One channel tile (16 registers: channel 8 g + 4 hi + q in [4 g + q]) -> its squares as MFMA B operands: [4 g + e] feeds K-step
// 4 g + e of the tile, k pairs ascending (the swaps of wave_gdn_inplace, done once by the tile's owner instead of by every reader)
__device__ __forceinline__ f32x16 squares_for_mfma(const f32x16& v) {
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s0 = v[4 * g + 0], s1 = v[4 * g + 1], s2 = v[4 * g + 2], s3 = v[4 * g + 3];
        swap_halves(s0, s1);     // s0 = channels (8g+0 | 8g+1), s1 = (8g+4 | 8g+5) in the (low | high) half-waves
        swap_halves(s2, s3);     // s2 = (8g+2 | 8g+3), s3 = (8g+6 | 8g+7)
        r[4 * g + 0] = s0 * s0; r[4 * g + 1] = s2 * s2; r[4 * g + 2] = s1 * s1; r[4 * g + 3] = s3 * s3;
    }
    return r;
}

template <bool GDN_IN, bool IGDN_OUT>
__global__ __launch_bounds__(256) void latent_quarter_kernel(const float* x, const float* __restrict__ gamma_in,
                                                             const float* __restrict__ beta_in, const float* __restrict__ map_mean,
                                                             const float* __restrict__ bin_widths, const float* __restrict__ gamma_out,
                                                             const float* __restrict__ beta_out, float* y_out,
                                                             float* shifted_out, float* t_out,
                                                             int16_t* __restrict__ symbols, unsigned int* __restrict__ nonzero,
                                                             unsigned int* __restrict__ checks, long rows, int hw) {
    __shared__ __attribute__((aligned(16))) float vec[4 * EAE_C];          // beta_in | beta_out | map_mean | bin_widths
    __shared__ __attribute__((aligned(16))) float xch[4 * 4 * 64 * 4];     // [channel tile][g][lane][4]: the exchange, 16 KB
    __shared__ unsigned int tally[4][3];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, lj = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < EAE_C) {
        vec[tid] = GDN_IN ? beta_in[tid] : 0.f;
        vec[EAE_C + tid] = IGDN_OUT ? beta_out[tid] : 0.f;
        vec[2 * EAE_C + tid] = map_mean ? map_mean[tid] : 0.f;
        vec[3 * EAE_C + tid] = bin_widths[tid];
    }
    const long row = (long)blockIdx.x * 32 + lj;
    const bool valid = row < rows;
    const long img = row / hw;
    const int pix = (int)(row - img * hw);
    const int cw = 32 * w + 4 * hi;                          // this lane's channels: cw + 8 g + q
    const size_t obase = (size_t)(valid ? row : 0) * EAE_C + cw;
    f32x16 own;                                              // [4 g + q]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = valid ? *reinterpret_cast<const float4*>(x + obase + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        own[4 * g + 0] = q.x; own[4 * g + 1] = q.y; own[4 * g + 2] = q.z; own[4 * g + 3] = q.w;
    }
    float4* slot = reinterpret_cast<float4*>(xch) + lane;    // + (16 t + 4 g... ) * 64: [t][g][lane]
    f32x16 full[4];
#define EAE_Q_EXCHANGE()                                                                                                 \
    {                                                                                                                    \
        const f32x16 sq = squares_for_mfma(own);                                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                    \
            slot[(4 * w + g) * 64] = make_float4(sq[4 * g], sq[4 * g + 1], sq[4 * g + 2], sq[4 * g + 3]);                \
        __syncthreads();                                                                                                 \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                    \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                              \
                const float4 q = slot[(4 * t + g) * 64];                                                                 \
                full[t][4 * g + 0] = q.x; full[t][4 * g + 1] = q.y; full[t][4 * g + 2] = q.z; full[t][4 * g + 3] = q.w;  \
            }                                                                                                            \
    }
    if (GDN_IN) {
        EAE_Q_EXCHANGE()
        f32x16 d[1] = {quarter_denominator<8>(full, gamma_in, w, lane)};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bt = *reinterpret_cast<const float4*>(vec + cw + 8 * g);
            d[0][4 * g + 0] = d[0][4 * g + 0] + bt.x;
            d[0][4 * g + 1] = d[0][4 * g + 1] + bt.y;
            d[0][4 * g + 2] = d[0][4 * g + 2] + bt.z;
            d[0][4 * g + 3] = d[0][4 * g + 3] + bt.w;
        }
        const f32x16 xn[1] = {own};
        gdn_tile<1, false, false>(xn, d, [&](int, int g, float4 y) {
            own[4 * g + 0] = y.x; own[4 * g + 1] = y.y; own[4 * g + 2] = y.z; own[4 * g + 3] = y.w;
        });
    } else {
        __syncthreads();                                     // vec
    }
    // the quantiser on this wave's 32 channels (quantize.hip, statement for statement). Planar symbols: the lanes of a
    // half-wave write neighbouring pixels of one map; offsets are relative to the tile's first image so that 32 bits do
    const long img0 = ((long)blockIdx.x * 32) / hw;
    const bool one_image = __all(!valid || img == img0) != 0;
    const __amdgpu_buffer_rsrc_t sym_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        symbols ? symbols + (size_t)img0 * EAE_C * hw : nullptr, 0, symbols ? 0x7FFFFFFF : 0, 0x00020000);
    const int sym_lane = valid ? (int)((((img - img0) * EAE_C + 4 * hi) * hw + pix) * 2) : -1;     // bytes; -1: no store
    unsigned int bad = 0, not_quantized = 0, altered = 0, flag_bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 m4 = *reinterpret_cast<const float4*>(vec + 2 * EAE_C + cw + 8 * g);
        const float4 b4 = *reinterpret_cast<const float4*>(vec + 3 * EAE_C + cw + 8 * g);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, bw[4] = {b4.x, b4.y, b4.z, b4.w};
        float sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float yv = own[4 * g + q];
            const float centered = yv - mm[q];
            const float rr = round_half_even(centered / bw[q]);
            const float cq = bw[q] * rr;
            const float rs = round_half_even(cq / bw[q]);
            sh[q] = cq + mm[q];
            if (valid) {
                if (!(fabsf(rs) < 32768.f)) bad++;
                if (!(fabs((double)cq - (double)centered) < 1.5e-10)) not_quantized++;
                if (!((float)(int16_t)(int)rs * bw[q] == centered)) altered++;
            }
            const bool nz = valid && cq != 0.f;
            if (nonzero) {
                if (one_image) {
                    // one bit per channel of this tile: a half-wave's 32 positions share the channel
                    const unsigned long long any = __ballot(nz);
                    if ((unsigned int)any) flag_bits |= 1u << (8 * g + q);
                    if ((unsigned int)(any >> 32)) flag_bits |= 1u << (8 * g + 4 + q);
                } else if (nz) {
                    nonzero[img * EAE_C + cw + 8 * g + q] = 1u;                      // benign race: every writer stores 1
                }
            }
            if (symbols)
                __builtin_amdgcn_raw_buffer_store_b16((short)(int16_t)(int)rs, sym_rsrc, sym_lane, (32 * w + 8 * g + q) * hw * 2, 0);
        }
        if (valid && y_out)
            *reinterpret_cast<float4*>(y_out + obase + 8 * g) = make_float4(own[4 * g], own[4 * g + 1], own[4 * g + 2], own[4 * g + 3]);
        if (valid && shifted_out) *reinterpret_cast<float4*>(shifted_out + obase + 8 * g) = make_float4(sh[0], sh[1], sh[2], sh[3]);
        own[4 * g + 0] = sh[0]; own[4 * g + 1] = sh[1]; own[4 * g + 2] = sh[2]; own[4 * g + 3] = sh[3];
    }
    if (nonzero && one_image && lane < 32 && ((flag_bits >> lane) & 1u)) nonzero[img0 * EAE_C + 32 * w + lane] = 1u;
#ifndef EAE_LATENT_NOCHECKS
    if (checks) {
        // one atomic per block and counter: all 4,608 waves adding to the same three words would queue behind each other in L2
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            bad += __shfl_down(bad, off, 64);
            not_quantized += __shfl_down(not_quantized, off, 64);
            altered += __shfl_down(altered, off, 64);
        }
        if (lane == 0) { tally[w][0] = bad; tally[w][1] = not_quantized; tally[w][2] = altered; }
    }
#endif
    if (IGDN_OUT) {
        if (GDN_IN) __syncthreads();                         // every wave has read the first exchange
        EAE_Q_EXCHANGE()
        f32x16 d[1] = {quarter_denominator<8>(full, gamma_out, w, lane)};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bt = *reinterpret_cast<const float4*>(vec + EAE_C + cw + 8 * g);
            d[0][4 * g + 0] = d[0][4 * g + 0] + bt.x;
            d[0][4 * g + 1] = d[0][4 * g + 1] + bt.y;
            d[0][4 * g + 2] = d[0][4 * g + 2] + bt.z;
            d[0][4 * g + 3] = d[0][4 * g + 3] + bt.w;
        }
        const f32x16 xn[1] = {own};
        gdn_tile<1, true, false>(xn, d, [&](int, int g, float4 y) {
            if (valid) *reinterpret_cast<float4*>(t_out + obase + 8 * g) = y;
        });
    }
#ifndef EAE_LATENT_NOCHECKS
    if (checks) {
        if (!IGDN_OUT) __syncthreads();                      // (with IGDN_OUT the second exchange's barrier came after the tally)
        if (tid < 3) {
            const unsigned int sum = tally[0][tid] + tally[1][tid] + tally[2][tid] + tally[3][tid];
            if (sum) atomicAdd(&checks[tid], sum);
        }
    }
#endif
#undef EAE_Q_EXCHANGE
}
}  // namespace

extern "C" int eae_hip_latent_stage(const float* x, const float* gamma_in_packed, const float* beta_in, const float* map_mean,
                                    const float* bin_widths, const float* gamma_out_packed, const float* beta_out, float* y_out,
                                    float* shifted_out, float* t_out, int16_t* symbols_planar, uint32_t* nonzero_flags,
                                    uint32_t* checks, int n, int hw, void* stream) {
    if (!x || !bin_widths || n <= 0 || hw <= 0) return EAE_HIP_BAD_ARGUMENT;
    if ((gamma_in_packed == nullptr) != (beta_in == nullptr)) return EAE_HIP_BAD_ARGUMENT;
    if ((gamma_out_packed == nullptr) != (beta_out == nullptr) || (gamma_out_packed && !t_out)) return EAE_HIP_BAD_ARGUMENT;
    const long rows = (long)n * hw;
    hipStream_t s = (hipStream_t)stream;
    // EAE_HIP_LATENT = q (four waves per tile, the default) | w (one wave per tile) | l (block-cooperative LDS form); read when the
    // library is loaded (misc.hip): the parity tests run all three.
    char form = g_eae_launch_options.latent;
    if (form == 'q' && hw > (1 << 21)) form = 'w';          // 32-bit symbol offsets inside two images
    if (form != 'l') {
        const unsigned wgrid = (unsigned)((rows + 31) / 32);
#define EAE_LATENT_W(A_, B_)                                                                                              \
        {                                                                                                                \
            if (form == 'q')                                                                                             \
                hipLaunchKernelGGL((latent_quarter_kernel<A_, B_>), dim3(wgrid), dim3(256), 0, s, x, gamma_in_packed, beta_in, map_mean,  \
                                   bin_widths, gamma_out_packed, beta_out, y_out, shifted_out, t_out, symbols_planar, nonzero_flags, checks, rows, hw); \
            else                                                                                                         \
                hipLaunchKernelGGL((latent_wave_kernel<A_, B_>), dim3(wgrid), dim3(64), 0, s, x, gamma_in_packed, beta_in, map_mean,  \
                                   bin_widths, gamma_out_packed, beta_out, y_out, shifted_out, t_out, symbols_planar, nonzero_flags, checks, rows, hw); \
        }
        if (gamma_in_packed && gamma_out_packed) EAE_LATENT_W(true, true)
        else if (gamma_in_packed) EAE_LATENT_W(true, false)
        else if (gamma_out_packed) EAE_LATENT_W(false, true)
        else EAE_LATENT_W(false, false)
#undef EAE_LATENT_W
        EAE_HIP_CHECK_LAUNCH();
        return EAE_HIP_OK;
    }
    const unsigned grid = (unsigned)((rows + TM - 1) / TM);
#define EAE_LATENT(A_, B_)                                                                                               \
    hipLaunchKernelGGL((latent_kernel<A_, B_>), dim3(grid), dim3(WAVES * 64), 0, s, x, gamma_in_packed, beta_in, map_mean, bin_widths, \
                       gamma_out_packed, beta_out, y_out, shifted_out, t_out, symbols_planar, nonzero_flags, checks, rows, hw)
    if (gamma_in_packed && gamma_out_packed) EAE_LATENT(true, true);
    else if (gamma_in_packed) EAE_LATENT(true, false);
    else if (gamma_out_packed) EAE_LATENT(false, true);
    else EAE_LATENT(false, false);
#undef EAE_LATENT
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
