// conv_gemm.hip -- the four dense 128->128 convolutions of the path as ONE implicit-GEMM kernel on f32 MFMA:
//   conv_2, conv_3            tf.nn.conv2d 5x5 s2 SAME            (eae/graph/components.py:126-136)
//   transpose_conv_1, _2      tf.nn.conv2d_transpose 5x5 s2 SAME  (components.py:63-75)
// with the bias_add and the GDN / IGDN that follows each of them fused into the epilogue (components.py:130-142,
// 68-78; tfutils.py:393-397, 505-509).
//
// Formulation. A "position" is an output pixel (conv) or an input-grid site (transposed conv, one of the 4 output
// phases (I mod 2, J mod 2)). For one tile of TM positions the kernel computes
//     Y[TM][128] = sum over taps t, input channels ci of  X[pos + offset(t)][ci] * Wp[t][ci][:]
// as a GEMM with K = taps * 128: per K-step a [TM x 32] slab of activations (gathered rows, zero outside the image)
// and a [32 x 128] slab of weights are staged in LDS and consumed by v_mfma_f32_32x32x2_f32.
// Accumulation order per output element: taps in table order, ci ascending, ONE accumulator -- exactly the f32 FMA
// chain oracle/transforms_oracle.c runs, so results are bit-identical to the CPU oracle.
//
// Roofline: MFMA-bound (f32 MFMA 157.3 TFLOP/s). Per tile of 128 positions and 25 taps: 105 MFLOP against 1.6 MB of
// weights (L2-resident, shared by every block) and ~0.34 MB of activations.
#include "common.h"

#include <cstdlib>

namespace {

constexpr int KC = 32;           // K-step (input channels per LDS slab)
constexpr int AS_STRIDE = 33;    // +1 float: 32 rows read the same k -> conflict-free ds_read_b32
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
    const float* w;      // [T][128 ci][128 co]
    const float* bias;   // [128] or nullptr
    const float* gamma;  // [128][128] (k, c) or nullptr
    const float* beta;   // [128]
    int norm;            // EAE_NORM_*
    int n, hin, win, hp, wp, hout, wout;
    int in_stride, out_stride;
    int tiles_r, tiles_c, n_phases;
    PhaseDesc phase[4];
};

template <int TM>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmParams p) {
    constexpr int TILE_W = TM / TILE_H;
    constexpr int WAVES_M = TM / 32;       // waves along positions
    constexpr int WAVES_N = 4 / WAVES_M;   // waves along output channels
    constexpr int NT = 4 / WAVES_N;        // 32-wide channel tiles per wave
    constexpr int A_PASSES = TM / 32;      // float4 loads per thread per K-step for the activation slab
    constexpr int LDS_MAIN = TM * AS_STRIDE + KC * EAE_C;
    constexpr int LDS_EPI = TM * XS_STRIDE;
    constexpr int LDS_FLOATS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* As = lds;                      // [TM][33]
    float* Bs = lds + TM * AS_STRIDE;     // [32][128]   (TM*33*4 bytes is a multiple of 16 for TM in {64,128})

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WAVES_M;
    const int wn = wave / WAVES_M;

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ph = b % p.n_phases; b /= p.n_phases;
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tr = b % p.tiles_r;
    const int img = b / p.tiles_r;
    const PhaseDesc& pd = p.phase[ph];

    // activation-slab loader: thread -> (position a_m[i], channel quad a_q)
    const int a_q = tid & 7;
    int a_pr[A_PASSES], a_pc[A_PASSES];
    bool a_ok[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int m = (tid >> 3) + 32 * i;
        a_pr[i] = tr * TILE_H + m / TILE_W;
        a_pc[i] = tc * TILE_W + m % TILE_W;
        a_ok[i] = a_pr[i] < p.hp && a_pc[i] < p.wp;
    }
    const float* in_img = p.in + (size_t)img * p.hin * p.win * EAE_C;

    // Staging registers for the next K-step (plain named registers + macros: lambdas capturing these arrays by
    // reference made hipcc demote them to scratch).
    float4 a_reg[A_PASSES];
    float4 b_reg0, b_reg1, b_reg2, b_reg3;
#define EAE_PREFETCH(step_)                                                                                          \
    {                                                                                                                \
        const int packed_ = pd.tap[(step_) >> 2];                                                                    \
        const int dr_ = (packed_ & 0xFF) - 8, dc_ = ((packed_ >> 8) & 0xFF) - 8, widx_ = packed_ >> 16;              \
        const int ci0_ = ((step_) & 3) * KC;                                                                         \
        _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
            const int r_ = a_pr[i] * p.in_stride + dr_;                                                              \
            const int c_ = a_pc[i] * p.in_stride + dc_;                                                              \
            const bool ok_ = a_ok[i] && r_ >= 0 && r_ < p.hin && c_ >= 0 && c_ < p.win;                              \
            const float4* src_ = reinterpret_cast<const float4*>(                                                    \
                in_img + ((size_t)(ok_ ? r_ : 0) * p.win + (ok_ ? c_ : 0)) * EAE_C + ci0_ + 4 * a_q);                \
            const float4 v_ = *src_;                                                                                 \
            a_reg[i] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
        }                                                                                                            \
        const float4* wslab_ = reinterpret_cast<const float4*>(p.w + ((size_t)widx_ * EAE_C + ci0_) * EAE_C) + tid; \
        b_reg0 = wslab_[0]; b_reg1 = wslab_[256]; b_reg2 = wslab_[512]; b_reg3 = wslab_[768];                        \
    }
#define EAE_STAGE()                                                                                                  \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
            float* dst_ = As + ((tid >> 3) + 32 * i) * AS_STRIDE + 4 * a_q;                                          \
            dst_[0] = a_reg[i].x; dst_[1] = a_reg[i].y; dst_[2] = a_reg[i].z; dst_[3] = a_reg[i].w;                  \
        }                                                                                                            \
        float4* bdst_ = reinterpret_cast<float4*>(Bs) + tid;                                                         \
        bdst_[0] = b_reg0; bdst_[256] = b_reg1; bdst_[512] = b_reg2; bdst_[768] = b_reg3;                            \
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = pd.ntaps * (EAE_C / KC);
    EAE_PREFETCH(0)
    const float* a_rd = As + (wm * 32 + (lane & 31)) * AS_STRIDE + (lane >> 5);
    const float* b_rd = Bs + (lane >> 5) * EAE_C + wn * NT * 32 + (lane & 31);
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();           // previous step's LDS reads are done
        EAE_STAGE()
        __syncthreads();
        {   // next slab's global loads stay in flight during the MFMAs below (the last step re-loads itself: harmless,
            // and keeps the loads unconditional so the staging registers are not demoted to scratch)
            const int nxt = step + 1 < nsteps ? step + 1 : step;
            EAE_PREFETCH(nxt)
        }
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const float a = a_rd[2 * kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(a, b_rd[2 * kk * EAE_C + t * 32], acc[t]);
        }
    }

    // ---- epilogue: bias_add, then (I)GDN over the 128 channels of each position ------------------------------------
    const int col0 = wn * NT * 32 + (lane & 31);
    if (p.bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bv = p.bias[col0 + t * 32];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] + bv;
        }
    }
    float* out_img = p.out + (size_t)img * p.hout * p.wout * EAE_C;
    if (p.norm == EAE_NORM_NONE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            const int pr = tr * TILE_H + m / TILE_W, pc = tc * TILE_W + m % TILE_W;
            if (pr < p.hp && pc < p.wp) {
                float* o = out_img + ((size_t)(pr * p.out_stride + pd.out_a) * p.wout + (pc * p.out_stride + pd.out_b)) * EAE_C;
#pragma unroll
                for (int t = 0; t < NT; ++t) o[col0 + t * 32] = acc[t][r];
            }
        }
        return;
    }
    __syncthreads();               // everyone is done with As/Bs
    float* Xs = lds;               // [TM][129]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(wm * 32 + acc_row32(r, lane)) * XS_STRIDE + col0 + t * 32] = acc[t][r];
    __syncthreads();
    f32x16 d[NT];
    gdn_denominator<NT>(Xs, wm, lane, p.gamma, col0, d);
    const bool inverse = p.norm == EAE_NORM_IGDN;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bt = p.beta[col0 + t * 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            const int pr = tr * TILE_H + m / TILE_W, pc = tc * TILE_W + m % TILE_W;
            const float y = gdn_apply(acc[t][r], d[t][r], bt, inverse);
            if (pr < p.hp && pc < p.wp)
                out_img[((size_t)(pr * p.out_stride + pd.out_a) * p.wout + (pc * p.out_stride + pd.out_b)) * EAE_C + col0 + t * 32] = y;
        }
    }
}

int launch(ConvGemmParams& p, hipStream_t stream) {
    // 128-position tiles when they fill the chip twice over, else 64-position tiles (more, smaller blocks).
    const long positions = (long)p.n * p.hp * p.wp * p.n_phases;
    bool big = positions >= 128L * 512 && p.wp >= 16;
    if (const char* force = std::getenv("EAE_HIP_FORCE_TILE")) big = std::atoi(force) == 128;   // tests cover both tiles
    const int tile_w = big ? 16 : 8;
    p.tiles_r = (p.hp + TILE_H - 1) / TILE_H;
    p.tiles_c = (p.wp + tile_w - 1) / tile_w;
    const int grid = p.n * p.tiles_r * p.tiles_c * p.n_phases;
    if (big) hipLaunchKernelGGL(conv_gemm_kernel<128>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv_gemm_kernel<64>, dim3(grid), dim3(256), 0, stream, p);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

}  // namespace

extern "C" int eae_hip_conv5x5s2(const float* x, const float* w, const float* bias, int norm, const float* gamma,
                                 const float* beta, float* out, int n, int h, int w_in, void* stream) {
    if (!x || !w || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (norm != EAE_NORM_NONE && (!gamma || !beta)) return EAE_HIP_BAD_ARGUMENT;
    if ((h & 1) || (w_in & 1)) return EAE_HIP_BAD_SHAPE;
    ConvGemmParams p{};
    p.in = x; p.out = out; p.w = w; p.bias = bias; p.gamma = gamma; p.beta = beta; p.norm = norm;
    p.n = n; p.hin = h; p.win = w_in; p.hp = h / 2; p.wp = w_in / 2; p.hout = h / 2; p.wout = w_in / 2;
    p.in_stride = 2; p.out_stride = 1; p.n_phases = 1;
    PhaseDesc& pd = p.phase[0];
    pd.out_a = 0; pd.out_b = 0; pd.ntaps = 25;
    for (int u = 0; u < 5; ++u)
        for (int v = 0; v < 5; ++v) {      // SAME padding for k5 s2 on even sizes: 1 before, 2 after (appendix A.2)
            pd.tap[u * 5 + v] = pack_tap(u - 1, v - 1, u * 5 + v);
        }
    return launch(p, (hipStream_t)stream);
}

extern "C" int eae_hip_tconv5x5s2(const float* x, const float* w_packed, const float* bias, int norm,
                                  const float* gamma, const float* beta, float* out, int n, int h, int w_in,
                                  void* stream) {
    if (!x || !w_packed || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (norm != EAE_NORM_NONE && (!gamma || !beta)) return EAE_HIP_BAD_ARGUMENT;
    ConvGemmParams p{};
    p.in = x; p.out = out; p.w = w_packed; p.bias = bias; p.gamma = gamma; p.beta = beta; p.norm = norm;
    p.n = n; p.hin = h; p.win = w_in; p.hp = h; p.wp = w_in; p.hout = 2 * h; p.wout = 2 * w_in;
    p.in_stride = 1; p.out_stride = 2; p.n_phases = 4;
    // Output pixel I = 2p' + a receives tap u from input row p = p' - (u - a - 1)/2 (appendix A.3, pad_before 1):
    //   a = 0: u = 1 (p'), 3 (p'-1);   a = 1: u = 0 (p'+1), 2 (p'), 4 (p'-1).   Taps in ascending (u, v).
    for (int a = 0; a < 2; ++a)
        for (int bq = 0; bq < 2; ++bq) {
            PhaseDesc& pd = p.phase[a * 2 + bq];
            pd.out_a = a; pd.out_b = bq; pd.ntaps = 0;
            for (int u = 0; u < 5; ++u) {
                if (((u - a - 1) & 1) != 0) continue;
                for (int v = 0; v < 5; ++v) {
                    if (((v - bq - 1) & 1) != 0) continue;
                    pd.tap[pd.ntaps++] = pack_tap(-(u - a - 1) / 2, -(v - bq - 1) / 2, u * 5 + v);
                }
            }
        }
    return launch(p, (hipStream_t)stream);
}

namespace {
__global__ void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int taps, int a, int b) {
    // src [taps][a][b] -> dst [taps][b][a]
    const long total = (long)taps * a * b;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / (a * b));
        const int rem = (int)(i % (a * b));
        const int bi = rem / a, ai = rem % a;   // i indexes dst
        dst[i] = src[((long)t * a + ai) * b + bi];
    }
}
}  // namespace

extern "C" int eae_hip_pack_tconv_weights(const float* w_tf, float* w_packed, int taps, int c_out, int c_in, void* stream) {
    if (!w_tf || !w_packed || taps <= 0 || c_out <= 0 || c_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, taps, c_out, c_in);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
