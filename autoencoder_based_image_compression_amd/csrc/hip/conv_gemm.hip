// conv_gemm.hip -- the four dense 128->128 convolutions of the path as ONE implicit-GEMM kernel on f32 MFMA:
//   conv_2, conv_3            tf.nn.conv2d 5x5 s2 SAME            (eae/graph/components.py:126-136)
//   transpose_conv_1, _2      tf.nn.conv2d_transpose 5x5 s2 SAME  (components.py:63-75)
// with the bias_add and the GDN / IGDN that follows each of them fused into the epilogue (components.py:130-142,
// 68-78; tfutils.py:393-397, 505-509).
//
// Formulation. A "position" is an output pixel (conv) or an input-grid site (transposed conv, one of the 4 output
// phases (I mod 2, J mod 2)). For one tile of TM positions the kernel computes
//     Y[TM][128] = sum over taps t, input channels ci of  X[pos + offset(t)][ci] * Wp[t][ci][:]
// as a GEMM with K = taps * 128. Per K-step (32 input channels of one tap) a [TM x 32] slab of activations (gathered
// rows, zero outside the image) and a [32 x 128] slab of weights are staged in LDS and consumed by
// v_mfma_f32_32x32x2_f32. Accumulation order per output element: 32-channel block (outer), taps in table order, channel
// within the block, ONE accumulator -- exactly the f32 FMA chain oracle/transforms_oracle.c runs, so results are
// bit-identical to the CPU oracle. The channel block is the OUTER loop on purpose: the 25 taps of a block then re-read
// the same 128-byte slice of each input pixel, which keeps the per-XCD reuse distance (~1.5 MB) inside the 4 MB L2;
// tap-outer order re-fetched the input ~6x from beyond L2 (profiles/traffic_conv_gemm.json).
//
// Data movement (what makes it MFMA-bound rather than LDS/issue-bound):
//  * weights arrive pre-packed (eae_hip_pack_*): output channels permuted so that the 4 values a lane needs for its 4
//    column tiles are contiguous -> the weight slab is a straight 16-byte copy global->LDS and ONE ds_read_b128 per k;
//  * the activation slab is stored [row][k parity][k/2] (+4 floats pad): the 16 k-values a lane needs in a K-step are
//    contiguous -> 4 ds_read_b128, conflict-free (row stride 144 B = 9 x 16 B);
//  * LDS is double-buffered: the next slab is fetched global->registers during the MFMAs, written to the other buffer
//    after them, ONE barrier per K-step; two blocks per CU overlap each other's staging.
// Roofline: f32 MFMA 157.3 TFLOP/s. Per 128-position tile and 25 taps: 105 MFLOP against 1.6 MB of weights
// (L2-resident, shared by every block) and ~0.34 MB of activations.
#include "conv_gemm.h"

#include <cstdlib>

using namespace eae_conv_gemm;

namespace {

unsigned long long* g_stamp_buffer = nullptr;

template <int TM>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmParams p) {
    constexpr int TILE_W = TM / TILE_H;
    constexpr int WAVES_M = TM / 32;       // waves along positions
    constexpr int WAVES_N = 4 / WAVES_M;   // waves along output channels
    constexpr int NT = 4 / WAVES_N;        // 32-wide channel tiles per wave
    constexpr int A_PASSES = TM / 32;      // float4 loads per thread per K-step for the activation slab
    constexpr int BUF = TM * AS_STRIDE + KC * EAE_C;   // floats per LDS buffer
    constexpr int LDS_MAIN = 2 * BUF;
    constexpr int LDS_EPI = TM * XS_STRIDE;
    constexpr int LDS_FLOATS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

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

    // activation-slab loader: thread -> (position (tid>>3) + 32 i, channel quad a_q)
    const int a_q = tid & 7;
    int a_pr[A_PASSES], a_pc[A_PASSES];
    int a_ok[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int m = (tid >> 3) + 32 * i;
        a_pr[i] = tr * TILE_H + m / TILE_W;
        a_pc[i] = tc * TILE_W + m % TILE_W;
        a_ok[i] = (a_pr[i] < p.hp) & (a_pc[i] < p.wp);
    }
    const float* in_img = p.in + (size_t)img * p.hin * p.win * EAE_C;

    // Staging registers for the next K-step (plain named registers + macros: lambdas capturing these arrays by
    // reference made hipcc demote them to scratch). Activation rows are fetched with BUFFER loads whose descriptor
    // covers exactly this image: positions outside the image (SAME zero padding, ragged tiles) get an offset beyond
    // the descriptor and the hardware returns 0 -- no branches, no selects.
    float4 a_reg[A_PASSES];
    float4 b_reg0, b_reg1, b_reg2, b_reg3;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in_img), 0, (int)((size_t)p.hin * p.win * EAE_C * sizeof(float)), 0x00020000);
    const int w_lane_off = tid * 16;
#define EAE_PREFETCH(step_)                                                                                          \
    {                                                                                                                \
        const int packed_ = pd.tap[(step_) % pd.ntaps];   /* K order: 32-channel chunk (outer), then tap */          \
        const int dr_ = (packed_ & 0xFF) - 8, dc_ = ((packed_ >> 8) & 0xFF) - 8, widx_ = packed_ >> 16;              \
        const int ci0_ = ((step_) / pd.ntaps) * KC;                                                                  \
        _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
            const int r_ = a_pr[i] * p.in_stride + dr_;                                                              \
            const int c_ = a_pc[i] * p.in_stride + dc_;                                                              \
            const int ok_ = a_ok[i] & ((unsigned)r_ < (unsigned)p.hin) & ((unsigned)c_ < (unsigned)p.win);           \
            const int lin_ = ((r_ * p.win + c_) * EAE_C + ci0_ + 4 * a_q) * 4;                                       \
            const int off_ = ok_ ? lin_ : -1;   /* beyond num_records -> the load returns 0 */                      \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off_, 0, 0);                             \
            a_reg[i] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),              \
                                   __uint_as_float(v_.w));                                                           \
        }                                                                                                            \
        const float4* wslab_ = reinterpret_cast<const float4*>(                                                      \
            reinterpret_cast<const char*>(p.w + ((size_t)widx_ * EAE_C + ci0_) * EAE_C) + w_lane_off);               \
        b_reg0 = wslab_[0]; b_reg1 = wslab_[256]; b_reg2 = wslab_[512]; b_reg3 = wslab_[768];                        \
    }
    // activation row layout: [parity of k][k >> 1]; channels 4q..4q+3 = k 4q (even), 4q+1 (odd), 4q+2 (even), 4q+3 (odd)
#define EAE_STAGE(buf_)                                                                                              \
    {                                                                                                                \
        float* As_ = lds + (buf_) * BUF;                                                                             \
        _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                       \
            float* dst_ = As_ + ((tid >> 3) + 32 * i) * AS_STRIDE + 2 * a_q;                                         \
            *reinterpret_cast<float2*>(dst_) = make_float2(a_reg[i].x, a_reg[i].z);                                  \
            *reinterpret_cast<float2*>(dst_ + 16) = make_float2(a_reg[i].y, a_reg[i].w);                             \
        }                                                                                                            \
        float4* bdst_ = reinterpret_cast<float4*>(As_ + TM * AS_STRIDE) + tid;                                       \
        bdst_[0] = b_reg0; bdst_[256] = b_reg1; bdst_[512] = b_reg2; bdst_[768] = b_reg3;                            \
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = pd.ntaps * (EAE_C / KC);
    const int hi = lane >> 5, lj = lane & 31;
    const int a_off = (wm * 32 + lj) * AS_STRIDE + hi * 16;
    const int b_off = TM * AS_STRIDE + hi * EAE_C + lj * 4 + wn * NT;

    EAE_PREFETCH(0)
    EAE_STAGE(0)
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        {   // next slab: global -> registers, in flight during the MFMAs below (the last step re-loads itself: harmless,
            // and keeps the loads unconditional so the staging registers are not demoted to scratch)
            const int nxt = step + 1 < nsteps ? step + 1 : step;
            EAE_PREFETCH(nxt)
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the global loads up here: hipcc otherwise sinks them below the MFMAs
        const float* base = lds + (step & 1) * BUF;
        const float4* a_rd = reinterpret_cast<const float4*>(base + a_off);
        const float4 a0 = a_rd[0], a1 = a_rd[1], a2 = a_rd[2], a3 = a_rd[3];
        const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
        const float* b_rd = base + b_off;
        // weight fragments: register double buffer, the read for k+1 is issued before the MFMAs of k
        float bv[2][NT];
#define EAE_LOAD_B(dst_, kk_)                                                                                        \
        if constexpr (NT == 4) {                                                                                     \
            const float4 v_ = *reinterpret_cast<const float4*>(b_rd + 2 * (kk_) * EAE_C);                            \
            dst_[0] = v_.x; dst_[1] = v_.y; dst_[2] = v_.z; dst_[3] = v_.w;                                          \
        } else {                                                                                                     \
            const float2 v_ = *reinterpret_cast<const float2*>(b_rd + 2 * (kk_) * EAE_C);                            \
            dst_[0] = v_.x; dst_[1] = v_.y;                                                                          \
        }
        // software pipeline, pinned with sched_group_barrier: reads run TWO k ahead of the MFMAs that consume them,
        //   [4 A reads, B(0), B(1)]  { [NT MFMAs of k]  [B(k+2)] } x 16
        // so each read has NT*64 cycles (one MFMA group) to land before the wave needs it.
        EAE_LOAD_B(bv[0], 0)
        EAE_LOAD_B(bv[1], 1)
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(av[kk], bv[kk & 1][t], acc[t]);
            if (kk + 2 < KC / 2) { EAE_LOAD_B(bv[kk & 1], kk + 2) }
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the other buffer was last read in step-1, and every wave passed the barrier that ended step-1
        EAE_STAGE((step + 1) & 1)
        __syncthreads();
    }

    // ---- epilogue: bias_add, then (I)GDN over the 128 channels of each position ------------------------------------
    const int t0 = wn * NT;
    if (p.bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bv = p.bias[(t0 + t) * 32 + lj];
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
                for (int t = 0; t < NT; ++t) o[(t0 + t) * 32 + lj] = acc[t][r];
            }
        }
        return;
    }
    // (the loop's last barrier guarantees every wave is done with both buffers)
    float* Xs = lds;               // [TM][129]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(wm * 32 + acc_row32(r, lane)) * XS_STRIDE + (t0 + t) * 32 + lj] = acc[t][r];
    __syncthreads();
    f32x16 d[NT];
    gdn_denominator<NT>(Xs, wm, lane, p.gamma, t0, d);
    const bool inverse = p.norm == EAE_NORM_IGDN;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bt = p.beta[(t0 + t) * 32 + lj];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            const int pr = tr * TILE_H + m / TILE_W, pc = tc * TILE_W + m % TILE_W;
            const float y = gdn_apply(acc[t][r], d[t][r], bt, inverse);
            if (pr < p.hp && pc < p.wp)
                out_img[((size_t)(pr * p.out_stride + pd.out_a) * p.wout + (pc * p.out_stride + pd.out_b)) * EAE_C + (t0 + t) * 32 + lj] = y;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Barrier-free, transposed variant (the default). Every wave owns 32 positions x 128 output channels and runs its OWN
// software pipeline; the product is computed TRANSPOSED, Y^T[co][pos] = sum_k W^T[co][k] * X^T[k][pos]:
//   * MFMA A operand = weights: each lane loads the 16 bytes it needs for (k, its 4 channel tiles) straight from
//     global memory into a register ring that runs RING k-pairs (RING * 256 MFMA cycles) ahead; the waves of a CU read
//     the same 16 KB slab within microseconds of each other, so these are L1 / L2 hits;
//   * MFMA B operand = activations: 4 buffer loads per K-step -> wave-private LDS double buffer ([parity][k/2] rows)
//     -> 4 ds_read_b128; only the wave itself touches that LDS region, so there is NO workgroup barrier anywhere;
//   * the accumulator then has the POSITION on the lane and the channel in the register index, which is exactly the B
//     operand layout of the GDN product d^T[c][pos] = sum_k gamma[k][c] * x^2[pos][k]: one v_permlane32_swap per register
//     pair turns (k, k+4 | k+1, k+5) into the natural pairs (k, k+1), (k+4, k+5), so x^2 goes from the accumulator
//     registers into the MFMA in ascending k with no LDS round trip; gamma streams through the same register ring;
//   * each lane ends up with 4 consecutive channels per register quad -> 16-byte output stores.
// Same per-element FMA chain (k ascending) as the cooperative kernel and the CPU oracle -> same bits.
// NT = 32-channel output tiles per wave: 4 (all 128 channels, epilogue fused) or, for layers too small to fill the
// machine with 32-position x 128-channel waves (conv_3 at Kodak batch sizes: 1152 waves on 1024 SIMDs ran as two rounds),
// 2 or 1: the channel tiles of one position tile are then spread over 4 / NT blocks of the same XCD, the epilogue is the
// bias only (NORM must be NONE) and launch() runs the GDN as its own pass over the output. Same per-element FMA chain.
// PACK (NT < 4, four waves): the block holds the 4 / NT channel parts of NT position tiles -- wave w = part w % PARTS of position
// tile w / PARTS -- instead of one wave. The waves are as independent as before; what changes is where they land: the dispatcher
// gives the four waves of a workgroup one SIMD each, while one-wave blocks arriving at a CU behind another kernel's last waves
// were seen two to a SIMD (conv_3 of 64 x 256x256 in the step: 96 SIMDs with two waves, 96 with none, 0.22 ms against 0.15).
template <int WAVES, int NORM, int NT = 4, bool PACK = false>
__global__ __launch_bounds__(WAVES * 64, 2) void conv_gemm_wave_kernel(const ConvGemmParams p) {
    static_assert(NT == 4 || NORM == EAE_NORM_NONE, "partial channel tiles carry no normalisation");
    static_assert(!PACK || (WAVES == 4 && NT < 4), "the packed form is four waves of partial channel tiles");
    constexpr int PARTS = 4 / NT;
    constexpr int GRID_PARTS = PACK ? 1 : PARTS;               // channel parts that are blocks of their own
    constexpr int TM = (PACK ? WAVES / PARTS : WAVES) * 32;    // positions of a block's tile
    constexpr int TILE_W = TM / TILE_H;
    constexpr int RING = 8;        // k-pairs of weights in flight (16 for the two-tile form measured slower: 0.136 against 0.128 ms, conv_3 of 64 x 256x256)
    constexpr int ABUF = 32 * AS_STRIDE;                       // one activation buffer of one wave
    constexpr int WAVE_LDS = 2 * ABUF + 2 * EAE_C;             // + bias[128] + beta[128]
    __shared__ __attribute__((aligned(16))) float lds[WAVES * WAVE_LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* wlds = lds + wave * WAVE_LDS;
    float* vec_lds = wlds + 2 * ABUF;                          // per-channel vectors for the epilogue
    {   // bias / beta land in LDS long before the epilogue needs them (16 global loads there were 16 serial L2 trips)
        const float2 bz = p.bias ? *reinterpret_cast<const float2*>(p.bias + 2 * lane) : make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(vec_lds + 2 * lane) = bz;
        if (NORM != EAE_NORM_NONE) *reinterpret_cast<float2*>(vec_lds + EAE_C + 2 * lane) = *reinterpret_cast<const float2*>(p.beta + 2 * lane);
    }

    // Block order: PHASE-MAJOR, longest phase first (launch() sorts p.phase by taps). The hardware hands workgroups to
    // CUs in order, round-robin, and does not run ahead of a full CU: with the phase as the fastest index every CU
    // kept receiving the SAME phase and the 4-tap CUs idled behind the 9-tap ones (measured: 70 % MFMA utilisation).
    // Phase-major, every CU works through the same mix of phases.
    // Blocks b and b + 8 share an XCD (round-robin over the 8 XCDs): XCD x = b % 8 owns a contiguous range of tiles_x
    // tiles (shared halos stay in its L2) and walks them once per phase, longest phase first. The grid is padded to
    // 8 * tiles_x tiles per phase; the (< 8 per phase) surplus blocks exit at once.
    const int tiles_x = (int)gridDim.x / (8 * p.n_phases * GRID_PARTS);
    const int part = PACK ? wave % PARTS : ((int)blockIdx.x >> 3) % PARTS;      // which NT channel tiles (neighbouring blocks of one XCD, or waves of the block)
    const int seq = ((int)blockIdx.x >> 3) / GRID_PARTS;
    const int sub = PACK ? wave / PARTS : wave;                // which 32 positions of the block's tile
    const int ph = seq / tiles_x;
    int b = ((int)blockIdx.x & 7) * tiles_x + seq % tiles_x;
    if (b >= p.n * p.tiles_r * p.tiles_c) return;
    const int tc = b % p.tiles_c; b /= p.tiles_c;
    const int tr = b % p.tiles_r;
    const int img = b / p.tiles_r;
    const PhaseDesc& pd = p.phase[ph];

    // activation loader: lane -> (row (lane>>3) + 8 i of this wave, channel quad lane & 7)
    const int a_q = lane & 7;
    int a_pr[4], a_pc[4], a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = sub * 32 + (lane >> 3) + 8 * i;
        a_pr[i] = tr * TILE_H + m / TILE_W;
        a_pc[i] = tc * TILE_W + m % TILE_W;
        a_ok[i] = (a_pr[i] < p.hp) & (a_pc[i] < p.wp);
    }
    const float* in_img = p.in + (size_t)img * p.hin * p.win * EAE_C;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(in_img), 0, (int)((size_t)p.hin * p.win * EAE_C * sizeof(float)), 0x00020000);
    const int hi = lane >> 5, lj = lane & 31;
    // weights / gamma: lane reads 16 bytes at row (k = 2 kk + hi), packed column lj*4
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)MAX_TAPS * EAE_C * EAE_C * sizeof(float)), 0x00020000);
    const int w_lane = (hi * EAE_C + lj * 4 + part * NT) * 4;   // byte offset inside a k-pair (+ this block's channel tiles)

    float4 a_reg[4];
#define EAE_W_PREFETCH_A(step_)                                                                                      \
    {                                                                                                                \
        const int packed_ = pd.tap[(step_) % pd.ntaps];   /* K order: 32-channel chunk (outer), then tap */          \
        const int dr_ = (packed_ & 0xFF) - 8, dc_ = ((packed_ >> 8) & 0xFF) - 8;                                     \
        const int ci0_ = ((step_) / pd.ntaps) * KC;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
            const int r_ = a_pr[i] * p.in_stride + dr_;                                                              \
            const int c_ = a_pc[i] * p.in_stride + dc_;                                                              \
            const int ok_ = a_ok[i] & ((unsigned)r_ < (unsigned)p.hin) & ((unsigned)c_ < (unsigned)p.win);           \
            const int lin_ = ((r_ * p.win + c_) * EAE_C + ci0_ + 4 * a_q) * 4;                                       \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok_ ? lin_ : -1, 0, 0);                  \
            a_reg[i] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),              \
                                   __uint_as_float(v_.w));                                                           \
        }                                                                                                            \
    }
#define EAE_W_STAGE_A(buf_)                                                                                          \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
            float* dst_ = wlds + (buf_) * ABUF + ((lane >> 3) + 8 * i) * AS_STRIDE + 2 * a_q;                        \
            *reinterpret_cast<float2*>(dst_) = make_float2(a_reg[i].x, a_reg[i].z);                                  \
            *reinterpret_cast<float2*>(dst_ + 16) = make_float2(a_reg[i].y, a_reg[i].w);                             \
        }                                                                                                            \
    }
    // byte offset of the weight slab of K-step `step_` (tap, 32-channel chunk)
#define EAE_W_SLAB(step_) ((((pd.tap[(step_) % pd.ntaps] >> 16) * EAE_C + ((step_) / pd.ntaps) * KC) * EAE_C) * 4)
#define EAE_W_LOAD(dst_, rsrc_, slab_, kk_)                                                                          \
    {                                                                                                                \
        if constexpr (NT == 4) {                                                                                     \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_, w_lane + (kk_) * 2 * EAE_C * 4, (slab_), 0); \
            dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),                  \
                               __uint_as_float(v_.w));                                                               \
        } else if constexpr (NT == 2) {                                                                              \
            const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(rsrc_, w_lane + (kk_) * 2 * EAE_C * 4, (slab_), 0);  \
            dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), 0.f, 0.f);                              \
        } else {                                                                                                     \
            const unsigned v_ = __builtin_amdgcn_raw_buffer_load_b32(rsrc_, w_lane + (kk_) * 2 * EAE_C * 4, (slab_), 0); \
            dst_ = make_float4(__uint_as_float(v_), 0.f, 0.f, 0.f);                                                  \
        }                                                                                                            \
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = pd.ntaps * (EAE_C / KC);
    const int a_off = lj * AS_STRIDE + hi * 16;

    unsigned long long t_start = 0, t_loop = 0, t_loop_end = 0, t_gdn_end = 0;
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
    float4 ring[RING];
    {
        const int slab0 = EAE_W_SLAB(0);
#pragma unroll
        for (int i = 0; i < RING; ++i) EAE_W_LOAD(ring[i], w_rsrc, slab0, i)
    }
    EAE_W_PREFETCH_A(0)
    EAE_W_STAGE_A(0)
    if (p.stamps) t_loop = __builtin_amdgcn_s_memtime();
    for (int step = 0; step < nsteps; ++step) {
        const int nxt = step + 1 < nsteps ? step + 1 : step;
        const int slab_cur = EAE_W_SLAB(step);
        const int slab_nxt = EAE_W_SLAB(nxt);
        EAE_W_PREFETCH_A(nxt)
        const float4* a_rd = reinterpret_cast<const float4*>(wlds + (step & 1) * ABUF + a_off);
        const float4 a0 = a_rd[0], a1 = a_rd[1], a2 = a_rd[2], a3 = a_rd[3];
        const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const float4 wq = ring[kk % RING];
            acc[0] = mfma32(wq.x, av[kk], acc[0]);      // A = W^T[co = 32 t + lj][k], B = X^T[k][pos = lj]
            if constexpr (NT >= 2) acc[1] = mfma32(wq.y, av[kk], acc[1]);
            if constexpr (NT == 4) {
                acc[2] = mfma32(wq.z, av[kk], acc[2]);
                acc[3] = mfma32(wq.w, av[kk], acc[3]);
            }
            // refill this ring slot with the k-pair RING ahead in the K stream (next step's slab once kk + RING >= 16)
            if (kk + RING < KC / 2) { EAE_W_LOAD(ring[kk % RING], w_rsrc, slab_cur, kk + RING) }
            else { EAE_W_LOAD(ring[kk % RING], w_rsrc, slab_nxt, kk + RING - KC / 2) }
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        EAE_W_STAGE_A((step + 1) & 1)
    }

    if (p.stamps) t_loop_end = __builtin_amdgcn_s_memtime();
    // ---- epilogue, all in registers (common.h: wave_epilogue) ---------------------------------------------------------
    const int m = sub * 32 + lj;
    const int pr = tr * TILE_H + m / TILE_W, pc = tc * TILE_W + m % TILE_W;
    const bool valid = pr < p.hp && pc < p.wp;
    float* o = p.out + (size_t)img * p.hout * p.wout * EAE_C +
               ((size_t)(pr * p.out_stride + pd.out_a) * p.wout + (pc * p.out_stride + pd.out_b)) * EAE_C;
    if constexpr (NT == 4) {
        wave_epilogue<NORM>(acc, vec_lds, p.bias != nullptr, p.gamma, o, valid, lane);
    } else if (valid) {
        // channel of (tile t, group g, q) = 32 t + 8 g + 4 hi + q (common.h wave_epilogue); bias only
        float* oo = o + 4 * hi + 32 * NT * part;
        const float* bias_lds = vec_lds + 4 * hi + 32 * NT * part;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = p.bias ? *reinterpret_cast<const float4*>(bias_lds + 32 * t + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 y = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                if (p.bias) y = make_float4(y.x + bv.x, y.y + bv.y, y.z + bv.z, y.w + bv.w);
                *reinterpret_cast<float4*>(oo + 32 * t + 8 * g) = y;
            }
    }
    if (p.stamps) t_gdn_end = __builtin_amdgcn_s_memtime();
    if (p.stamps && lane == 0) {
        unsigned long long* st = p.stamps + ((size_t)blockIdx.x * WAVES + wave) * 8;
        st[0] = t_start; st[1] = t_loop; st[2] = t_loop_end; st[3] = t_gdn_end; st[4] = __builtin_amdgcn_s_memtime();
        st[5] = (unsigned long long)nsteps;
        st[6] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        st[7] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    }
}

// conv_gemm_split.hip when EAE_HIP_GEMM asks for it or the layer has at least one 32-position tile per SIMD; 1 = not taken
int try_split(ConvGemmParams& p, hipStream_t stream) {
    const char f = g_eae_launch_options.gemm;
    if (!(f == 's' || f == 'u' || f == 0)) return 1;
    int cut = f == 'u' ? 0 : -1;
    if (f == 's') cut = g_eae_launch_options.split_waves;
    const long tiles32 = ((long)p.n * ((p.hp + 3) / 4) * ((p.wp + 7) / 8)) * p.n_phases;
    if (f == 0 && tiles32 < 1024) return 1;
    return launch_split(p, stream, cut);
}

int launch(ConvGemmParams& p, hipStream_t stream) {
    // EAE_HIP_GEMM (read ONCE at library load, misc.hip; the parity tests of every form call eae_hip_debug_reload_launch_options):
    //   unset  conv_gemm_split.hip (one item per wave; the last tiles cut when the shape calls for it and the caller gave a
    //          workspace) for layers with at least one 32-position tile per SIMD, the small-layer forms below otherwise;
    //   's'    conv_gemm_split.hip with the cut forced, sized for EAE_HIP_SPLIT_WAVES (1..3, default 3) waves per SIMD;
    //   'u'    conv_gemm_split.hip, whole tiles only;   'w'  conv_gemm_wave_kernel;   'l'  block-cooperative LDS slabs.
    const char f = g_eae_launch_options.gemm;
    const int force_tile = g_eae_launch_options.force_tile, force_nt = g_eae_launch_options.force_nt;
    const int variant = f == 'l' ? 0 : 1;
    p.stamps = g_stamp_buffer;
    {
        const int rc = try_split(p, stream);
        if (rc != 1) return rc;
    }
    const long positions = (long)p.n * p.hp * p.wp;
    if (variant == 0) {          // block-cooperative LDS slabs (one barrier per K-step); phase is the fastest index
        bool big = positions * p.n_phases >= 128L * 512 && p.wp >= 16;
        if (force_tile) big = force_tile == 128;
        const int tile_w = big ? 16 : 8;
        p.tiles_r = (p.hp + TILE_H - 1) / TILE_H;
        p.tiles_c = (p.wp + tile_w - 1) / tile_w;
        const int grid = p.n * p.tiles_r * p.tiles_c * p.n_phases;
        if (big) hipLaunchKernelGGL(conv_gemm_kernel<128>, dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_gemm_kernel<64>, dim3(grid), dim3(256), 0, stream, p);
        EAE_HIP_CHECK_LAUNCH();
        return EAE_HIP_OK;
    }
    // wave-private pipelines, no barriers. Longest phase first (insertion sort of <= 4 descriptors).
    for (int i = 1; i < p.n_phases; ++i)
        for (int j = i; j > 0 && p.phase[j].ntaps > p.phase[j - 1].ntaps; --j) {
            const PhaseDesc tmp = p.phase[j]; p.phase[j] = p.phase[j - 1]; p.phase[j - 1] = tmp;
        }
    // Waves are independent, so the block size only sets the dispatch granularity: 64-position blocks (2 waves) unless
    // the layer is so small that even those leave CUs without work, then 32-position blocks (1 wave).
    int waves = positions >= 64L * 1024 ? 2 : 1;
    if (p.wp < 8 && waves == 2) waves = 1;
    // single-phase layers with fewer than three 32-position waves per SIMD (1024 SIMDs): 1-wave blocks, and their output
    // channels go to two blocks each below (measured on 64 images of 256x256: conv_2 0.62 -> 0.50 ms)
    const bool small_conv = p.n_phases == 1 && positions < 3L * 1024 * 32;
    if (small_conv) waves = 1;
    if (force_tile) waves = force_tile == 128 ? 4 : (force_tile == 64 ? 2 : 1);
    const int tile_w = waves * 32 / TILE_H;
    p.tiles_r = (p.hp + TILE_H - 1) / TILE_H;
    p.tiles_c = (p.wp + tile_w - 1) / tile_w;
    const int tiles = p.n * p.tiles_r * p.tiles_c;
    int grid = ((tiles + 7) / 8) * 8 * p.n_phases;           // padded: see the block decode in the kernel
    // A layer with fewer waves than ~2 per SIMD runs in coarse rounds (conv_3 at Kodak batch 24: 1152 waves on 1024 SIMDs
    // = two rounds, 48 % MFMA utilisation). Spread the output channels of each position tile over 2 (or 4) blocks and run
    // the normalisation as its own small pass over the output (gdn_kernel, in place: same arithmetic, same bits).
    // Below one wave per SIMD the launch is as long as ONE wave's chain, so the finer split wins as long as its waves still fit
    // the machine at once (one Kodak image, launches alone: conv_2 0.156 -> 0.121 ms, conv_3 0.122 -> 0.081, transpose_conv_1
    // 0.078 -> 0.063 with quarter tiles; transpose_conv_2, 768 items, 0.107 -> 0.093 ms with WHOLE tiles and the fused
    // epilogue instead of two rounds of half tiles and the extra normalisation pass: profiles/r03_gemm_forms_1x512x768.txt).
    int nt = 4;
    if (waves == 1) {
        const long items = (long)tiles * p.n_phases, simds = 4L * eae_compute_units();
        if (items * 4 <= simds) nt = 1;
        else if (items * 2 <= simds || (small_conv && items >= simds)) nt = 2;
        else if (items < 2048 && p.norm == EAE_NORM_NONE) nt = 2;
        // between half a wave and one wave per SIMD, a layer with a normalisation keeps whole tiles (conv_2 of four Kodak
        // images, 768 items: 0.223 ms whole, 0.252 as half tiles + the normalisation pass)
    }
    if (force_nt) nt = force_nt == 1 ? 1 : (force_nt == 2 ? 2 : 4);
    if (nt != 4 && waves == 1) {
        // four-wave blocks (the channel parts of nt position tiles: one wave per SIMD of a CU) when the wider tile wastes no lanes and
        // there is a block for every CU -- conv_3 of 64 x 256x256, 1,024 waves: 0.22 -> 0.13 ms in the step; below that the launch
        // does not fill the GPU either way and one-wave blocks spread over more CUs (one Kodak image at a time 1.18 -> 1.13 ms, but
        // pipelined 0.31 -> 0.33 ms per image) --, one-wave blocks otherwise (EAE_HIP_PACK=0 / 1 forces either: the parity tests run both)
        const int pack_w = nt * 32 / TILE_H;
        const int packed_tiles = p.n * p.tiles_r * (p.wp / pack_w);
        const int force_pack = g_eae_launch_options.pack;
        if (force_pack != 0 && p.wp % pack_w == 0 && (packed_tiles * p.n_phases >= eae_compute_units() || force_pack == 1)) {
            p.tiles_c = p.wp / pack_w;
            grid = ((packed_tiles + 7) / 8) * 8 * p.n_phases;
            if (nt == 2) hipLaunchKernelGGL((conv_gemm_wave_kernel<4, EAE_NORM_NONE, 2, true>), dim3(grid), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((conv_gemm_wave_kernel<4, EAE_NORM_NONE, 1, true>), dim3(grid), dim3(256), 0, stream, p);
        } else {
            grid *= 4 / nt;
            if (nt == 2) hipLaunchKernelGGL((conv_gemm_wave_kernel<1, EAE_NORM_NONE, 2>), dim3(grid), dim3(64), 0, stream, p);
            else hipLaunchKernelGGL((conv_gemm_wave_kernel<1, EAE_NORM_NONE, 1>), dim3(grid), dim3(64), 0, stream, p);
        }
        EAE_HIP_CHECK_LAUNCH();
        if (p.norm != EAE_NORM_NONE)
            return eae_hip_gdn(p.out, p.gamma, p.beta, p.norm == EAE_NORM_IGDN ? 1 : 0, p.out, (int64_t)p.n * p.hout * p.wout, stream);
        return EAE_HIP_OK;
    }
#define EAE_LAUNCH_WAVE(W_, N_) hipLaunchKernelGGL((conv_gemm_wave_kernel<W_, N_>), dim3(grid), dim3(W_ * 64), 0, stream, p)
#define EAE_LAUNCH_NORM(W_)                                                   \
    if (p.norm == EAE_NORM_GDN) EAE_LAUNCH_WAVE(W_, EAE_NORM_GDN);            \
    else if (p.norm == EAE_NORM_IGDN) EAE_LAUNCH_WAVE(W_, EAE_NORM_IGDN);     \
    else EAE_LAUNCH_WAVE(W_, EAE_NORM_NONE);
    if (waves == 4) { EAE_LAUNCH_NORM(4) } else if (waves == 2) { EAE_LAUNCH_NORM(2) } else { EAE_LAUNCH_NORM(1) }
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

}  // namespace

namespace {
// The kernels address inside ONE image with 32-bit byte offsets (buffer descriptors): an activation plane of 128 channels must
// stay below 2 GB, i.e. at most 4,194,303 positions (2048 x 2047) per image on the larger side of the layer.
bool plane_fits(long h, long w) { return h * w * EAE_C * (long)sizeof(float) <= 0x7FFFFFFFL; }

int conv5x5s2(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed, const float* beta,
              float* out, int n, int h, int w_in, unsigned int* workspace, void* stream) {
    if (!x || !w_packed || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (norm != EAE_NORM_NONE && (!gamma_packed || !beta)) return EAE_HIP_BAD_ARGUMENT;
    if ((h & 1) || (w_in & 1) || !plane_fits(h, w_in)) return EAE_HIP_BAD_SHAPE;
    ConvGemmParams p{};
    p.in = x; p.out = out; p.w = w_packed; p.bias = bias; p.gamma = gamma_packed; p.beta = beta; p.norm = norm;
    p.n = n; p.hin = h; p.win = w_in; p.hp = h / 2; p.wp = w_in / 2; p.hout = h / 2; p.wout = w_in / 2;
    p.in_stride = 2; p.out_stride = 1; p.n_phases = 1; p.split_ws = workspace;
    PhaseDesc& pd = p.phase[0];
    pd.out_a = 0; pd.out_b = 0; pd.ntaps = 25;
    for (int u = 0; u < 5; ++u)
        for (int v = 0; v < 5; ++v)        // SAME padding for k5 s2 on even sizes: 1 before, 2 after (appendix A.2)
            pd.tap[u * 5 + v] = pack_tap(u - 1, v - 1, u * 5 + v);
    return launch(p, (hipStream_t)stream);
}

int tconv5x5s2(const float* x, const float* w_packed, const float* bias, int norm, const float* gamma_packed, const float* beta,
               float* out, int n, int h, int w_in, unsigned int* workspace, void* stream) {
    if (!x || !w_packed || !out || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (norm != EAE_NORM_NONE && (!gamma_packed || !beta)) return EAE_HIP_BAD_ARGUMENT;
    if (!plane_fits(2L * h, 2L * w_in)) return EAE_HIP_BAD_SHAPE;
    ConvGemmParams p{};
    p.in = x; p.out = out; p.w = w_packed; p.bias = bias; p.gamma = gamma_packed; p.beta = beta; p.norm = norm;
    p.n = n; p.hin = h; p.win = w_in; p.hp = h; p.wp = w_in; p.hout = 2 * h; p.wout = 2 * w_in;
    p.in_stride = 1; p.out_stride = 2; p.n_phases = 4; p.split_ws = workspace;
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

}  // namespace

extern "C" int eae_hip_conv5x5s2(const float* x, const float* w_packed, const float* bias, int norm,
                                 const float* gamma_packed, const float* beta, float* out, int n, int h, int w_in,
                                 void* stream) {
    return conv5x5s2(x, w_packed, bias, norm, gamma_packed, beta, out, n, h, w_in, nullptr, stream);
}
extern "C" int eae_hip_tconv5x5s2(const float* x, const float* w_packed, const float* bias, int norm,
                                  const float* gamma_packed, const float* beta, float* out, int n, int h, int w_in,
                                  void* stream) {
    return tconv5x5s2(x, w_packed, bias, norm, gamma_packed, beta, out, n, h, w_in, nullptr, stream);
}
// conv_3 + bias with the latent stage behind it (latent_body.h) in ONE launch when the layer is large enough for
// conv_gemm_split_kernel; otherwise the convolution followed by eae_hip_latent_stage in place: same bits either way.
extern "C" int eae_hip_conv5x5s2_latent(const float* x, const float* w_packed, const float* bias, const float* gamma_in_packed,
                                        const float* beta_in, const float* map_mean, const float* bin_widths,
                                        const float* gamma_out_packed, const float* beta_out, float* y_out, float* shifted_out,
                                        float* t_out, int16_t* symbols_planar, uint32_t* nonzero_flags, uint32_t* checks, int n,
                                        int h, int w_in, void* workspace, void* stream) {
    if (!x || !w_packed || !bin_widths || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if ((gamma_in_packed == nullptr) != (beta_in == nullptr)) return EAE_HIP_BAD_ARGUMENT;
    if ((gamma_out_packed == nullptr) != (beta_out == nullptr) || (gamma_in_packed == nullptr) != (gamma_out_packed == nullptr))
        return EAE_HIP_BAD_ARGUMENT;                      // both normalisations (fixed bin widths) or neither (learned)
    const bool fixed = gamma_in_packed != nullptr;
    float* main_out = fixed ? t_out : shifted_out;        // the decoder's input: also where a cut tile's accumulators wait
    if (!main_out) return EAE_HIP_BAD_ARGUMENT;
    if ((h & 1) || (w_in & 1) || !plane_fits(h, w_in)) return EAE_HIP_BAD_SHAPE;
    ConvGemmParams p{};
    p.in = x; p.out = main_out; p.w = w_packed; p.bias = bias; p.gamma = gamma_in_packed; p.beta = beta_in;
    p.norm = fixed ? NORM_LATENT : NORM_LATENT_PLAIN;
    p.n = n; p.hin = h; p.win = w_in; p.hp = h / 2; p.wp = w_in / 2; p.hout = h / 2; p.wout = w_in / 2;
    p.in_stride = 2; p.out_stride = 1; p.n_phases = 1; p.split_ws = static_cast<unsigned int*>(workspace);
    p.map_mean = map_mean; p.bin_widths = bin_widths; p.gamma_out = gamma_out_packed; p.beta_out = beta_out;
    p.latent = LatentOut{y_out, shifted_out, t_out, symbols_planar, nonzero_flags, checks};
    PhaseDesc& pd = p.phase[0];
    pd.out_a = 0; pd.out_b = 0; pd.ntaps = 25;
    for (int u = 0; u < 5; ++u)
        for (int v = 0; v < 5; ++v) pd.tap[u * 5 + v] = pack_tap(u - 1, v - 1, u * 5 + v);
    p.stamps = g_stamp_buffer;
    int rc = try_split(p, (hipStream_t)stream);
    if (rc != 1) return rc;
    // small layer (or another kernel form forced): the two launches, the stage in place on the convolution's output
    rc = conv5x5s2(x, w_packed, bias, EAE_NORM_NONE, nullptr, nullptr, main_out, n, h, w_in, static_cast<unsigned int*>(workspace), stream);
    if (rc) return rc;
    return eae_hip_latent_stage(main_out, gamma_in_packed, beta_in, map_mean, bin_widths, gamma_out_packed, beta_out, y_out,
                                shifted_out, t_out, symbols_planar, nonzero_flags, checks, n, (h / 2) * (w_in / 2), stream);
}

extern "C" uint64_t eae_hip_conv_workspace_bytes(void) { return (uint64_t)SPLIT_WORDS * sizeof(unsigned int); }

// The error word of a workspace, in stream order: nothing to do when it is zero (one block, one load); otherwise the count
// of tails that gave up goes to *error_word (added) and the whole workspace -- flags that their consumers never reset, the
// word itself -- is zeroed again, so that the launches behind this one start from the documented state.
__global__ __launch_bounds__(1024) void conv_workspace_collect_kernel(unsigned int* ws, unsigned int* error_word) {
    const unsigned int e = ws[SPLIT_ERROR_WORD];
    if (e == 0u) return;
    __syncthreads();
    for (int i = threadIdx.x; i < SPLIT_WORDS; i += 1024) ws[i] = 0u;
    if (threadIdx.x == 0) { atomicAdd(error_word, e); __threadfence_system(); }
}
extern "C" int eae_hip_conv_workspace_collect(void* workspace, uint32_t* error_word, void* stream) {
    if (!workspace || !error_word) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(conv_workspace_collect_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       static_cast<unsigned int*>(workspace), error_word);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
extern "C" int eae_hip_conv5x5s2_ws(const float* x, const float* w_packed, const float* bias, int norm,
                                    const float* gamma_packed, const float* beta, float* out, int n, int h, int w_in,
                                    void* workspace, void* stream) {
    if (!workspace) return EAE_HIP_BAD_ARGUMENT;
    return conv5x5s2(x, w_packed, bias, norm, gamma_packed, beta, out, n, h, w_in, static_cast<unsigned int*>(workspace), stream);
}
extern "C" int eae_hip_tconv5x5s2_ws(const float* x, const float* w_packed, const float* bias, int norm,
                                     const float* gamma_packed, const float* beta, float* out, int n, int h, int w_in,
                                     void* workspace, void* stream) {
    if (!workspace) return EAE_HIP_BAD_ARGUMENT;
    return tconv5x5s2(x, w_packed, bias, norm, gamma_packed, beta, out, n, h, w_in, static_cast<unsigned int*>(workspace), stream);
}

namespace {
// src [rows][128] (or, transposed: [taps][128 cols-of-dst][128 rows-of-dst]) -> dst [rows][packed channel]
__global__ void pack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int transpose_blocks) {
    const long total = rows * EAE_C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / EAE_C;
        const int c = (int)(i % EAE_C);
        float v;
        if (transpose_blocks) {      // dst row = (tap, ci), dst col = co ; src = [tap][co][ci]
            const long tap = row / EAE_C;
            const int ci = (int)(row % EAE_C);
            v = src[(tap * EAE_C + c) * EAE_C + ci];
        } else {
            v = src[i];
        }
        dst[row * EAE_C + packed_channel(c)] = v;
    }
}
}  // namespace

// Diagnostic (test build only, -DEAE_TEST_HOOKS): when set (device pointer to grid * waves * 8 u64), the wave kernel records
// s_memtime stamps per wave. The product library has no way to set it: the pointer stays null there.
#ifdef EAE_TEST_HOOKS
extern "C" int eae_hip_debug_set_stamp_buffer(uint64_t* device_buffer) {
    g_stamp_buffer = reinterpret_cast<unsigned long long*>(device_buffer);
    return EAE_HIP_OK;
}
#endif

// [5][5][ci][co] (HWIO) -> [25][ci][packed co]
extern "C" int eae_hip_pack_conv_weights(const float* w_hwio, float* w_packed, int taps, void* stream) {
    if (!w_hwio || !w_packed || taps <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, w_hwio, w_packed, (long)taps * EAE_C, 0);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
// TF conv2d_transpose filter [5][5][co][ci] -> [25][ci][packed co]
extern "C" int eae_hip_pack_tconv_weights(const float* w_tf, float* w_packed, int taps, void* stream) {
    if (!w_tf || !w_packed || taps <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, (long)taps * EAE_C, 1);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
// gamma [128 k][128 c] -> [128 k][packed c]
extern "C" int eae_hip_pack_gamma(const float* gamma, float* gamma_packed, void* stream) {
    if (!gamma || !gamma_packed) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, gamma, gamma_packed, (long)EAE_C, 0);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
