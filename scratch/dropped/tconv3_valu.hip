// tconv3_valu.hip -- transpose_conv_3 (components.py:79-83: tf.nn.conv2d_transpose 9x9, 128 -> 1 channel, stride 4, 'SAME',
// no bias) + tls.cast_bt601 (tools.py:93) + the squared error of tls.psnr_2d (tools.py:873-875), on the VECTOR unit.
//
// Why not the matrix unit. This layer has ONE output channel: the only N a GEMM can be given is the 16 output phases
// (I mod 4, J mod 4) of an input site, and then every MFMA column must see all 9 neighbour sites although a phase only has
// 4, 6 or 9 real taps: 144 slots for 81 taps, 44 % of the issued work multiplies structural zeros (tconv3.hip). The f32 MFMA
// runs at exactly the vector FMA rate (64 FLOP / clk / SIMD: MI355X_MICROARCH.md), so the same chain as plain v_fma_f32 with
// the zeros skipped is the cheaper program: 81 x 128 FMAs per site instead of 144 x 128.
//
// Formulation. Output pixel (4p + a, 4q + b) receives tap (u, v) = (a + 2 - 4 dr, b + 2 - 4 dc) from site (p + dr, q + dc):
// every tap (u, v) belongs to exactly ONE phase (a, b) = ((u - 2) mod 4, (v - 2) mod 4) and ONE neighbour (dr, dc). A lane owns
// one site = 16 output pixels = 16 accumulators; per 32-channel block it walks the 81 taps in (u, v) order and, for each,
// runs acc[phase(u, v)] = fma(x[neighbour(u, v)][c], w[u][v][c], acc[...]) over the 32 channels. That is, for every output
// element, the oracle's chain: channel block (outer), u ascending, v ascending, channel inside the block, one accumulator
// (the 16 chains of a lane interleave, each keeps its own order) -> bit-identical to oracle/transforms_oracle.c and to the
// MFMA kernel. The weights are the same for all lanes: they come in through SCALAR loads straight from the TF layout
// [9][9][1][128] (no packing), the activations of the 3 neighbours of one row sit in 96 registers and are reused by up to
// 4 x 9 taps.
//
// One block = 256 threads = 16 x 16 sites -> 64 x 64 output pixels. Per channel block the 18 x 18 site patch (32 channels,
// 46 KB with padding) is staged in LDS (zero outside the image), three blocks per CU cover each other's staging.
#include "common.h"

#include <cstdlib>

namespace {
constexpr int TS = 16;                       // tile: TS x TS sites
constexpr int PT = TS + 2;                   // patch: 18 x 18 sites
constexpr int CB = 32;                       // channels per block
constexpr int SS = CB + 4;                   // floats per site in LDS (144 B: b128 reads of 16 neighbouring sites hit 16 slots)
constexpr int PATCH_FLOATS = PT * PT * SS;   // 11,664 floats = 46,656 B
constexpr int K9 = 9;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 3) void tconv3_valu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                             const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                             int h, int win, int tiles_r, int tiles_c, int debug_mode) {
    __shared__ __attribute__((aligned(16))) float patch[PATCH_FLOATS];
    __shared__ unsigned int red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = b % tiles_c; b /= tiles_c;
    const int tr = b % tiles_r;
    const int img = b / tiles_r;
    const float* x_img = x + (size_t)img * h * win * EAE_C;
    const int r0 = tr * TS - 1, c0 = tc * TS - 1;
    const int sr = tid >> 4, sc = tid & 15;                   // this lane's site inside the tile
    const float* centre = patch + ((sr + 1) * PT + (sc + 1)) * SS;

    f32x2 acc2[8];                                            // [phase row a][phase column pair]: (a, 0..1), (a, 2..3)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = (f32x2){0.f, 0.f};

    // staging: 18 x 18 sites x 32 channels = 2592 float4 per channel block, 10 or 11 per thread, in two batches of loads that
    // are all in flight before the first store (a load -> store loop paid one L2 round trip per float4)
    constexpr int PER_THREAD = (PT * PT * (CB / 4) + 255) / 256;      // 11
    constexpr int BATCH = 6;
    for (int cb = 0; cb < EAE_C / CB; ++cb) {
        if (cb) __syncthreads();                              // everybody is done reading the previous channel block
        if (!((debug_mode & 1) && cb))
#pragma unroll
        for (int j0 = 0; j0 < PER_THREAD; j0 += BATCH) {
            float4 stage[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int i = tid + 256 * (j0 + j);
                const int site = i >> 3, q = i & 7;
                const int r = r0 + site / PT, c = c0 + site % PT;
                const bool ok = j0 + j < PER_THREAD && i < PT * PT * (CB / 4) && (unsigned)r < (unsigned)h && (unsigned)c < (unsigned)win;
                stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) stage[j] = *reinterpret_cast<const float4*>(x_img + ((size_t)r * win + c) * EAE_C + cb * CB + 4 * q);
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int i = tid + 256 * (j0 + j);
                if (j0 + j < PER_THREAD && i < PT * PT * (CB / 4)) *reinterpret_cast<float4*>(patch + (i >> 3) * SS + 4 * (i & 7)) = stage[j];
            }
        }
        __syncthreads();
        if (debug_mode & 2) continue;
        const float* wcb = w + cb * (K9 * K9 * CB);           // this channel block's weights in unit order (pack_tconv3_units_kernel)
        float xs[3][CB];                                      // the three neighbours (dc = +1, 0, -1) of the current row dr
        // Work unit = 32 weights = two s_load_dwordx16: per kernel row u, taps {0..3} x 8 channels (4 units), taps {4..7} x 8
        // channels (4 units), tap 8 x 32 channels (1 unit). The weights are the same for every lane: scalar loads, scalar FMA
        // operands. In the 4-tap units consecutive (packed) FMAs alternate between two accumulator pairs; every accumulator
        // still sees its own taps in v order and its channels in ascending order.
        //  * Two weight buffers; the next unit's loads go out first (pinned by a scheduling barrier), this unit's FMAs cover
        //    their latency. The barriers also keep the compiler from hoisting more units' worth of scalars than the file holds
        //    (left alone it kept hundreds live, spilled them into vector lanes and packed pairs of FMAs into v_pk_fma_f32
        //    behind ~1,700 s_mov per channel block: the CU's single scalar unit became the bound, 0.53 ms per Kodak batch).
        //  * Scalar loads return out of order, so the only wait there is for them is lgkmcnt(0): placed by hand BEFORE the next
        //    unit's loads are issued (current weights ready, nothing else outstanding). Placed by the compiler at the first FMA
        //    it also covered the loads issued a moment earlier: a full scalar-cache round trip per unit (0.35 ms).
        float wbuf[2][CB];
#define EAE_T3V_LOAD(n_) _Pragma("unroll") for (int c = 0; c < CB; ++c) wbuf[(n_) & 1][c] = wcb[(n_) * CB + c];
        EAE_T3V_LOAD(0)
#pragma unroll
        for (int u = 0; u < K9; ++u)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int n = u * 9 + k;
            const int a = (u + 2) & 3;                        // (u - 2) mod 4
            const int dr = (a + 2 - u) / 4;                   // +1, 0, -1 as u grows
            if (k == 0 && (u == 0 || u == 2 || u == 6)) {     // dr changed: fetch the row's three neighbours
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* src = centre + (dr * PT + (1 - j)) * SS;      // j = 0, 1, 2 <-> dc = +1, 0, -1
#pragma unroll
                    for (int q = 0; q < CB / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
                        xs[j][4 * q] = t.x; xs[j][4 * q + 1] = t.y; xs[j][4 * q + 2] = t.z; xs[j][4 * q + 3] = t.w;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): this unit's weights (and the row's activations) are in
            if (n + 1 < K9 * 9) { EAE_T3V_LOAD(n + 1) }
            __builtin_amdgcn_sched_barrier(0);
            if (k < 8) {
                // taps (v, v + 1) of a pair share their neighbour site and feed neighbouring phases: one packed FMA
                // (v_pk_fma_f32: lo and hi are two independent IEEE FMAs) = activation broadcast x weight pair + accumulator pair
                const int g = k >> 2, o = k & 3;
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const int v = 4 * g + 2 * jp;               // the pair's first tap
                        const int bq = (v + 2) & 3;                 // 2 (taps 0, 1 / 4, 5) or 0 (taps 2, 3 / 6, 7): even
                        const int dc = (bq + 2 - v) / 4;
                        const float xv = xs[1 - dc][8 * o + c];
                        const f32x2 wv = {wbuf[n & 1][4 * c + 2 * jp], wbuf[n & 1][4 * c + 2 * jp + 1]};
                        acc2[a * 2 + (bq >> 1)] = __builtin_elementwise_fma((f32x2){xv, xv}, wv, acc2[a * 2 + (bq >> 1)]);
                    }
            } else {                                           // v = 8: phase column 2, neighbour dc = -1
#pragma unroll
                for (int c = 0; c < CB; ++c) acc2[a * 2 + 1].x = fmaf(xs[2][c], wbuf[n & 1][c], acc2[a * 2 + 1].x);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef EAE_T3V_LOAD
    }
    __syncthreads();                                           // everybody is done reading the patch
    // ---- epilogue: the 64 x 64 pixel tile through LDS, then 16 consecutive pixels per thread ------------------------
    float* ot = patch;                                         // [64][64]
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(ot + (4 * sr + a) * 64 + 4 * sc) = make_float4(acc2[2 * a].x, acc2[2 * a].y, acc2[2 * a + 1].x, acc2[2 * a + 1].y);
    __syncthreads();
    const int ho = 4 * h, wo = 4 * win;
    const int prow = tid >> 2, pcol = (tid & 3) * 16;
    const int gr = tr * TS * 4 + prow;
    unsigned int se = 0;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        const int gc = tc * TS * 4 + pcol + 4 * part;
        if (gr < ho && gc < wo) {                              // wo is a multiple of 4: the 4 pixels are inside together
            const float4 v = *reinterpret_cast<const float4*>(ot + prow * 64 + pcol + 4 * part);
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
        if (tid == 0) atomicAdd(&sse[img], (unsigned long long)red[0] + red[1] + red[2] + red[3]);
    }
}
// TF filter [9][9][1][128] -> [4 channel blocks][81 units][32]: unit u * 9 + k holds, for k < 8, taps v = 4 (k / 4) + j, j = 0..3,
// channels 8 (k % 4) + c, c = 0..7, at [4 c + j] (the weight pairs of the packed FMAs adjacent); for k = 8, tap v = 8,
// channels c = 0..31, at [c].
__global__ void pack_tconv3_units_kernel(const float* __restrict__ w_tf, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K9 * K9 * EAE_C) return;
    const int e = i % CB, n = (i / CB) % (K9 * K9), cb = i / (CB * K9 * K9);
    const int u = n / 9, k = n % 9;
    int v, c;
    if (k < 8) { v = 4 * (k >> 2) + (e & 3); c = 8 * (k & 3) + (e >> 2); }
    else { v = 8; c = e; }
    dst[i] = w_tf[(u * K9 + v) * EAE_C + cb * CB + c];
}
}  // namespace

int eae_tconv3_valu_pack(const float* w_tf, float* units, hipStream_t stream) {
    hipLaunchKernelGGL(pack_tconv3_units_kernel, dim3((K9 * K9 * EAE_C + 255) / 256), dim3(256), 0, stream, w_tf, units);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

// w_units: the filter in unit order (eae_tconv3_valu_pack)
int eae_tconv3_valu_launch(const float* x, const float* w_tf, float* out_f32, uint8_t* out_u8, const uint8_t* ref_u8,
                           uint64_t* sse, int n, int h, int w_in, hipStream_t stream) {
    const int tiles_r = (h + TS - 1) / TS, tiles_c = (w_in + TS - 1) / TS;
    hipLaunchKernelGGL(tconv3_valu_kernel, dim3(n * tiles_r * tiles_c), dim3(256), 0, stream, x, w_tf, out_f32, out_u8, ref_u8,
                       reinterpret_cast<unsigned long long*>(sse), h, w_in, tiles_r, tiles_c, std::getenv("EAE_T3V_DEBUG") ? std::atoi(std::getenv("EAE_T3V_DEBUG")) : 0);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
