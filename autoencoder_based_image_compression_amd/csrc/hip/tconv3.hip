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
// One block = WAVES waves (4) = a tile of 8 x 16 sites -> 32 x 64 output pixels; wave w owns tile rows 2w, 2w+1 (two
// independent accumulator chains, enough to keep the 32-cycle MFMA issue rate alone on its SIMD). The input patch (10 x 18
// sites) is staged in LDS PASSES (4) times, 32 channels at a time, with each site's channels permuted to [ci mod 4][ci / 4]:
// the 8 k-values a lane needs per (channel block, neighbour) are then two conflict-free ds_read_b128 per tile row.
// 40 KB per block: three blocks = three waves per SIMD. (Two waves on 4 x 16 sites, the first persistent form, re-read
// 1.69 x the input for its halo against 1.41 x here: 0.196 against 0.192 ms per Kodak batch, 23 against 17 us for one image.)
// Weights are pre-packed per lane ([block][neighbour][lane][8]) and stream from L1/L2 through a register ring RING steps
// ahead: no barrier inside the K loop. Epilogue: the pixel tile goes through LDS, 8 pixels per thread, BT.601 cast, exact
// integer squared error (wave shuffle -> one u64 atomic per tile).
// Bound: MFMA for the contraction (2,304 issued / 1,296 algorithmic FLOP per pixel); 32 B/px read is the HBM term.
#include "common.h"

namespace {
#ifndef EAE_T3_WAVES
#define EAE_T3_WAVES 4
#endif
constexpr int WAVES = EAE_T3_WAVES;           // waves per block, two tile rows each
constexpr int NT = 64 * WAVES;                // threads per block
constexpr int TH = 2 * WAVES, TW = 16;
#ifndef EAE_T3_PASSES
#define EAE_T3_PASSES 4
#endif
constexpr int PASSES = EAE_T3_PASSES;         // the input patch is staged PASSES times, 128 / PASSES channels at a time
constexpr int HALF_C = EAE_C / PASSES;
constexpr int REGION = HALF_C / 4 + 4;        // floats per (site, ci mod 4): HALF_C / 4 used + 4 pad (20 or 12)
constexpr int PS = PASSES == 2 ? 88 : 56;     // floats per site: 4 regions, rounded so that 8 consecutive sites hit 8 distinct
constexpr int PATCH_R = TH + 2, PATCH_C = TW + 2;     //   16-byte bank groups (PS/4 = 22 or 14) -> conflict-free b128 reads
constexpr int PATCH_FLOATS = PATCH_R * PATCH_C * PS;   // 38,016 B (4 blocks = 8 waves per CU) or 24,192 B (6 blocks)
constexpr int Q = HALF_C / 4;                 // float4 per site per pass
constexpr int LOADS = (PATCH_R * PATCH_C * Q + NT - 1) / NT;   // per thread per pass
constexpr int STEPS = 4 * 9;                  // (channel block, neighbour)
constexpr int OT_STRIDE = 68;                 // floats per row of the epilogue's pixel tile in LDS (64 pixels + 4)
#ifndef EAE_T3_RING
#define EAE_T3_RING 6
#endif
constexpr int RING = EAE_T3_RING;             // steps of weights in flight (2 float4 each); divides STEPS: the ring runs on
static_assert(STEPS % RING == 0, "ring");     //   from one tile into the next
constexpr int BLOCKS_PER_CU = (PASSES == 2 ? 4 : 6) * 2 / WAVES;    // by LDS (38 or 24 KB each with two waves)
constexpr int WAVES_PER_SIMD = PASSES == 2 ? 2 : 3;

#ifdef EAE_T3_TRACE                   // scratch/t3_trace.sh: cycles per phase, summed behind the per-image squared errors
#define T3_MARK(i_) { const long long t_ = clock64(); tr_acc[i_] += t_ - tr_last; tr_last = t_; }
#else
#define T3_MARK(i_)
#endif

struct Tile { int img, tr, tc; };
__device__ __forceinline__ Tile tile_of(int t, int tiles_r, int tiles_c) {
    Tile r;
    r.tc = t % tiles_c; t /= tiles_c;
    r.tr = t % tiles_r;
    r.img = t / tiles_r;
    return r;
}

// Persistent blocks: block b works through tiles first + slot, first + slot + S, ... of its XCD's contiguous share (blocks b
// and b + 8 share an XCD: neighbouring tiles, which share halo sites, run at the same time under the same L2). While the
// MFMAs of one pass run, the input sites of the next pass -- the next channels of this tile, or the first ones of the block's
// next tile -- are already on their way into registers, and so are the tile's reference pixels for the epilogue: the only
// staging time left on the critical path is the LDS write between two barriers.
__global__ __launch_bounds__(NT, WAVES_PER_SIMD) void tconv3_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                                     float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                                     const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                                     int n_tiles, int h, int win, int tiles_r, int tiles_c) {
    __shared__ __attribute__((aligned(16))) float patch[PATCH_FLOATS];
    __shared__ unsigned int red[WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, S = (int)gridDim.x >> 3;
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int first = xcd * q8 + (xcd < r8 ? xcd : r8);
    if (slot >= cnt) return;
    // weights: lane-private 32 bytes per step, streamed from L1 / L2 through a register ring RING steps ahead
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
    // A pass = HALF_C channels of the patch (the K order is channel-block outer anyway). PATCH_R x PATCH_C sites x Q float4,
    // LOADS per thread; sites outside the image read zero (offset
    // beyond the buffer: zero-fill at THIS layer, appendix C.3); channel ci of the pass lands at (ci & 3) * REGION + (ci >> 2).
    const int img_bytes = h * win * EAE_C * (int)sizeof(float);
    // what never changes from tile to tile, per load j of this thread (patch element tid + NT j): the site's place inside the
    // patch (row, column), its byte offset from the patch's first site, and where its four values go in LDS
    int rel_rc[LOADS], rel_off[LOADS], lds_at[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int i = tid + NT * j;
        const int site = i / Q, q = i % Q;
        const int rr = site / PATCH_C, cc = site % PATCH_C;
        rel_rc[j] = i < PATCH_R * PATCH_C * Q ? (rr << 16 | cc) : 0x40000000;     // beyond the patch: a row no image has
        rel_off[j] = ((rr * win + cc) * EAE_C + 4 * q) * 4;
        lds_at[j] = site * PS + q;
    }
    float4 v[LOADS];
#define EAE_T3_FETCH(tile_, pass_)                                                                                   \
    {                                                                                                                \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                        \
            const_cast<float*>(x + (size_t)(tile_).img * h * win * EAE_C), 0, img_bytes, 0x00020000);                \
        const int r0_ = (tile_).tr * TH - 1, c0_ = (tile_).tc * TW - 1;                                              \
        const int base_ = ((r0_ * win + c0_) * EAE_C + HALF_C * (pass_)) * 4;                                        \
        _Pragma("unroll") for (int j = 0; j < LOADS; ++j) {                                                          \
            const int r = r0_ + (rel_rc[j] >> 16), c = c0_ + (rel_rc[j] & 0xFFFF);                                   \
            const bool ok = (unsigned)r < (unsigned)h && (unsigned)c < (unsigned)win;                                \
            const u32x4 t_ = __builtin_amdgcn_raw_buffer_load_b128(rs_, ok ? base_ + rel_off[j] : -1, 0, 0);         \
            v[j] = make_float4(__uint_as_float(t_.x), __uint_as_float(t_.y), __uint_as_float(t_.z),                  \
                               __uint_as_float(t_.w));                                                               \
        }                                                                                                            \
    }
    const int i16 = lane & 15, kq = lane >> 4;
    const float* a_base = patch + ((2 * wave + 1) * PATCH_C + (i16 + 1)) * PS + kq * REGION;   // site (row 2w, col i16)
    const int ho = 4 * h, wo = 4 * win;
    const int prow = tid >> 3, pcol = (tid & 7) * 8;             // epilogue: this thread's 8 pixels of the 16 x 64 tile
    Tile cur = tile_of(first + slot, tiles_r, tiles_c);
#ifdef EAE_T3_TRACE
    long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = clock64();
#endif
    EAE_T3_FETCH(cur, 0)
#pragma unroll
    for (int i = 0; i < RING; ++i) EAE_T3_LOAD(i, i)
    // Everything above has landed before the loop is entered. Without this the compiler, merging the loop's back edge (fetch
    // long complete, only weight loads in flight) with this entry (fetch in flight), put s_waitcnt vmcnt(0) in front of the LDS
    // write of EVERY tile: a wait for the weight loads issued a few hundred cycles earlier, for nothing.
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
    for (int k = slot; k < cnt; k += S) {
        const bool more = k + S < cnt;
        const Tile nxt = tile_of(first + (more ? k + S : k), tiles_r, tiles_c);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        unsigned int ref_px[2] = {0u, 0u};
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            T3_MARK(7)
            __syncthreads();                               // nobody reads the patch any more (previous pass / previous epilogue)
            T3_MARK(0)
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                if (tid + NT * j < PATCH_R * PATCH_C * Q) {
                    float* dst = patch + lds_at[j];
                    dst[0] = v[j].x; dst[REGION] = v[j].y; dst[2 * REGION] = v[j].z; dst[3 * REGION] = v[j].w;
                }
            }
            T3_MARK(1)
            __syncthreads();
            T3_MARK(2)
            if (pass + 1 < PASSES) EAE_T3_FETCH(cur, pass + 1)
            else {
                if (more) EAE_T3_FETCH(nxt, 0)
                if (ref) {                                 // the tile's reference pixels, for the epilogue behind these MFMAs
                    const int gr_ = cur.tr * TH * 4 + prow, gc_ = cur.tc * TW * 4 + pcol;
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        ref_px[half] = (gr_ < ho && gc_ + 4 * half < wo)
                            ? *reinterpret_cast<const unsigned int*>(ref + ((size_t)cur.img * ho + gr_) * wo + gc_ + 4 * half) : 0u;
                }
            }
            T3_MARK(3)
            // the A fragments of step ls + 1 are read from LDS before the 16 MFMAs of step ls are issued (two register sets)
            float4 af[2][4];
#define EAE_T3_READ_A(dst_, ls_)                                                                                     \
            {                                                                                                        \
                const int nb_ = (ls_) % 9;                                                                           \
                const float* a0p_ = a_base + ((1 - nb_ / 3) * PATCH_C + (1 - nb_ % 3)) * PS + 8 * ((ls_) / 9);       \
                dst_[0] = *reinterpret_cast<const float4*>(a0p_);                                                    \
                dst_[1] = *reinterpret_cast<const float4*>(a0p_ + 4);                                                \
                dst_[2] = *reinterpret_cast<const float4*>(a0p_ + PATCH_C * PS);                                     \
                dst_[3] = *reinterpret_cast<const float4*>(a0p_ + PATCH_C * PS + 4);                                 \
            }
            EAE_T3_READ_A(af[0], 0)                        // neighbours (dr, dc) = (+1,+1), (+1,0), ... (-1,-1), per channel block
#pragma unroll
            for (int ls = 0; ls < STEPS / PASSES; ++ls) {
                const int step = pass * (STEPS / PASSES) + ls;
                if (ls + 1 < STEPS / PASSES) EAE_T3_READ_A(af[(ls + 1) & 1], ls + 1)
                __builtin_amdgcn_sched_barrier(0);
                const float4 a00 = af[ls & 1][0], a01 = af[ls & 1][1], a10 = af[ls & 1][2], a11 = af[ls & 1][3];
                const float4 w0 = ring[step % RING][0], w1 = ring[step % RING][1];
                acc0 = mfma16(a00.x, w0.x, acc0); acc1 = mfma16(a10.x, w0.x, acc1);
                acc0 = mfma16(a00.y, w0.y, acc0); acc1 = mfma16(a10.y, w0.y, acc1);
                acc0 = mfma16(a00.z, w0.z, acc0); acc1 = mfma16(a10.z, w0.z, acc1);
                acc0 = mfma16(a00.w, w0.w, acc0); acc1 = mfma16(a10.w, w0.w, acc1);
                acc0 = mfma16(a01.x, w1.x, acc0); acc1 = mfma16(a11.x, w1.x, acc1);
                acc0 = mfma16(a01.y, w1.y, acc0); acc1 = mfma16(a11.y, w1.y, acc1);
                acc0 = mfma16(a01.z, w1.z, acc0); acc1 = mfma16(a11.z, w1.z, acc1);
                acc0 = mfma16(a01.w, w1.w, acc0); acc1 = mfma16(a11.w, w1.w, acc1);
                EAE_T3_LOAD(step % RING, (step + RING) % STEPS)      // past the last step: the next tile's first steps
                __builtin_amdgcn_sched_barrier(0);
            }
#undef EAE_T3_READ_A
            T3_MARK(4)
        }
        __syncthreads();                                   // both waves are done reading the patch
        T3_MARK(5)
        // ---- epilogue: 16 x 64 pixel tile through LDS, then 8 consecutive pixels per thread ------------------------
        float* ot = patch;                                 // [16][OT_STRIDE]: rows 4 floats apart modulo the banks (the four rows a lane quad
                                                           // group writes at once land in distinct banks; 64 was a four-way conflict)
        {
            const int a = i16 >> 2, bq = i16 & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ot[(8 * wave + a) * OT_STRIDE + 4 * (4 * kq + r) + bq] = acc0[r];
                ot[(8 * wave + 4 + a) * OT_STRIDE + 4 * (4 * kq + r) + bq] = acc1[r];
            }
        }
        __syncthreads();
        const int gr = cur.tr * TH * 4 + prow, gc = cur.tc * TW * 4 + pcol;
        unsigned int se = 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int gcc = gc + 4 * half;
            if (gr < ho && gcc < wo) {     // wo is a multiple of 4, so the 4 pixels are inside together
                const float4 o4 = *reinterpret_cast<const float4*>(ot + prow * OT_STRIDE + pcol + 4 * half);
                const size_t o = ((size_t)cur.img * ho + gr) * wo + gcc;
                if (out_f32) *reinterpret_cast<float4*>(out_f32 + o) = o4;
                if (out_u8 || ref) {
                    // tls.cast_bt601: clip to [16, 235], round half to even, uint8
                    const unsigned int q0 = (unsigned int)round_half_even(fminf(fmaxf(o4.x, 16.f), 235.f));
                    const unsigned int q1 = (unsigned int)round_half_even(fminf(fmaxf(o4.y, 16.f), 235.f));
                    const unsigned int q2 = (unsigned int)round_half_even(fminf(fmaxf(o4.z, 16.f), 235.f));
                    const unsigned int q3 = (unsigned int)round_half_even(fminf(fmaxf(o4.w, 16.f), 235.f));
                    if (out_u8) *reinterpret_cast<unsigned int*>(out_u8 + o) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
                    if (ref) {
                        const unsigned int rv = ref_px[half];
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
            if (tid == 0) {
                unsigned long long total = 0;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) total += red[i];
                atomicAdd(&sse[cur.img], total);
            }
        }
        cur = nxt;
        T3_MARK(6)
    }
#ifdef EAE_T3_TRACE
    if (lane == 0 && out_f32 == nullptr && sse) {
        for (int i = 0; i < 8; ++i) atomicAdd(&sse[64 + i], (unsigned long long)tr_acc[i]);
    }
#endif
#undef EAE_T3_LOAD
#undef EAE_T3_FETCH
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
    if ((long)h * w_in * EAE_C * (long)sizeof(float) > 0x7FFFFFFFL) return EAE_HIP_BAD_SHAPE;       // 32-bit offsets inside an image
    const int tiles_r = (h + TH - 1) / TH, tiles_c = (w_in + TW - 1) / TW;
    const long n_tiles = (long)n * tiles_r * tiles_c;
    if (n_tiles > 0x7FFFFFFFL) return EAE_HIP_BAD_ARGUMENT;
    // persistent blocks: as many as the GPU holds at once, in 8 XCD shares
    const int cus = eae_compute_units();
    if (cus <= 0) return (int)hipErrorInvalidDevice;
    long grid = (long)cus * BLOCKS_PER_CU;
    if (grid > n_tiles) grid = n_tiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(tconv3_kernel, dim3((unsigned)grid), dim3(NT), 0, (hipStream_t)stream, x, w_phase, out_f32,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), (int)n_tiles, h, w_in, tiles_r, tiles_c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
