// tconv3.hip -- transpose_conv_3 (eae/graph/components.py:79-83: tf.nn.conv2d_transpose 9x9, 128 -> 1 channel,
// stride 4, 'SAME', no bias) fused with what follows it on the path: tls.cast_bt601 (tools/tools.py:93) and the
// squared error of tls.psnr_2d (tools.py:873-875).
//
// Formulation (round 6): the deconvolution as a dense GEMM followed by col2im.
//     part[site][tap (u, v)] = sum over ci = 0 .. 127 ascending of X[site][ci] * W[u][v][ci]           81 taps per input site
//     out[4p'+a][4q'+b]      = the <= 9 parts with u = a + 2 - 4 dr, v = b + 2 - 4 dc of the sites (p'+dr, q'+dc), added
//                              site by site in raster order (dr ascending, then dc ascending), starting from +0
// which is oracle/transforms_oracle.c: orc_conv2d_transpose_same_col2im, bit for bit. Rounds 1-5 ran every output pixel as ONE
// fmaf chain over (channel block, u, v, channel): as a matrix product that is N = 16 output phases per neighbour site, of
// which 81 / 144 exist -- 7 of every 16 products were structural zeros, and the kernel sat at 0.40 of the f32 MFMA peak for four
// rounds. Here the GEMM is [81 taps, padded to 96] x [128] x [sites]: 81 / 96 live, every site read ONCE (no halo in the
// operand), and the overlap-add is ~5 float additions per output pixel.
//
// One block = 4 waves = one per SIMD, persistent over a STRIP of consecutive site rows (all images of the batch stacked), full
// width (segments of <= 512 sites for wider images). A chunk = 64 consecutive sites of a row, 16 per wave:
//   * the whole filter lives in registers for the life of the block (v_mfma_f32_16x16x4_f32, A = taps x k: 6 tap tiles x 32
//     k-steps = 192 registers per lane, loaded once); the 16 sites of a wave are the B operand, fetched straight from HBM as
//     eight 16-byte loads per lane one chunk ahead and turned to k-ascending order in registers (v_permlane16_swap /
//     v_permlane32_swap: a 4 x 4 transpose between the register index and the lane row, four instructions per four registers);
//   * 192 MFMAs per wave and chunk, two tap tiles interleaved at a time (their accumulators alternate: no dependent-issue
//     stall), the parts of a chunk written tap-major into one of two LDS buffers ([position][site + 2 zero sites either side]);
//   * col2im of the PREVIOUS chunk rides in the same instruction stream, under the MFMAs: a thread takes one quad of four output
//     pixels of one kernel row u -- nine parts from three neighbouring sites (consecutive lanes = consecutive sites: conflict-free)
//     plus the quad's running sum, a rolling accumulator of nine pixel rows in LDS ((4 p + u) mod 9) --, adds them in site order
//     and either writes the sum back or, when no later site contributes (u <= 3: the next site row has been through), casts to
//     BT.601 uint8, stores four pixels as one dword, adds the squared error and returns the slot as zeros. Straight-line code:
//     rows out of range, quads of another segment and idle lanes go to a trash row, stores to an out-of-range buffer offset.
//   * one barrier per chunk, placed behind the first third of the NEXT chunk's MFMAs, so that the last parts of a chunk are in
//     LDS long before anybody waits for them.
// Site rows at a strip's ends that belong to a neighbour strip are recomputed (only the tap tiles that reach across: u >= 6 of
// the row above = tiles 0-1, u <= 1 of the row below = tiles 4-5: the tap positions are ordered for that): 2/3 of a row per strip.
// Bound: MFMA (1,296 algorithmic / 1,536 issued FLOP per pixel); 32 B/px read + 1 B/px written is the HBM term.
#include "common.h"

#include <type_traits>

namespace {
constexpr int NT = 256;                        // 4 waves, one per SIMD (the filter in registers needs the whole register file)
constexpr int CHUNK = 64;                      // sites per chunk, 16 per wave
constexpr int PSITES = CHUNK + 4;              // a row of the part buffer: 2 zero sites | 64 sites | 2 zero sites
constexpr int NPOS = 84;                       // rows of the part buffer: 82 tap positions + 2 trash rows for tile 5's idle lanes
constexpr int P_FLOATS = NPOS * PSITES;        // 22,848 B per buffer, two buffers
constexpr int SLOTS = 9;                       // rolling pixel rows; row 9 of the accumulator is the trash row
constexpr int MAX_SEG_CHUNKS = 8;              // a segment is at most 512 sites wide (the accumulator rows must fit in LDS)
constexpr int W_FLOATS = 6 * 8 * 64 * 4;       // packed filter: [tile][k-step / 4][lane][4]

// Position of tap (u, v) in the GEMM's M dimension. Kernel rows in the order 6 7 8 | 2 3 4 5 | (one unused position) 0 1, so that
// the rows a strip's upper neighbour needs (u >= 6) are tiles 0-1 and those its lower neighbour needs (u <= 1) are tiles 4-5.
__host__ __device__ constexpr int tap_rank(int u) { return u >= 6 ? u - 6 : (u >= 2 ? u + 1 : u + 7); }
__host__ __device__ constexpr int tap_pos(int u, int v) { return 9 * tap_rank(u) + v + (tap_rank(u) >= 7 ? 1 : 0); }
static_assert(tap_pos(8, 8) == 26 && tap_pos(5, 8) == 62 && tap_pos(0, 0) == 64 && tap_pos(1, 8) == 81, "tap positions");

enum { ROW_TOP = 0, ROW_BODY = 1, ROW_BOTTOM = 2 };

#ifdef EAE_T3_TRACE                   // scratch/r06/t3_trace.py: shader clock ticks per stretch of a chunk, summed behind the squared errors
#define T3_STAMP(i_) const long long ts##i_ = __builtin_amdgcn_s_memtime();
#define T3_ADD(i_, a_, b_) tr_acc[i_] += ts##b_ - ts##a_;
#else
#define T3_STAMP(i_)
#define T3_ADD(i_, a_, b_)
#endif

// what the col2im of a chunk needs to know (all wave-uniform)
struct Gather {
    int on;                 // 0: no previous chunk (first iteration)
    int img, p, c0;         // image, site row, first site of the chunk
    int ulo, uhi, ufin;     // kernel rows taken; rows <= ufin are complete behind this site row
    int edge_final;         // last chunk of the row: the two quads behind its last complete one end here too
    int s0;                 // (4 p) mod 9
    int buf;                // which part buffer
};

__device__ __forceinline__ void swap16(float& a, float& b) {
    // rows (16 lanes) 1 and 3 of a <-> rows 0 and 2 of b (v_permlane16_swap_b32)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

__global__ __launch_bounds__(NT, 1) void tconv3_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                        float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                        const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                        int n, int h, int w, int rows_per_strip, int n_row_strips, int seg_chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n16 = lane & 15;
    // ---- this block's strip: site rows [g0, g1) of the stacked batch, chunks [kown0, kown1) of every row ---------------------
    const int b = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int seg = b / n_row_strips, strip = b - seg * n_row_strips;
    const int rows_total = n * h;
    const int g0 = strip * rows_per_strip, g1 = min(rows_total, g0 + rows_per_strip);
    const int chunks_total = (w + CHUNK - 1) / CHUNK;
    const int kown0 = seg * seg_chunks, kown1 = min(chunks_total, kown0 + seg_chunks);
    if (g0 >= g1 || kown0 >= kown1) return;
    const int kbeg = kown0 > 0 ? kown0 - 1 : 0, kend = kown1 < chunks_total ? kown1 + 1 : kown1;   // + the chunk either side
    const int qown0 = kown0 * CHUNK, qown1 = kown1 * CHUNK;                                         // quads this block owns
    const int acc_row = (kown1 - kown0) * CHUNK;                                                     // float4 per accumulator row
    float4* const ACC = reinterpret_cast<float4*>(lds);
    float* const P0 = lds + (SLOTS + 1) * acc_row * 4;
    const int p_first = g0 % h, p_last = (g1 - 1) % h;
    const int has_top = p_first > 0, has_bot = p_last < h - 1;
    const int gs = g0 - has_top, ge = g1 + has_bot;
    const int n_chunks = (ge - gs) * (kend - kbeg);

#ifdef EAE_T3_TRACE
    long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long ts_begin = __builtin_amdgcn_s_memtime();
#endif
    // ---- the filter: 6 tiles x 32 k-steps, one register each, for the life of the block --------------------------------------
    float wreg[6][32];
    {
        const float4* wp = reinterpret_cast<const float4*>(wq) + lane;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const float4 v = wp[(t * 8 + s4) * 64];
                wreg[t][4 * s4 + 0] = v.x; wreg[t][4 * s4 + 1] = v.y; wreg[t][4 * s4 + 2] = v.z; wreg[t][4 * s4 + 3] = v.w;
            }
    }
    // ---- LDS: accumulator rows and both part buffers start as zeros (the zero sites of the part buffers stay zeros) ----------
    {
        const int total4 = ((SLOTS + 1) * acc_row * 4 + 2 * P_FLOATS) / 4;
        for (int i = tid; i < total4; i += NT) ACC[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- per-thread constants ---------------------------------------------------------------------------------------------------
    const int x_lane = ((16 * wave + n16) * EAE_C + 4 * kq) * 4;                  // bytes, inside a chunk
    const int pw = (4 * kq * PSITES + 2 + 16 * wave + n16) * 4;                   // bytes: this lane's first part of a tile
    const int pw5 = kq == 0 ? pw + 80 * PSITES * 4 : (82 * PSITES + 2 + 16 * wave + n16) * 4;   // tile 5: lanes of rows 84.. -> trash rows
    // the three col2im items of this thread: kernel row u and quad index Qi inside the chunk's 66
    const int it_u2 = wave == 0 ? 8 : (lane >> 1), it_q2 = wave == 0 ? lane : 64 + (lane & 1);
    const int it_on2 = (int)(wave == 0) | ((int)(wave == 1) & (int)(lane < 18));

    const int img_bytes = h * w * EAE_C * 4;
    const int pix = 16 * h * w;                                                   // output pixels per image
    unsigned long long se = 0;
    int se_img = -1;

    // ---- chunk walk -----------------------------------------------------------------------------------------------------------
    int g = gs, k = kbeg;
    int img = gs / h, p = gs - img * h;
    int s0 = (4 * p) % SLOTS;
    float4 xb[8];
#define EAE_T3_FETCH(img_, p_, k_, valid_)                                                                            \
    {                                                                                                                 \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                         \
            const_cast<float*>(x + (size_t)(img_) * h * w * EAE_C), 0, img_bytes, 0x00020000);                        \
        const int site_ = (k_) * CHUNK + 16 * wave + n16;                                                             \
        const int off_ = ((valid_) && site_ < w) ? ((p_) * w + (k_) * CHUNK) * (EAE_C * 4) + x_lane : (int)0x80000000; \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                               \
            const u32x4 t_ = __builtin_amdgcn_raw_buffer_load_b128(rs_, off_ + 64 * j, 0, 0);                         \
            xb[j] = make_float4(__uint_as_float(t_.x), __uint_as_float(t_.y), __uint_as_float(t_.z),                  \
                                __uint_as_float(t_.w));                                                               \
        }                                                                                                             \
    }
    // lane (kq, site) holds channels 16 j + 4 kq + e of its site in component e of load j; the MFMA wants channel 4 s + kq in the
    // register of k-step s: a 4 x 4 transpose between e and kq, per load
#define EAE_T3_TRANSPOSE(dst_)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                   \
        float r0 = xb[j].x, r1 = xb[j].y, r2 = xb[j].z, r3 = xb[j].w;                                                 \
        swap16(r0, r1); swap16(r2, r3); swap_halves(r0, r2); swap_halves(r1, r3);                                     \
        dst_[4 * j + 0] = r0; dst_[4 * j + 1] = r1; dst_[4 * j + 2] = r2; dst_[4 * j + 3] = r3;                        \
    }
    EAE_T3_FETCH(img, p, k, true)
    float xs_a[32], xs_b[32];
    EAE_T3_TRANSPOSE(xs_a)
    __syncthreads();                                        // LDS zeroed

#ifdef EAE_T3_TRACE
    tr_acc[0] = __builtin_amdgcn_s_memtime() - ts_begin;
#endif
    Gather prev;
    prev.on = 0; prev.img = 0; prev.p = 0; prev.c0 = 0; prev.ulo = 0; prev.uhi = -1; prev.ufin = -1; prev.edge_final = 0; prev.s0 = 0;
    prev.buf = 0;
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f}, c5 = {0.f, 0.f, 0.f, 0.f};      // the last two tiles of a chunk are written a chunk later

    // One quad of one kernel row of the previous chunk (see the head of the file), in three stages that the chunk body places a third
    // of a chunk's MFMAs apart, so that no wait of this one-wave-per-SIMD stream ever finds its data still on the way:
    //   prepare: where the quad lives, what happens to it (masks), its reference pixels requested from HBM
    //   read:    nine parts and the running sum requested from LDS
    //   finish:  the additions in site order, the cast, the stores, the squared error, the running sum written back
    // Straight-line on purpose (bitwise masks instead of && / ?: around loads): a branch would cut the block the scheduler spreads
    // under the MFMAs. `u` is wave-uniform for the first two items of a thread and per lane for the third.
    struct Item {
        const float* pu;
        float4* ap;
        unsigned int keep, m_st;
        int o1, o4;
        unsigned int rv;
        float a6, a7, a8, b2, b3, b4, b5, d0, d1;
        float4 a;
    };
    Item items[3];
    auto item_prepare = [&](Item& it, const Gather& gp, const int u, const int Qi, const int on) {
        const int rank = u >= 6 ? u - 6 : (u >= 2 ? u + 1 : u + 7);
        it.pu = P0 + gp.buf * P_FLOATS + (9 * rank + (rank >= 7 ? 1 : 0)) * PSITES + Qi;
        const int Q = gp.c0 - 1 + Qi;
        const int active = on & gp.on & (int)(u >= gp.ulo) & (int)(u <= gp.uhi) & (int)(Q >= qown0) & (int)(Q < qown1);
        const int m_act = -active;                                                 // all ones / zero
        int slot = gp.s0 + u;
        slot = slot >= SLOTS ? slot - SLOTS : slot;
        const int arow = SLOTS + ((slot - SLOTS) & m_act), aq = (Qi & 63) ^ (((Qi & 63) ^ (Q - qown0)) & m_act);
        it.ap = ACC + arow * acc_row + aq;
        const int fin = active & (int)(u <= gp.ufin) & ((int)(Qi < CHUNK) | gp.edge_final);
        it.keep = (unsigned int)(fin - 1);                                         // zero when the quad is complete
        const int st = fin & (int)(Q < w);
        it.m_st = (unsigned int)(-st);
        const int o = (4 * gp.p + u - 2) * (4 * w) + 4 * Q;                        // first pixel of the quad inside its image
        const int oob = (int)0x80000000;
        it.o1 = oob ^ ((oob ^ o) & (int)it.m_st);
        it.o4 = oob ^ ((oob ^ (4 * o)) & (int)it.m_st);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(ref) + (size_t)gp.img * pix, 0, ref ? pix : 0, 0x00020000);
        it.rv = __builtin_amdgcn_raw_buffer_load_b32(rs_r, it.o1, 0, 0);           // out of range reads 0
    };
    // Relaxed wavefront-scope atomic loads (plain ds_read_b32, no wait, no fence): these loads stay HERE, a third of a chunk ahead of
    // their use. Plain loads are sunk to just in front of the first addition, scheduling fences or not, and the wave then waits for
    // LDS inside the stretch that was meant to hide it; volatile ones are each followed by a wait.
    auto lds_get = [](const float* p) -> float {
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
    };
    auto item_read = [&](Item& it) {
        const float* const pu = it.pu;
        it.a6 = lds_get(pu + 6 * PSITES); it.a7 = lds_get(pu + 7 * PSITES); it.a8 = lds_get(pu + 8 * PSITES);          // site Q - 1: v = 6, 7, 8
        it.b2 = lds_get(pu + 2 * PSITES + 1); it.b3 = lds_get(pu + 3 * PSITES + 1);                                  // site Q: v = 2 .. 5
        it.b4 = lds_get(pu + 4 * PSITES + 1); it.b5 = lds_get(pu + 5 * PSITES + 1);
        it.d0 = lds_get(pu + 2); it.d1 = lds_get(pu + PSITES + 2);                                                   // site Q + 1: v = 0, 1
        const float* const pa = reinterpret_cast<const float*>(it.ap);
        it.a.x = lds_get(pa); it.a.y = lds_get(pa + 1); it.a.z = lds_get(pa + 2); it.a.w = lds_get(pa + 3);
    };
    auto item_finish = [&](const Item& it, const Gather& gp) {
        const float px0 = (it.a.x + it.a6) + it.b2;
        const float px1 = (it.a.y + it.a7) + it.b3;
        const float px2 = ((it.a.z + it.a8) + it.b4) + it.d0;
        const float px3 = (it.a.w + it.b5) + it.d1;
        u32x4 fv;
        fv.x = __float_as_uint(px0); fv.y = __float_as_uint(px1); fv.z = __float_as_uint(px2); fv.w = __float_as_uint(px3);
        u32x4 back;
        back.x = fv.x & it.keep; back.y = fv.y & it.keep; back.z = fv.z & it.keep; back.w = fv.w & it.keep;
        *reinterpret_cast<u32x4*>(it.ap) = back;
        const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(
            out_f32 + (size_t)gp.img * pix, 0, out_f32 ? pix * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(out_u8 + (size_t)gp.img * pix, 0, out_u8 ? pix : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(fv, rs_f, it.o4, 0, 0);
        // tls.cast_bt601: clip to [16, 235], round half to even, uint8
        const unsigned int q0 = (unsigned int)round_half_even(fminf(fmaxf(px0, 16.f), 235.f));
        const unsigned int q1 = (unsigned int)round_half_even(fminf(fmaxf(px1, 16.f), 235.f));
        const unsigned int q2 = (unsigned int)round_half_even(fminf(fmaxf(px2, 16.f), 235.f));
        const unsigned int q3 = (unsigned int)round_half_even(fminf(fmaxf(px3, 16.f), 235.f));
        __builtin_amdgcn_raw_buffer_store_b32(q0 | (q1 << 8) | (q2 << 16) | (q3 << 24), rs_u, it.o1, 0, 0);
        const unsigned int rv = it.rv;
        const int e0 = (int)(rv & 0xFF) - (int)q0, e1 = (int)((rv >> 8) & 0xFF) - (int)q1;
        const int e2 = (int)((rv >> 16) & 0xFF) - (int)q2, e3 = (int)(rv >> 24) - (int)q3;
        se += (unsigned int)(e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3) & it.m_st;
    };
    // the squared errors of one image leave the block: one atomic per wave
    auto flush = [&]() {
        if (ref && sse && se_img >= 0) {
            unsigned long long t = se;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t != 0) atomicAdd(&sse[se_img], t);
        }
        se = 0;
    };
#define EAE_T3_PHASE(t0_, ca_, cb_, xs_)                                                                              \
    _Pragma("unroll") for (int s = 0; s < 32; ++s) {                                                                  \
        ca_ = mfma16(wreg[t0_][s], xs_[s], ca_);                                                                      \
        cb_ = mfma16(wreg[t0_ + 1][s], xs_[s], cb_);                                                                  \
    }
#define EAE_T3_WRITE(buf_, t_, c_)                                                                                    \
    {                                                                                                                 \
        char* const d_ = reinterpret_cast<char*>(P0 + (buf_) * P_FLOATS) + pw + 16 * (t_) * PSITES * 4;               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<float*>(d_ + i * PSITES * 4) = c_[i];         \
    }
#define EAE_T3_WRITE5(buf_, c_)                                                                                       \
    {                                                                                                                 \
        char* const d_ = reinterpret_cast<char*>(P0 + (buf_) * P_FLOATS) + pw5;                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) *reinterpret_cast<float*>(d_ + i * PSITES * 4) = c_[i];         \
    }
    // One chunk. KIND says which tap tiles the row needs; xs_cur holds the chunk's sites, xs_nxt receives the next chunk's.
    auto chunk = [&](auto kind_c, float (&xs_cur)[32], float (&xs_nxt)[32], const int ci) {
        constexpr int KIND = decltype(kind_c)::value;
        // the chunk behind this one (its sites go on their way now) and what the col2im of this one will need
        int gn = g, kn = k + 1, pn = p, imgn = img, s0n = s0;
        if (kn == kend) {
            kn = kbeg; ++gn; ++pn; s0n += 4;
            s0n = s0n >= SLOTS ? s0n - SLOTS : s0n;
            if (pn == h) { pn = 0; ++imgn; s0n = 0; }
        }
        Gather cur;
        cur.on = 1; cur.img = img; cur.p = p; cur.c0 = k * CHUNK; cur.s0 = s0; cur.buf = ci & 1;
        cur.edge_final = k + 1 == kend;
        if (KIND == ROW_TOP) { cur.ulo = 6; cur.uhi = 8; cur.ufin = -1; }
        else if (KIND == ROW_BOTTOM) { cur.ulo = 0; cur.uhi = 1; cur.ufin = 1; }
        else {
            cur.ulo = (g == g0 || p == 0) ? 2 : 0;
            cur.uhi = (g == g1 - 1 || p == h - 1) ? 5 : 8;
            cur.ufin = p == h - 1 ? 5 : 3;
        }
        if (prev.on && prev.img != se_img) { flush(); se_img = prev.img; }
        T3_STAMP(0)
        // ---- what is requested from HBM now and used two thirds of a chunk later: the previous chunk's reference pixels, the next
        //      chunk's sites; the previous chunk's last two tiles go to LDS ---------------------------------------------------------
        item_prepare(items[0], prev, wave, lane, 1);
        item_prepare(items[1], prev, 4 + wave, lane, 1);
        item_prepare(items[2], prev, it_u2, it_q2, it_on2);
        EAE_T3_FETCH(imgn, pn, kn, ci + 1 < n_chunks)
        EAE_T3_WRITE(prev.buf, 4, c4)
        EAE_T3_WRITE5(prev.buf, c5)
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f}, c3 = {0.f, 0.f, 0.f, 0.f};
        // ---- first third: tiles 0-1 (with the above spread under them), then the barrier: the previous chunk's parts are complete, its
        //      col2im may read them --------------------------------------------------------------------------------------------------------
        if (KIND != ROW_BOTTOM) {
            EAE_T3_PHASE(0, c0, c1, xs_cur)
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                 // three vector instructions
                if (i % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // a load
                if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // an LDS write
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        T3_STAMP(2)
        if (KIND != ROW_BOTTOM) { EAE_T3_WRITE(cur.buf, 0, c0) EAE_T3_WRITE(cur.buf, 1, c1) }
        item_read(items[0]);
        item_read(items[1]);
        item_read(items[2]);
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(3)
        // ---- second third: tiles 2-3, nothing else: the LDS and HBM round trips above run out under it -----------------------------------
        if (KIND == ROW_BODY) EAE_T3_PHASE(2, c2, c3, xs_cur)
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(4)
        if (KIND == ROW_BODY) { EAE_T3_WRITE(cur.buf, 2, c2) EAE_T3_WRITE(cur.buf, 3, c3) }
        // ---- last third: tiles 4-5 beside the arithmetic of the col2im and the next chunk's transpose ----------------------------------
        c4 = f32x4{0.f, 0.f, 0.f, 0.f}; c5 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (KIND != ROW_TOP) EAE_T3_PHASE(4, c4, c5, xs_cur)
        item_finish(items[0], prev);
        item_finish(items[1], prev);
        item_finish(items[2], prev);
        EAE_T3_TRANSPOSE(xs_nxt)
        if (KIND != ROW_TOP) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                 // four vector instructions
                if (i % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // an LDS write
                if (i % 8 == 7) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); // a store
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        T3_STAMP(5)
        T3_ADD(1, 0, 1) T3_ADD(2, 1, 2) T3_ADD(3, 2, 3) T3_ADD(4, 3, 4) T3_ADD(5, 4, 5)
#ifdef EAE_T3_TRACE
        tr_acc[6] += 1;
#endif
        prev = cur;
        g = gn; k = kn; p = pn; img = imgn; s0 = s0n;
    };
    auto dispatch = [&](float (&xs_cur)[32], float (&xs_nxt)[32], const int ci) {
        const int kind = g < g0 ? ROW_TOP : (g >= g1 ? ROW_BOTTOM : ROW_BODY);
        if (kind == ROW_BODY) chunk(std::integral_constant<int, ROW_BODY>{}, xs_cur, xs_nxt, ci);
        else if (kind == ROW_TOP) chunk(std::integral_constant<int, ROW_TOP>{}, xs_cur, xs_nxt, ci);
        else chunk(std::integral_constant<int, ROW_BOTTOM>{}, xs_cur, xs_nxt, ci);
    };
    for (int ci = 0; ci < n_chunks; ci += 2) {
        dispatch(xs_a, xs_b, ci);
        if (ci + 1 < n_chunks) dispatch(xs_b, xs_a, ci + 1);
    }
    // ---- the last chunk's col2im --------------------------------------------------------------------------------------------------
    EAE_T3_WRITE(prev.buf, 4, c4)
    EAE_T3_WRITE5(prev.buf, c5)
    __syncthreads();
    if (prev.img != se_img) { flush(); se_img = prev.img; }
    item_prepare(items[0], prev, wave, lane, 1);
    item_prepare(items[1], prev, 4 + wave, lane, 1);
    item_prepare(items[2], prev, it_u2, it_q2, it_on2);
#pragma unroll
    for (int i = 0; i < 3; ++i) item_read(items[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) item_finish(items[i], prev);
    flush();
#ifdef EAE_T3_TRACE
    tr_acc[7] = __builtin_amdgcn_s_memtime() - ts_begin;
    if (lane == 0 && wave == 0 && sse) {
        for (int i = 0; i < 8; ++i) atomicAdd(&sse[64 + i], (unsigned long long)tr_acc[i]);
        atomicMax(&sse[72], (unsigned long long)tr_acc[7]);
    }
#endif
#undef EAE_T3_FETCH
#undef EAE_T3_TRANSPOSE
#undef EAE_T3_PHASE
#undef EAE_T3_WRITE
#undef EAE_T3_WRITE5
}

// TF filter [9][9][1][128] -> the A fragments of the 6 tap tiles: [tile][k-step / 4][lane][4]; lane = kq * 16 + m holds
// W[position 16 tile + m][channel 4 s + kq] for the four k-steps s of its group; zero at the positions no tap has.
__global__ void pack_tconv3_kernel(const float* __restrict__ w_tf, float* __restrict__ wq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_FLOATS) return;
    const int e = i & 3, lane = (i >> 2) & 63, s4 = (i >> 8) & 7, t = i >> 11;
    const int pos = 16 * t + (lane & 15), ci = 4 * (4 * s4 + e) + (lane >> 4);
    float v = 0.f;
    for (int u = 0; u < 9; ++u)
        for (int vv = 0; vv < 9; ++vv)
            if (tap_pos(u, vv) == pos) v = w_tf[(u * 9 + vv) * EAE_C + ci];
    wq[i] = v;
}
}  // namespace

extern "C" int eae_hip_pack_tconv9x9s4_weights(const float* w_tf, float* w_phase, void* stream) {
    if (!w_tf || !w_phase) return EAE_HIP_BAD_ARGUMENT;
    static_assert(W_FLOATS == EAE_HIP_TCONV9X9S4_PACKED_FLOATS, "packed filter size");
    hipLaunchKernelGGL(pack_tconv3_kernel, dim3((W_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_tf, w_phase);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_tconv9x9s4_luma(const float* x, const float* w_phase, float* out_f32, uint8_t* out_u8,
                                       const uint8_t* ref_u8, uint64_t* sse, int n, int h, int w_in, void* stream) {
    if (!x || !w_phase || n <= 0 || h <= 0 || w_in <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (!out_f32 && !out_u8 && !ref_u8) return EAE_HIP_BAD_ARGUMENT;
    if ((ref_u8 != nullptr) != (sse != nullptr)) return EAE_HIP_BAD_ARGUMENT;
    if ((long)h * w_in * EAE_C * (long)sizeof(float) > 0x7FFFFFFFL) return EAE_HIP_BAD_SHAPE;       // 32-bit offsets inside an image
    if ((long)n * h > 0x3FFFFFFFL) return EAE_HIP_BAD_ARGUMENT;
    const int cus = eae_compute_units();
    if (cus <= 0) return (int)hipErrorInvalidDevice;
    // segments of at most MAX_SEG_CHUNKS chunks; strips of whole site rows, one block per CU where the batch has that many rows
    const int chunks_total = (w_in + CHUNK - 1) / CHUNK;
    const int n_seg = (chunks_total + MAX_SEG_CHUNKS - 1) / MAX_SEG_CHUNKS;
    const int seg_chunks = (chunks_total + n_seg - 1) / n_seg;
    const long rows = (long)n * h;
    long row_strips = cus / n_seg > 0 ? cus / n_seg : 1;
    if (g_eae_launch_options.t3_strips_per_cu > 1) row_strips *= g_eae_launch_options.t3_strips_per_cu;
    if (row_strips > rows) row_strips = rows;
    const int rows_per_strip = (int)((rows + row_strips - 1) / row_strips);
    row_strips = (rows + rows_per_strip - 1) / rows_per_strip;
    const size_t lds_bytes = ((size_t)(SLOTS + 1) * seg_chunks * CHUNK * 4 + 2 * P_FLOATS) * sizeof(float);
    static bool attr_set[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(tconv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(((size_t)(SLOTS + 1) * MAX_SEG_CHUNKS * CHUNK * 4 + 2 * P_FLOATS) * sizeof(float)));
        if (err != hipSuccess) return (int)err;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(tconv3_kernel, dim3((unsigned)(row_strips * n_seg)), dim3(NT), lds_bytes, (hipStream_t)stream, x, w_phase, out_f32,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), n, h, w_in, rows_per_strip, (int)row_strips, seg_chunks);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
