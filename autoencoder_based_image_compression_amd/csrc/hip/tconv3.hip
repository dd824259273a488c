// tconv3.hip -- transpose_conv_3 (eae/graph/components.py:79-83: tf.nn.conv2d_transpose 9x9, 128 -> 1 channel,
// stride 4, 'SAME', no bias) fused with what follows it on the path: tls.cast_bt601 (tools/tools.py:93) and the
// squared error of tls.psnr_2d (tools.py:873-875).
//
// Formulation (round 6): the deconvolution as a dense GEMM followed by col2im.
//     part[site][tap (u, v)] = one fmaf chain over the 128 channels of X[site][.] * W[u][v][.], started from +0, the channels of
//                              each block of 16 in the order 0 4 8 12 1 5 9 13 2 6 10 14 3 7 11 15 (below)      81 taps per site
//     out[4p'+a][4q'+b]      = the <= 9 parts with u = a + 2 - 4 dr, v = b + 2 - 4 dc of the sites (p'+dr, q'+dc), added
//                              site by site in raster order (dr ascending, then dc ascending), starting from +0
// which is oracle/transforms_oracle.c: orc_conv2d_transpose_same_col2im, bit for bit. Rounds 1-5 ran every output pixel as ONE
// fmaf chain over (channel block, u, v, channel): as a matrix product that is N = 16 output phases per neighbour site, of
// which 81 / 144 exist -- 7 of every 16 products were structural zeros, and the kernel sat at 0.40 of the f32 MFMA peak for four
// rounds. Here the GEMM is [81 taps, padded to 96] x [128] x [sites]: 81 / 96 live, every site read ONCE (no halo in the
// operand), and the overlap-add is ~5 float additions per output pixel.
//
// What shapes everything below: on a SIMD the f32 MFMA and ordinary vector instructions do not overlap (they add up, whichever
// wave issues them: DESIGN.md section 4), so beside the 192 MFMAs of a chunk every vector instruction costs its four cycles. Hence:
//   * the sites never pass through a vector instruction: lane (kq, site) fetches eight 16-byte pieces of its site (channels
//     16 j + 4 kq + e) from HBM STRAIGHT INTO ACCUMULATOR REGISTERS, which v_mfma_f32_16x16x4_f32 takes as its B operand. k-step
//     s = 4 j + e therefore sums channels 16 j + e, + 4, + 8, + 12 in that order: the channel order stated above (a legitimate fixed
//     order like any other; the oracle runs the same);
//   * the whole filter sits in registers for the life of the block (6 tap tiles x 32 k-steps = 192 per lane, a third of them in
//     accumulator registers, which the MFMA also takes as A); loads, MFMAs and their waits are inline assembly: hipcc only has
//     vector registers for MFMA operands and copies;
//   * col2im decides per WAVE, on scalars, what a kernel row of a chunk needs (nothing / add into the running sum / add, cast and
//     store), instead of per lane with masks.
//
// One block = 4 waves = one per SIMD, persistent over a STRIP of consecutive site rows (all images of the batch stacked), full
// width (segments of <= 512 sites for wider images). A chunk = 64 consecutive sites of a row, 16 per wave; per chunk and wave
// 192 MFMAs, two tap tiles interleaved at a time (their accumulators alternate: no dependent-issue stall), the parts written
// tap-major into one of two LDS buffers ([position][2 sites of the previous chunk | 64 sites | a zero site]). The col2im of a chunk
// is pipelined over the next two chunks: behind the barrier that completes the chunk's parts, a thread requests what its quads
// need -- a quad = four output pixels of one kernel row u: nine parts of three neighbouring sites (consecutive lanes =
// consecutive sites: conflict-free) plus the quad's running sum, a rolling accumulator of nine pixel rows in LDS ((4 p + u) mod 9),
// plus its reference pixels from HBM -- and a whole chunk of MFMAs later adds them in site order and either writes the sum back
// or, when no later site row contributes (u <= 3), casts to BT.601 uint8, stores four pixels as one dword, adds the squared error
// and returns the slot as zeros. One barrier per chunk, placed behind the first third of the NEXT chunk's MFMAs.
// Site rows at a strip's ends that belong to a neighbour strip are recomputed (only the tap tiles that reach across: u >= 6 of
// the row above = tiles 0-1, u <= 1 of the row below = tiles 4-5: the tap positions are ordered for that): 2/3 of a row per strip.
// Bound: MFMA (1,296 algorithmic / 1,536 issued FLOP per pixel); 32 B/px read + 1 B/px written is the HBM term.
#include "common.h"

#include <type_traits>

namespace {
constexpr int NT = 256;                        // 4 waves, one per SIMD (the filter in registers needs the whole register file)
constexpr int CHUNK = 64;                      // sites per chunk, 16 per wave
constexpr int PSITES = CHUNK + 4;              // a row of the part buffer: the 2 sites before the chunk | 64 sites | a zero site | pad
constexpr int NPOS = 84;                       // rows of the part buffer: 82 tap positions + 2 trash rows for tile 5's idle lanes
constexpr int P_FLOATS = NPOS * PSITES;        // 22,848 B per buffer, two buffers
constexpr int SLOTS = 9;                       // rolling pixel rows; row 9 of the accumulator is the trash row
constexpr int MAX_SEG_CHUNKS = 8;              // a segment is at most 512 sites wide (the accumulator rows must fit in LDS)
constexpr int W_FLOATS = 6 * 32 * 64;          // packed filter: [tile][k-step][lane]
constexpr int OOB = (int)0x80000000;           // a buffer offset beyond every image: loads give 0, stores go nowhere

// Position of tap (u, v) in the GEMM's M dimension. Kernel rows in the order 6 7 8 | 2 3 4 5 | (one unused position) 0 1, so that
// the rows a strip's upper neighbour needs (u >= 6) are tiles 0-1 and those its lower neighbour needs (u <= 1) are tiles 4-5.
__host__ __device__ constexpr int tap_rank(int u) { return u >= 6 ? u - 6 : (u >= 2 ? u + 1 : u + 7); }
__host__ __device__ constexpr int tap_pos(int u, int v) { return 9 * tap_rank(u) + v + (tap_rank(u) >= 7 ? 1 : 0); }
static_assert(tap_pos(8, 8) == 26 && tap_pos(5, 8) == 62 && tap_pos(0, 0) == 64 && tap_pos(1, 8) == 81, "tap positions");
// Channel that k-step s takes from lane row kq (see the head of the file).
__host__ __device__ constexpr int step_channel(int s, int kq) { return 16 * (s >> 2) + 4 * kq + (s & 3); }

enum { ROW_TOP = 0, ROW_BODY = 1, ROW_BOTTOM = 2 };

#ifdef EAE_T3_TRACE                   // scratch/r06/t3_trace.py: shader clock ticks per stretch of a chunk, summed behind the squared errors
#define T3_STAMP(i_) const long long ts##i_ = __builtin_amdgcn_s_memtime();
#define T3_ADD(i_, a_, b_) tr_acc[i_] += ts##b_ - ts##a_;
#else
#define T3_STAMP(i_)
#define T3_ADD(i_, a_, b_)
#endif

// what the col2im of a chunk needs to know (all wave-uniform; packed: scalar registers are scarce here)
struct Gather {
    int img, p, c0;         // image, site row, first site of the chunk
    int bits;               // on | last << 1 | own << 2 | ulo << 4 | uhi << 8 | (ufin + 1) << 12 | s0 << 16
    __device__ __forceinline__ int on() const { return bits & 1; }          // 0: no such chunk (the first iterations)
    __device__ __forceinline__ int last() const { return (bits >> 1) & 1; } // last chunk of its row: its last quad has no chunk behind it to be finished in
    __device__ __forceinline__ int own() const { return (bits >> 2) & 1; }  // a chunk with quads of this block (not the extra chunk left of a segment)
    __device__ __forceinline__ int ulo() const { return (bits >> 4) & 15; } // kernel rows taken: ulo .. uhi
    __device__ __forceinline__ int uhi() const { return (bits >> 8) & 15; }
    __device__ __forceinline__ int ufin() const { return ((bits >> 12) & 15) - 1; }   // rows <= ufin are complete behind this site row
    __device__ __forceinline__ int s0() const { return (bits >> 16) & 15; } // (4 p) mod 9
    static __device__ __forceinline__ int pack(int on, int last, int own, int ulo, int uhi, int ufin, int s0) {
        return on | last << 1 | own << 2 | ulo << 4 | uhi << 8 | (ufin + 1) << 12 | s0 << 16;
    }
};

// one quad of one kernel row, between request and use
struct Item {
    float4* ap;             // the quad's running sum
    int o1;                 // byte offset of its four pixels inside the image (OOB: not stored)
    unsigned int rv;        // its four reference pixels
    float a6, a7, a8, b2, b3, b4, b5, d0, d1;
    float a0, a1, a2, a3;   // the running sum
};

// ---- the instructions hipcc cannot be talked into ------------------------------------------------------------------------------
// one float of the filter straight into an accumulator register (at the head of the kernel, followed by one wait for all of them)
#define T3_ASM_WLOAD(dst_, voff_, rsrc_, soff_, imm_)                                                                 \
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=a"(dst_) : "v"(voff_), "s"(rsrc_), "s"(soff_), "i"(imm_))
// D = A x B (+ C): the filter from an accumulator register, the sites from a vector register, the sums in accumulator registers
#define T3_ASM_MFMA0(c_, w_, x_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(c_) : "a"(w_), "v"(x_))
#define T3_ASM_MFMA(c_, w_, x_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c_) : "a"(w_), "v"(x_))
// the matrix unit's result must be in the register file before anything but another MFMA reads it (8 passes: 11 wait states)
#define T3_ASM_SETTLE(ca_, cb_) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(ca_), "+a"(cb_))

__global__ __launch_bounds__(NT, 1) void tconv3_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                        float* __restrict__ out_f32, uint8_t* __restrict__ out_u8,
                                                        const uint8_t* __restrict__ ref, unsigned long long* sse,
                                                        int n, int h, int w, int rows_per_strip, int n_row_strips, int seg_chunks, int acc_rs) {
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
    const int qown0 = kown0 * CHUNK, own_quads = (kown1 - kown0) * CHUNK;                           // the quads this block owns
    float4* const ACC = reinterpret_cast<float4*>(lds);
    // accumulator rows of acc_rs quads: one leading column (the quad left of the image), the quads of the segment, and where the
    // image has several segments the 63 quads of the extra chunk on the right that are not this block's
    float* const P0 = lds + (SLOTS + 1) * acc_rs * 4;
    const int p_first = g0 % h, p_last = (g1 - 1) % h;
    const int has_top = p_first > 0, has_bot = p_last < h - 1;
    const int gs = g0 - has_top, ge = g1 + has_bot;
    const int n_chunks = (ge - gs) * (kend - kbeg);

#ifdef EAE_T3_TRACE
    long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long ts_begin = __builtin_amdgcn_s_memtime();
#endif
    // ---- the filter: 6 tiles x 32 k-steps, one ACCUMULATOR register each, for the life of the block (the MFMA takes its A operand
    //      from either file; the vector file is left to the sites and the col2im). Loaded float by float: a 16-byte load would give
    //      a register tuple, whose elements hipcc hands to an asm operand through copies. ------------------------------------------------
    float wr[6][32];
    {
        const unsigned long long base = reinterpret_cast<unsigned long long>(wq);
        u32x4 rs;
        rs.x = __builtin_amdgcn_readfirstlane((unsigned int)base);
        rs.y = __builtin_amdgcn_readfirstlane((unsigned int)(base >> 32) & 0xFFFFu);
        rs.z = (unsigned int)(W_FLOATS * sizeof(float));
        rs.w = 0x00020000u;
        const int voff = lane * 4;
#define T3_W16(t_, s16_)                                                                                              \
        {                                                                                                             \
            const int so_ = ((t_) * 32 + (s16_) * 16) * 256;     /* [tile][k-step][lane]: 256 bytes per (tile, k-step) */  \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 0], voff, rs, so_, 0);     T3_ASM_WLOAD(wr[t_][16 * (s16_) + 1], voff, rs, so_, 256);   \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 2], voff, rs, so_, 512);   T3_ASM_WLOAD(wr[t_][16 * (s16_) + 3], voff, rs, so_, 768);   \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 4], voff, rs, so_, 1024);  T3_ASM_WLOAD(wr[t_][16 * (s16_) + 5], voff, rs, so_, 1280);  \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 6], voff, rs, so_, 1536);  T3_ASM_WLOAD(wr[t_][16 * (s16_) + 7], voff, rs, so_, 1792);  \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 8], voff, rs, so_, 2048);  T3_ASM_WLOAD(wr[t_][16 * (s16_) + 9], voff, rs, so_, 2304);  \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 10], voff, rs, so_, 2560); T3_ASM_WLOAD(wr[t_][16 * (s16_) + 11], voff, rs, so_, 2816); \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 12], voff, rs, so_, 3072); T3_ASM_WLOAD(wr[t_][16 * (s16_) + 13], voff, rs, so_, 3328); \
            T3_ASM_WLOAD(wr[t_][16 * (s16_) + 14], voff, rs, so_, 3584); T3_ASM_WLOAD(wr[t_][16 * (s16_) + 15], voff, rs, so_, 3840); \
        }
        T3_W16(0, 0) T3_W16(0, 1) T3_W16(1, 0) T3_W16(1, 1) T3_W16(2, 0) T3_W16(2, 1)
        T3_W16(3, 0) T3_W16(3, 1) T3_W16(4, 0) T3_W16(4, 1) T3_W16(5, 0) T3_W16(5, 1)
#undef T3_W16
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- LDS: accumulator rows and both part buffers start as zeros (the zero site of a part buffer stays zero) --------------------
    {
        const int total4 = ((SLOTS + 1) * acc_rs * 4 + 2 * P_FLOATS) / 4;
        for (int i = tid; i < total4; i += NT) ACC[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- per-thread constants ---------------------------------------------------------------------------------------------------
    const int x_lane = ((16 * wave + n16) * EAE_C + 4 * kq) * 4;                  // bytes, inside a chunk
    // the ragged last chunk of an image row: sites beyond the image read as zeros
    const int x_lane_last = (chunks_total - 1) * CHUNK + 16 * wave + n16 < w ? x_lane : OOB;
    const int pw = (4 * kq * PSITES + 2 + 16 * wave + n16) * 4;                   // bytes: this lane's first part of a tile
    const int pw5 = kq == 0 ? pw + 80 * PSITES * 4 : (82 * PSITES + 2 + 16 * wave + n16) * 4;   // tile 5: lanes of rows 84.. -> trash rows
    const int img_bytes = h * w * EAE_C * 4;
    const int pix = 16 * h * w;                                                   // output pixels per image
    unsigned long long se = 0;
    int se_img = -1;
    // col2im: this wave takes the kernel rows uA = wave and uB = 4 + wave of every chunk (quad = lane), wave 0 also row 8; wave 1 the
    // last quad of a row (kernel row = lane); wave 2 carries the last two sites of a chunk over into the next chunk's buffer
    auto row_base = [](const int u) { const int r = tap_rank(u); return (9 * r + (r >= 7 ? 1 : 0)) * PSITES; };
    const float* const puA = P0 + row_base(wave) + lane;
    const float* const puB = P0 + row_base(4 + wave) + lane;
    const float* const puC = P0 + row_base(8) + lane;
    const int uT = lane < 9 ? lane : 8;
    const float* const puT = P0 + row_base(uT) + CHUNK;
    const int ctx_e0 = lane, ctx_e1 = lane + 64, ctx_e2 = lane + 128 < 2 * 82 ? lane + 128 : 2 * 82 - 1;
    const int ctx_src0 = (ctx_e0 >> 1) * PSITES + CHUNK + (ctx_e0 & 1), ctx_dst0 = (ctx_e0 >> 1) * PSITES + (ctx_e0 & 1);
    const int ctx_src1 = (ctx_e1 >> 1) * PSITES + CHUNK + (ctx_e1 & 1), ctx_dst1 = (ctx_e1 >> 1) * PSITES + (ctx_e1 & 1);
    const int ctx_src2 = (ctx_e2 >> 1) * PSITES + CHUNK + (ctx_e2 & 1), ctx_dst2 = (ctx_e2 >> 1) * PSITES + (ctx_e2 & 1);

    // ---- the sites of a chunk: eight 16-byte pieces per lane, used by the MFMAs as they come (no vector instruction touches them) ----
    float4 xa[2][8];
    __amdgpu_buffer_rsrc_t x_rs;
    int x_soff, x_voff;
    auto fetch_setup = [&](const int img_, const int p_, const int k_, const bool valid_) {
        x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)img_ * h * w * EAE_C), 0, valid_ ? img_bytes : 0, 0x00020000);
        x_soff = (p_ * w + k_ * CHUNK) * (EAE_C * 4);
        x_voff = k_ == chunks_total - 1 ? x_lane_last : x_lane;
    };
    auto fetch_one = [&](float4& dst, const int j) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(x_rs, x_voff + 64 * j, x_soff, 0);
        dst = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    };

    // ---- chunk walk -----------------------------------------------------------------------------------------------------------
    int g = gs, k = kbeg;
    int img = gs / h, p = gs - img * h;
    int s0 = (4 * p) % SLOTS;
    fetch_setup(img, p, k, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) fetch_one(xa[0][j], j);
    __syncthreads();                                        // LDS zeroed

#ifdef EAE_T3_TRACE
    tr_acc[0] = __builtin_amdgcn_s_memtime() - ts_begin;
#endif
    Gather g1d, g2d;                 // the chunk behind this one (its col2im is requested in this iteration) and the one behind that (finished)
    g1d.img = 0; g1d.p = 0; g1d.c0 = 0; g1d.bits = 0;
    g2d = g1d;
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f}, c5 = {0.f, 0.f, 0.f, 0.f};      // the last two tiles of a chunk are written a chunk later
    Item itA, itB, itC, itT;         // kernel rows wave, 4 + wave, 8 (wave 0), and the row's last quad (wave 1, per lane)
    itA.ap = itB.ap = itC.ap = itT.ap = ACC + SLOTS * acc_rs + 1 + lane;
    itA.o1 = itB.o1 = itC.o1 = itT.o1 = OOB;
    float ctx0 = 0.f, ctx1 = 0.f, ctx2 = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // Relaxed wavefront-scope atomic loads (plain ds_read_b32, no wait, no fence): memory instructions keep their place between the
    // MFMAs they are written between (inline assembly with side effects orders them), which is the point: LDS, HBM and scalar
    // instructions issue in the shadow of this wave's own MFMAs (32.0 -> 32.8 cycles per MFMA with two LDS reads behind each,
    // scratch/r06/probe_shadow.hip), vector instructions do not (+ 4 cycles each, + 8 per switch between the two kinds).
    auto lds_get = [](const float* q) -> float {
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
    };
    // which kernel rows of a chunk this wave works on, and how (wave-uniform)
    auto acts = [](const Gather& gp, const int u) { return gp.on() && u >= gp.ulo() && u <= gp.uhi(); };
    auto fins = [](const Gather& gp, const int u) { return gp.on() && u >= gp.ulo() && u <= gp.uhi() && u <= gp.ufin(); };
    // the running sum of quad Qi of kernel row u of chunk gp: its slot row, or the trash row
    auto acc_of = [&](const Gather& gp, const int u, const bool act) -> float4* {
        int slot = gp.s0() + u;
        slot = slot >= SLOTS ? slot - SLOTS : slot;
        return ACC + ((act && gp.own()) ? slot * acc_rs + (gp.c0 - qown0) : SLOTS * acc_rs + 1) + lane;     // + 1 (the leading column) - 1 (Q = c0 - 1 + lane)
    };
    // one pixel row of an image as a buffer: quads left of the image (offset < 0) and right of it (>= 4 w) fall out by themselves
    auto row_rsrc = [&](const void* base, const Gather& gp, const int u, const int bytes_per_px, const bool valid) {
        const size_t first = ((size_t)gp.img * pix + (size_t)(4 * gp.p + u - 2) * (4 * w)) * bytes_per_px;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base)) + first, 0,
                                                 (base && valid) ? 4 * w * bytes_per_px : 0, 0x00020000);
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
    // ---- col2im, the vector part, kept in one stretch at the head of an iteration: the additions in site order (the quad's running
    //      sum, then v = 6, 7, 8 of site Q - 1, v = 2 .. 5 of site Q, v = 0, 1 of site Q + 1), and for a complete quad tls.cast_bt601
    //      (clip to [16, 235], round half to even, uint8) and the squared error: sum (r - q)^2 over four bytes = r.r + q.q - 2 r.q ------
    struct Px { float p0, p1, p2, p3; unsigned int qv; };
    auto sums = [&](const Item& it, Px& o) {
        o.p0 = (it.a0 + it.a6) + it.b2;
        o.p1 = (it.a1 + it.a7) + it.b3;
        o.p2 = ((it.a2 + it.a8) + it.b4) + it.d0;
        o.p3 = (it.a3 + it.b5) + it.d1;
    };
    auto cast = [&](const Item& it, Px& o, const bool valid) {
        unsigned int qv = 0;
        qv = __builtin_amdgcn_cvt_pk_u8_f32(round_half_even(__builtin_amdgcn_fmed3f(o.p0, 16.f, 235.f)), 0, qv);
        qv = __builtin_amdgcn_cvt_pk_u8_f32(round_half_even(__builtin_amdgcn_fmed3f(o.p1, 16.f, 235.f)), 1, qv);
        qv = __builtin_amdgcn_cvt_pk_u8_f32(round_half_even(__builtin_amdgcn_fmed3f(o.p2, 16.f, 235.f)), 2, qv);
        qv = __builtin_amdgcn_cvt_pk_u8_f32(round_half_even(__builtin_amdgcn_fmed3f(o.p3, 16.f, 235.f)), 3, qv);
        o.qv = qv;
        const unsigned int both = __builtin_amdgcn_udot4(qv, qv, __builtin_amdgcn_udot4(it.rv, it.rv, 0u, false), false);
        const unsigned int e = both - 2u * __builtin_amdgcn_udot4(it.rv, qv, 0u, false);
        se += valid ? e : 0u;
    };
    // ---- col2im, the memory part, handed out between the MFMAs: the running sum back (zeros behind a complete quad), the pixels out ----
    auto put = [&](const Item& it, const Px& o, const Gather& gp, const int u, const bool fin) {
        if (fin) {
            *it.ap = zero4;
            __builtin_amdgcn_raw_buffer_store_b32(o.qv, row_rsrc(out_u8, gp, u, 1, true), it.o1, 0, 0);
            if (out_f32) {
                u32x4 fv;
                fv.x = __float_as_uint(o.p0); fv.y = __float_as_uint(o.p1); fv.z = __float_as_uint(o.p2); fv.w = __float_as_uint(o.p3);
                __builtin_amdgcn_raw_buffer_store_b128(fv, row_rsrc(out_f32, gp, u, 4, true), it.o1 == OOB ? OOB : 4 * it.o1, 0, 0);
            }
        } else {
            *it.ap = make_float4(o.p0, o.p1, o.p2, o.p3);
        }
    };
    // the nine parts and the running sum of a quad, and its reference pixels when it will be complete
    auto get_parts_a = [&](Item& it, const float* pu) {
        it.a6 = lds_get(pu + 6 * PSITES); it.a7 = lds_get(pu + 7 * PSITES); it.a8 = lds_get(pu + 8 * PSITES);
        it.b2 = lds_get(pu + 2 * PSITES + 1); it.b3 = lds_get(pu + 3 * PSITES + 1);
    };
    auto get_parts_b = [&](Item& it, const float* pu) {
        it.b4 = lds_get(pu + 4 * PSITES + 1); it.b5 = lds_get(pu + 5 * PSITES + 1);
        it.d0 = lds_get(pu + 2); it.d1 = lds_get(pu + PSITES + 2);
        const float* const pa = reinterpret_cast<const float*>(it.ap);
        it.a0 = lds_get(pa); it.a1 = lds_get(pa + 1); it.a2 = lds_get(pa + 2); it.a3 = lds_get(pa + 3);
    };
    auto get_ref = [&](Item& it, const Gather& gp, const int u, const bool fin) {
        it.rv = __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(ref, gp, u, 1, fin), it.o1, 0, 0);       // out of range reads 0
    };

#define EAE_T3_WRITE(Pb_, t_, c_)                                                                                     \
    {                                                                                                                 \
        char* const d_ = reinterpret_cast<char*>(Pb_) + pw + 16 * (t_) * PSITES * 4;                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<float*>(d_ + i * PSITES * 4) = c_[i];         \
    }
#define EAE_T3_WRITE5(Pb_, c_)                                                                                        \
    {                                                                                                                 \
        char* const d_ = reinterpret_cast<char*>(Pb_) + pw5;                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) *reinterpret_cast<float*>(d_ + i * PSITES * 4) = c_[i];         \
    }
    // two tap tiles of a chunk: 64 MFMAs, the accumulators alternating; behind every fourth pair a few memory instructions (hook_(0..7))
#define EAE_T3_X(xq_, s_) ((s_) % 4 == 0 ? xq_[(s_) / 4].x : (s_) % 4 == 1 ? xq_[(s_) / 4].y : (s_) % 4 == 2 ? xq_[(s_) / 4].z : xq_[(s_) / 4].w)
#define EAE_T3_PHASE(t0_, ca_, cb_, xq_, hook_)                                                                       \
    {                                                                                                                 \
        T3_ASM_MFMA0(ca_, wr[t0_][0], xq_[0].x);                                                                      \
        T3_ASM_MFMA0(cb_, wr[t0_ + 1][0], xq_[0].x);                                                                  \
        _Pragma("unroll") for (int s = 1; s < 32; ++s) {                                                              \
            T3_ASM_MFMA(ca_, wr[t0_][s], EAE_T3_X(xq_, s));                                                           \
            T3_ASM_MFMA(cb_, wr[t0_ + 1][s], EAE_T3_X(xq_, s));                                                       \
            if (s % 4 == 1) hook_(s / 4);                                                                             \
        }                                                                                                             \
        T3_ASM_SETTLE(ca_, cb_);                                                                                      \
    }
    // One chunk. KIND says which tap tiles the row needs, B which buffers and site registers are this chunk's.
    auto chunk = [&](auto kind_c, auto parity_c, const int ci) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int B = decltype(parity_c)::value;
        float* const Pcur = P0 + B * P_FLOATS;
        float* const Pprv = P0 + (1 - B) * P_FLOATS;
        constexpr int PRV = (1 - B) * P_FLOATS;
        // the chunk behind this one (its sites go on their way now) and what the col2im of this one will need
        int gn = g, kn = k + 1, pn = p, imgn = img, s0n = s0;
        if (kn == kend) {
            kn = kbeg; ++gn; ++pn; s0n += 4;
            s0n = s0n >= SLOTS ? s0n - SLOTS : s0n;
            if (pn == h) { pn = 0; ++imgn; s0n = 0; }
        }
        Gather cur;
        cur.img = img; cur.p = p; cur.c0 = k * CHUNK;
        {
            const int last = k + 1 == kend, own = k >= kown0 && k <= kown1;
            if (KIND == ROW_TOP) cur.bits = Gather::pack(1, last, own, 6, 8, -1, s0);
            else if (KIND == ROW_BOTTOM) cur.bits = Gather::pack(1, last, own, 0, 1, 1, s0);
            else cur.bits = Gather::pack(1, last, own, (g == g0 || p == 0) ? 2 : 0, (g == g1 - 1 || p == h - 1) ? 5 : 8, p == h - 1 ? 5 : 3, s0);
        }
        const bool first_in_row = k == kbeg;
        fetch_setup(imgn, pn, kn, ci + 1 < n_chunks);
        T3_STAMP(0)
        // ---- the vector part of the col2im of the chunk before the previous one (g2d): everything its quads need arrived a chunk ago ----
        if (g2d.on() && g2d.img != se_img) { flush(); se_img = g2d.img; }
        const int uA = wave, uB = 4 + wave;
        const bool actA2 = acts(g2d, uA), finA2 = fins(g2d, uA), actB2 = acts(g2d, uB), finB2 = fins(g2d, uB);
        const bool actC2 = wave == 0 && acts(g2d, 8), tail2 = wave == 1 && g2d.on() && g2d.last();
        // which quads of that chunk are this block's and inside the image (all of them, but for the first and the last chunk of a row)
        const bool in_img2 = (unsigned int)(g2d.c0 - 1 - qown0 + lane) < (unsigned int)(min(w - qown0, own_quads));
        Px oA, oB, oC, oT;
        if (actA2) { sums(itA, oA); if (finA2) cast(itA, oA, in_img2); }
        if (actB2) { sums(itB, oB); if (finB2) cast(itB, oB, in_img2); }
        if (actC2) sums(itC, oC);
        bool finT2 = false;
        if (tail2) {
            sums(itT, oT);
            finT2 = lane < 9 && lane >= g2d.ulo() && lane <= g2d.uhi() && lane <= g2d.ufin();         // per lane here: kernel row u = lane
            cast(itT, oT, itT.o1 != OOB);
        }
        // ---- where the previous chunk's (g1d) quads live -------------------------------------------------------------------------------
        const bool actA1 = acts(g1d, uA), finA1 = fins(g1d, uA), actB1 = acts(g1d, uB), finB1 = fins(g1d, uB);
        const bool actC1 = wave == 0 && acts(g1d, 8), tail1 = wave == 1 && g1d.on() && g1d.last();
        float4* const apA1 = acc_of(g1d, uA, actA1);
        float4* const apB1 = acc_of(g1d, uB, actB1);
        float4* const apC1 = acc_of(g1d, 8, actC1);
        // byte offset of quad (c0 - 1 + lane) inside its pixel row, where the quad is this block's and inside the image
        const int o1_1 = (unsigned int)(g1d.c0 - 1 - qown0 + lane) < (unsigned int)(min(w - qown0, own_quads)) ? 4 * (g1d.c0 - 1) + 4 * lane : OOB;
        f32x4 c0, c1, c2, c3;
        T3_STAMP(1)
        // ---- first third: tiles 0-1. Between the MFMAs: the col2im of g2d goes out, the next chunk's sites are asked for, the previous
        //      chunk's last two tiles go to LDS. Then the barrier: the previous chunk's parts are complete -------------------------------------
        auto hook0 = [&](const int i) {
            switch (i) {
            case 0: if (actA2) put(itA, oA, g2d, uA, finA2); break;
            case 1: if (actB2) put(itB, oB, g2d, uB, finB2); break;
            case 2: if (actC2) put(itC, oC, g2d, 8, false); break;
            case 3:
                if (tail2) {
                    *itT.ap = finT2 ? zero4 : make_float4(oT.p0, oT.p1, oT.p2, oT.p3);
                    // one kernel row per lane: the pixel rows differ from lane to lane, the offset is inside the whole image here
                    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(out_u8 + (size_t)g2d.img * pix, 0, out_u8 ? pix : 0, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b32(oT.qv, ru, itT.o1, 0, 0);
                    if (out_f32) {
                        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(out_f32 + (size_t)g2d.img * pix, 0, pix * 4, 0x00020000);
                        u32x4 fv;
                        fv.x = __float_as_uint(oT.p0); fv.y = __float_as_uint(oT.p1); fv.z = __float_as_uint(oT.p2); fv.w = __float_as_uint(oT.p3);
                        __builtin_amdgcn_raw_buffer_store_b128(fv, rf, itT.o1 == OOB ? OOB : 4 * itT.o1, 0, 0);
                    }
                }
                break;
            case 4: fetch_one(xa[1 - B][0], 0); fetch_one(xa[1 - B][1], 1); fetch_one(xa[1 - B][2], 2); fetch_one(xa[1 - B][3], 3); break;
            case 5: fetch_one(xa[1 - B][4], 4); fetch_one(xa[1 - B][5], 5); fetch_one(xa[1 - B][6], 6); fetch_one(xa[1 - B][7], 7); break;
            case 6: EAE_T3_WRITE(Pprv, 4, c4) break;
            case 7: EAE_T3_WRITE5(Pprv, c5) break;
            }
        };
        if (KIND != ROW_BOTTOM) {
            EAE_T3_PHASE(0, c0, c1, xa[B], hook0)
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hook0(i);
        }
        T3_STAMP(2)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        T3_STAMP(3)
        // ---- second third: tiles 2-3. Between the MFMAs: tiles 0-1 to LDS; everything the previous chunk's (g1d) quads need is asked for;
        //      its last two sites are read for this chunk's buffer ---------------------------------------------------------------------------
        itA.ap = apA1; itB.ap = apB1; itC.ap = apC1;
        itA.o1 = itB.o1 = o1_1;
        auto hook1 = [&](const int i) {
            switch (i) {
            case 0: if (KIND != ROW_BOTTOM) EAE_T3_WRITE(Pcur, 0, c0) break;
            case 1: if (KIND != ROW_BOTTOM) EAE_T3_WRITE(Pcur, 1, c1) break;
            case 2: get_parts_a(itA, puA + PRV); break;
            case 3: get_parts_b(itA, puA + PRV); get_ref(itA, g1d, uA, finA1); break;
            case 4: get_parts_a(itB, puB + PRV); break;
            case 5: get_parts_b(itB, puB + PRV); get_ref(itB, g1d, uB, finB1); break;
            case 6: if (actC1) { get_parts_a(itC, puC + PRV); get_parts_b(itC, puC + PRV); } break;
            case 7:
                if (tail1) {
                    // the last quad of a row (Q = c1 - 1: its third site would be the next chunk's first, and there is none): all nine
                    // kernel rows at once, one per lane -- rare enough for masks
                    const int Q = g1d.c0 - 1 + CHUNK, aq = Q - qown0;
                    const bool ok = lane < 9 && lane >= g1d.ulo() && lane <= g1d.uhi() && g1d.own() && (unsigned int)aq < (unsigned int)own_quads;
                    int slot = g1d.s0() + uT;
                    slot = slot >= SLOTS ? slot - SLOTS : slot;
                    itT.ap = ACC + (ok ? slot * acc_rs + 1 + aq : SLOTS * acc_rs + 1 + lane);
                    itT.o1 = (ok && lane <= g1d.ufin() && Q < w) ? (4 * g1d.p + uT - 2) * (4 * w) + 4 * Q : OOB;
                    get_parts_a(itT, puT + PRV);
                    get_parts_b(itT, puT + PRV);
                    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<uint8_t*>(ref) + (size_t)g1d.img * pix, 0, ref ? pix : 0, 0x00020000);
                    itT.rv = __builtin_amdgcn_raw_buffer_load_b32(rr, itT.o1, 0, 0);
                }
                if (wave == 2 && !first_in_row) {
                    ctx0 = lds_get(Pprv + ctx_src0); ctx1 = lds_get(Pprv + ctx_src1); ctx2 = lds_get(Pprv + ctx_src2);
                }
                break;
            }
        };
        if (KIND == ROW_BODY) {
            EAE_T3_PHASE(2, c2, c3, xa[B], hook1)
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hook1(i);
        }
        T3_STAMP(4)
        // ---- last third: tiles 4-5 (to LDS a chunk later). Between the MFMAs: tiles 2-3 to LDS, the first two columns of this chunk's buffer --
        auto hook2 = [&](const int i) {
            switch (i) {
            case 0: if (KIND == ROW_BODY) EAE_T3_WRITE(Pcur, 2, c2) break;
            case 1: if (KIND == ROW_BODY) EAE_T3_WRITE(Pcur, 3, c3) break;
            case 2:
                if (wave == 2) {
                    Pcur[ctx_dst0] = first_in_row ? 0.f : ctx0; Pcur[ctx_dst1] = first_in_row ? 0.f : ctx1; Pcur[ctx_dst2] = first_in_row ? 0.f : ctx2;
                }
                break;
            default: break;
            }
        };
        if (KIND != ROW_TOP) {
            EAE_T3_PHASE(4, c4, c5, xa[B], hook2)
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hook2(i);
        }
        T3_STAMP(5)
        T3_ADD(1, 0, 1) T3_ADD(2, 1, 2) T3_ADD(3, 2, 3) T3_ADD(4, 3, 4) T3_ADD(5, 4, 5)
        g2d = g1d;
        g1d = cur;
        g = gn; k = kn; p = pn; img = imgn; s0 = s0n;
    };
    auto dispatch = [&](auto parity_c, const int ci) {
        const int kind = g < g0 ? ROW_TOP : (g >= g1 ? ROW_BOTTOM : ROW_BODY);
        if (kind == ROW_BODY) chunk(std::integral_constant<int, ROW_BODY>{}, parity_c, ci);
        else if (kind == ROW_TOP) chunk(std::integral_constant<int, ROW_TOP>{}, parity_c, ci);
        else chunk(std::integral_constant<int, ROW_BOTTOM>{}, parity_c, ci);
    };
    for (int ci = 0; ci < n_chunks; ci += 2) {
        dispatch(std::integral_constant<int, 0>{}, ci);
        if (ci + 1 < n_chunks) dispatch(std::integral_constant<int, 1>{}, ci + 1);
    }
    // ---- the col2im of the last two chunks: the same steps, one after the other -------------------------------------------------------------
    {
        float* const Plast = P0 + ((n_chunks - 1) & 1) * P_FLOATS;
        const int PRV = ((n_chunks - 1) & 1) * P_FLOATS;
        const int uA = wave, uB = 4 + wave;
        auto drain = [&](const Gather& gp) {               // the vector part and the memory part for the quads held in itA .. itT
            if (gp.on() && gp.img != se_img) { flush(); se_img = gp.img; }
            const bool in_img = (unsigned int)(gp.c0 - 1 - qown0 + lane) < (unsigned int)(min(w - qown0, own_quads));
            Px o;
            if (acts(gp, uA)) { sums(itA, o); if (fins(gp, uA)) cast(itA, o, in_img); put(itA, o, gp, uA, fins(gp, uA)); }
            if (acts(gp, uB)) { sums(itB, o); if (fins(gp, uB)) cast(itB, o, in_img); put(itB, o, gp, uB, fins(gp, uB)); }
            if (wave == 0 && acts(gp, 8)) { sums(itC, o); put(itC, o, gp, 8, false); }
            if (wave == 1 && gp.on() && gp.last()) {
                sums(itT, o);
                const bool fin = lane < 9 && lane >= gp.ulo() && lane <= gp.uhi() && lane <= gp.ufin();
                cast(itT, o, itT.o1 != OOB);
                *itT.ap = fin ? zero4 : make_float4(o.p0, o.p1, o.p2, o.p3);
                const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(out_u8 + (size_t)gp.img * pix, 0, out_u8 ? pix : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(o.qv, ru, itT.o1, 0, 0);
                if (out_f32) {
                    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(out_f32 + (size_t)gp.img * pix, 0, pix * 4, 0x00020000);
                    u32x4 fv;
                    fv.x = __float_as_uint(o.p0); fv.y = __float_as_uint(o.p1); fv.z = __float_as_uint(o.p2); fv.w = __float_as_uint(o.p3);
                    __builtin_amdgcn_raw_buffer_store_b128(fv, rf, itT.o1 == OOB ? OOB : 4 * itT.o1, 0, 0);
                }
            }
        };
        drain(g2d);
        EAE_T3_WRITE(Plast, 4, c4)
        EAE_T3_WRITE5(Plast, c5)
        __syncthreads();
        const bool actA1 = acts(g1d, uA), actB1 = acts(g1d, uB), actC1 = wave == 0 && acts(g1d, 8);
        itA.ap = acc_of(g1d, uA, actA1); itB.ap = acc_of(g1d, uB, actB1); itC.ap = acc_of(g1d, 8, actC1);
        itA.o1 = itB.o1 = (unsigned int)(g1d.c0 - 1 - qown0 + lane) < (unsigned int)(min(w - qown0, own_quads)) ? 4 * (g1d.c0 - 1) + 4 * lane : OOB;
        get_parts_a(itA, puA + PRV); get_parts_b(itA, puA + PRV); get_ref(itA, g1d, uA, fins(g1d, uA));
        get_parts_a(itB, puB + PRV); get_parts_b(itB, puB + PRV); get_ref(itB, g1d, uB, fins(g1d, uB));
        if (actC1) { get_parts_a(itC, puC + PRV); get_parts_b(itC, puC + PRV); }
        if (wave == 1 && g1d.on() && g1d.last()) {
            const int Q = g1d.c0 - 1 + CHUNK, aq = Q - qown0;
            const bool ok = lane < 9 && lane >= g1d.ulo() && lane <= g1d.uhi() && g1d.own() && (unsigned int)aq < (unsigned int)own_quads;
            int slot = g1d.s0() + uT;
            slot = slot >= SLOTS ? slot - SLOTS : slot;
            itT.ap = ACC + (ok ? slot * acc_rs + 1 + aq : SLOTS * acc_rs + 1 + lane);
            itT.o1 = (ok && lane <= g1d.ufin() && Q < w) ? (4 * g1d.p + uT - 2) * (4 * w) + 4 * Q : OOB;
            get_parts_a(itT, puT + PRV);
            get_parts_b(itT, puT + PRV);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(ref) + (size_t)g1d.img * pix, 0, ref ? pix : 0, 0x00020000);
            itT.rv = __builtin_amdgcn_raw_buffer_load_b32(rr, itT.o1, 0, 0);
        }
        drain(g1d);
        flush();
        (void)Plast;
    }
#ifdef EAE_T3_TRACE
    tr_acc[7] = __builtin_amdgcn_s_memtime() - ts_begin;
    if (lane == 0 && wave == 0 && sse) {
        for (int i = 0; i < 8; ++i) atomicAdd(&sse[64 + i], (unsigned long long)tr_acc[i]);
        atomicMax(&sse[72], (unsigned long long)tr_acc[7]);
        atomicAdd(&sse[73], (unsigned long long)n_chunks);
    }
#endif
#undef EAE_T3_PHASE
#undef EAE_T3_X
#undef EAE_T3_WRITE
#undef EAE_T3_WRITE5
}

// TF filter [9][9][1][128] -> the A fragments of the 6 tap tiles: [tile][k-step][lane]; lane = kq * 16 + m holds
// W[position 16 tile + m][channel of (k-step, kq)]; zero at the positions no tap has.
__global__ void pack_tconv3_kernel(const float* __restrict__ w_tf, float* __restrict__ wq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_FLOATS) return;
    const int lane = i & 63, s = (i >> 6) & 31, t = i >> 11;
    const int pos = 16 * t + (lane & 15), ci = step_channel(s, lane >> 4);
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
    const int acc_rs = seg_chunks * CHUNK + 1 + (n_seg > 1 ? CHUNK : 0);
    const size_t lds_bytes = ((size_t)(SLOTS + 1) * acc_rs * 4 + 2 * P_FLOATS) * sizeof(float);
    static bool attr_set[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(tconv3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)(((size_t)(SLOTS + 1) * ((MAX_SEG_CHUNKS + 1) * CHUNK + 1) * 4 + 2 * P_FLOATS) * sizeof(float)));
        if (err != hipSuccess) return (int)err;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(tconv3_kernel, dim3((unsigned)(row_strips * n_seg)), dim3(NT), lds_bytes, (hipStream_t)stream, x, w_phase, out_f32,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), n, h, w_in, rows_per_strip, (int)row_strips, seg_chunks, acc_rs);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
