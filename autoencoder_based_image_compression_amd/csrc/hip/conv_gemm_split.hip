// conv_gemm_split.hip -- the four dense 128->128 convolutions (conv_2, conv_3, transpose_conv_1, _2; components.py:126-136,
// 63-75) with bias and (I)GDN epilogue: one work item per wave, where an item is a whole tile (32 positions x 128 channels x
// all of K) or, for the last tiles of the launch, the head or the tail of a tile cut at a K-step boundary.
//
// Why. A batch of Kodak images is 4.5 whole tiles per SIMD for conv_2 and 1.1 for conv_3: the last round of tiles runs on
// a half-empty machine (9 % of the step at batch 24). A tile cannot be cut across positions or
// channels any finer without paying for it (a wave's 4 accumulators are what keeps the 64-cycle MFMA back to back), and it
// cannot be cut along K and summed afterwards either: every output element must stay ONE f32 FMA chain in the documented
// order. What CAN be done without touching a single bit: interrupt the chain at a K-step boundary,
// write the 64 accumulator registers out, and let another wave reload them and continue. The chain, hence the result, is
// the same.
//
// How. Per XCD (blocks b and b + 8 share one) the tiles are laid out as a sequence of items, one per wave of the grid:
//     [ heads of the D last tiles ]  [ the other tiles, whole ]  [ the tails of the D tiles, longest first ]
// D = min(tiles, waves the XCD holds at once), or 0 when the launch is not cut (then this is simply the one-tile-per-wave
// kernel). Head d covers K-steps [0, h_d) of its tile with h_d spread evenly over the tile's length, parks its accumulators
// in the tile's own output pixels (that region has exactly the size of the accumulator tile) and sets a flag; the matching
// tail reloads them, runs [h_d, T) and finishes with the normal epilogue. Workgroups are dispatched in order, so the heads
// start first and the tails last: a tail only ever waits for a wave that was dispatched before it (no residency assumption,
// no deadlock), and by the time the tails come up the heads have long been published. The launch ends with many short
// items that the dispatcher hands to whichever SIMD frees a slot, so all SIMDs run dry within a few K-steps of each other.
// (A persistent form, waves pulling the same items from an atomic counter, was measured 3-4 % slower: a queue balances over
// workers, the dispatcher over SIMDs, which is what matters once fewer items than workers are left.)
//
// Hand-off: the head stores its accumulators WRITE-THROUGH (sc1), waits for the stores (vmcnt(0)) and sets its flag with an
// sc1 store; the tail polls the flag with sc1 loads and reloads the accumulators with sc1 loads (L1 bypassed). Nothing else
// touches those bytes during the launch (the output is write-only), so no L2 holds a stale copy, and there is no release
// fence: an agent-scope release writes back EVERY dirty line of the XCD's L2, which in these store-heavy kernels cost more
// than the tile quantisation it was meant to remove (transpose_conv_2: 1.105 ms with fences against 1.041 uncut).
//
// Workspace (cut launches only): SPLIT_WORDS uint32, zero on entry, zero again on exit (a flag is reset by its consumer),
// so a caller zeroes it once. Launches that may run concurrently need their own workspace. The one exception is the error
// word: a tail whose head has not published within SPIN_LIMIT polls writes NOTHING, counts itself in word SPLIT_ERROR_WORD
// and leaves its flag alone; eae_hip_conv_workspace_collect (conv_gemm.hip) hands the word to the host in stream order and
// zeroes the workspace again. In-order dispatch makes the wait finite in practice; the timeout makes a violation loud.
#include "conv_gemm.h"

#include <cstdlib>

using namespace eae_conv_gemm;

namespace {

constexpr int RING = 8;
#ifndef EAE_EPI_RING
#define EAE_EPI_RING 4
#endif
constexpr int EPI_RING = EAE_EPI_RING;   // gamma ring of the epilogue (see the register budget at the kernel)
constexpr int ABUF = 32 * AS_STRIDE;                 // one activation buffer of one wave
// LDS of a block: two activation buffers per wave + ONE copy of the per-channel vectors of the epilogue (bias | beta, or, with
// the latent stage behind conv_3, bias | beta_in | beta_out | map_mean | bin_widths). Every wave writes the whole copy itself
// (identical values: a benign race, no barrier), so three blocks take 3 x 37.9 KB and leave room for two of the coder's
// decoder blocks (20 KB each) on the same CU: with a copy per wave (3 x 41 KB) a CU that drew two of them ran one GEMM block
// short for the length of a decode.
constexpr int vec_floats(int norm) { return (norm >= NORM_LATENT ? 5 : 2) * EAE_C; }
constexpr int QT_H = 4, QT_W = 8;                    // a wave's tile: 4 x 8 positions
constexpr int MIN_PIECE = 4;                         // K-steps: no head or tail shorter than this
constexpr int SPIN_LIMIT = 1 << 22;                  // ~1 s of polling: a bug, not a wait; counts in the error word, no result

// K-steps of head d of D on a tile of T steps: [MIN_PIECE, T - MIN_PIECE); T itself (no split) for very short tiles
__device__ __forceinline__ int head_steps(int d, int D, int T) {
    return T < 3 * MIN_PIECE ? T : MIN_PIECE + (d * (T - 2 * MIN_PIECE)) / D;
}

// Three waves per SIMD (<= 168 registers) for the plain epilogues; the latent-stage epilogue (two normalisations and the
// quantiser on the register tile) needs ~190 and gets two: conv_3 has a quarter of conv_2's tiles, at Kodak batch sizes its
// SIMDs hold one or two waves anyway.
// Blocks of four waves, each wave on its own (the block is only the unit of dispatch; one-wave blocks measured 1-10 % slower
// alone and no better next to the coder).
// WPB: waves (= items) per block, the unit of dispatch (4 is what ships; 1 behind EAE_HIP_SPLIT_WPB and in the parity tests: see the launcher)
template <int NORM, int WPB>
__global__ __launch_bounds__(64 * WPB, NORM >= NORM_LATENT ? 2 : 3) void conv_gemm_split_kernel(const ConvGemmParams p) {
    __shared__ __attribute__((aligned(16))) float lds[WPB * 2 * ABUF + vec_floats(NORM)];
#ifdef EAE_GEMM_PRIO       // scratch/r04/prio_waves.sh: issue priority of the GEMM waves against the coder's (3: conv_2 1.03 -> 0.96 ms on one
                           // stream, the coder's chains longer, the product mode -0.4 %: not set in the shipped build)
    __builtin_amdgcn_s_setprio(EAE_GEMM_PRIO);
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* wlds = lds + wave * 2 * ABUF;
    float* vec_lds = lds + WPB * 2 * ABUF;
    {
        const float2 z = make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(vec_lds + 2 * lane) = p.bias ? *reinterpret_cast<const float2*>(p.bias + 2 * lane) : z;
        if (NORM == EAE_NORM_GDN || NORM == EAE_NORM_IGDN || NORM == NORM_LATENT)
            *reinterpret_cast<float2*>(vec_lds + EAE_C + 2 * lane) = *reinterpret_cast<const float2*>(p.beta + 2 * lane);
        if (NORM >= NORM_LATENT) {
            if (NORM == NORM_LATENT_PLAIN) *reinterpret_cast<float2*>(vec_lds + EAE_C + 2 * lane) = z;
            *reinterpret_cast<float2*>(vec_lds + 2 * EAE_C + 2 * lane) = NORM == NORM_LATENT ? *reinterpret_cast<const float2*>(p.beta_out + 2 * lane) : z;
            *reinterpret_cast<float2*>(vec_lds + 3 * EAE_C + 2 * lane) = p.map_mean ? *reinterpret_cast<const float2*>(p.map_mean + 2 * lane) : z;
            *reinterpret_cast<float2*>(vec_lds + 4 * EAE_C + 2 * lane) = *reinterpret_cast<const float2*>(p.bin_widths + 2 * lane);
        }
    }
    const int hi = lane >> 5, lj = lane & 31, a_q = lane & 7;
    const int cbase = 4 * hi;

    // this XCD's share of the spatial tiles, and this wave's item
    const int x = (int)blockIdx.x & 7;
    const int wx = p.split_resident_waves_per_xcd;            // waves the XCD holds at once: how many tiles are cut
    const int nsp = p.n * p.tiles_r * p.tiles_c;
    const int q8 = nsp >> 3, r8 = nsp & 7;
    const int cnt_sp = q8 + (x < r8 ? 1 : 0);
    const int first_sp = x * q8 + (x < r8 ? x : r8);
    const int ntt = cnt_sp * p.n_phases;                      // tiles of this share, phase-major (longest phase first)
    const int D = p.split ? (ntt < wx ? ntt : wx) : 0;         // tiles cut in two (0: whole tiles only)
    const int F = ntt - D;
    const int n_items = 2 * D + F;
    unsigned int* flags = p.split_ws + 256 + x * 1024;       // D > 0 only

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)MAX_TAPS * EAE_C * EAE_C * sizeof(float)), 0x00020000);
    const int w_lane = (hi * EAE_C + lj * 4) * 4;             // byte offset inside a k-pair of weight rows
    const int a_off = lj * AS_STRIDE + hi * 16;

    {
        const int item = ((int)blockIdx.x >> 3) * WPB + wave;
        if (item >= n_items) return;
        int d = -1, tau;
        bool is_tail = false;
        if (item < D) { d = item; tau = ntt - D + d; }
        else if (item < D + F) { tau = item - D; }
        else { d = item - D - F; tau = ntt - D + d; is_tail = true; }
        int ph = 0, sp = tau;
        while (sp >= cnt_sp) { sp -= cnt_sp; ++ph; }
        const PhaseDesc& pd = p.phase[ph];
        const int ntaps = pd.ntaps;
        // the phase's tap table, one tap per lane: a K-step picks its tap with v_readlane instead of a scalar load from the
        // kernel arguments (measured neutral, 0-2 %: profiles/r03_gemm_ab.txt; one memory operation fewer per step)
        const int tap_of_lane = pd.tap[lane < MAX_TAPS ? lane : 0];
#define EAE_Q_TAP(ti_) __builtin_amdgcn_readlane(tap_of_lane, (ti_))
        const int T = (EAE_C / KC) * ntaps;
        int s0 = 0, s1 = T;
        if (d >= 0) {
            const int h = head_steps(d, D, T);
            if (is_tail) { if (h >= T) return; s0 = h; }
            else s1 = h;
        }
        int b = first_sp + sp;
        const int tc = b % p.tiles_c; b /= p.tiles_c;
        const int tr = b % p.tiles_r;
        const int img = b / p.tiles_r;

        // activation loader: lane -> (position m_i of this tile, channel quad lane & 7). The eight 8-lane groups of load / staging
        // store i take the positions 4 (g & 3) + 16 (g >> 2) + i: the rows that one LDS cycle of the 8-byte staging stores touches
        // (two or four groups) then start 4 rows = 144 floats apart, i.e. 16 banks modulo 32 and modulo 64 -- disjoint bank sets.
        // (Rounds 1-3: position g + 8 i, adjacent rows per cycle, 36 floats apart: two-way conflicts on every staging store,
        // SQ_LDS_BANK_CONFLICT 14.05 M per conv_2 launch.)
        int a_pr[4], a_pc[4], a_ok[4];
        const int a_m0 = 4 * ((lane >> 3) & 3) + 16 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = a_m0 + i;
            a_pr[i] = tr * QT_H + m / QT_W;
            a_pc[i] = tc * QT_W + m % QT_W;
            a_ok[i] = (a_pr[i] < p.hp) & (a_pc[i] < p.wp);
        }
        const float* in_img = p.in + (size_t)img * p.hin * p.win * EAE_C;
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(in_img), 0, (int)((size_t)p.hin * p.win * EAE_C * sizeof(float)), 0x00020000);
        // this lane's output pixel (position lj of the tile): also where the accumulators of an interrupted tile wait
        const int pr = tr * QT_H + lj / QT_W, pc = tc * QT_W + lj % QT_W;
        const bool valid = pr < p.hp && pc < p.wp;
        float* out_img = p.out + (size_t)img * p.hout * p.wout * EAE_C;
        const int o_off = ((pr * p.out_stride + pd.out_a) * p.wout + (pc * p.out_stride + pd.out_b)) * EAE_C;   // floats, inside the image
        float* o = out_img + o_off;
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            out_img, 0, (int)((size_t)p.hout * p.wout * EAE_C * sizeof(float)), 0x00020000);
        const int park = valid ? (o_off + cbase) * 4 : -1;      // byte offset of this lane's parked accumulators (-1: beyond the buffer)

        float4 a_reg[4];
#define EAE_Q_PREFETCH_A(packed_, ci0_)                                                                              \
        {                                                                                                            \
            const int dr_ = ((packed_) & 0xFF) - 8, dc_ = (((packed_) >> 8) & 0xFF) - 8;                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
                const int r_ = a_pr[i] * p.in_stride + dr_;                                                          \
                const int c_ = a_pc[i] * p.in_stride + dc_;                                                          \
                const int ok_ = a_ok[i] & ((unsigned)r_ < (unsigned)p.hin) & ((unsigned)c_ < (unsigned)p.win);       \
                const int lin_ = ((r_ * p.win + c_) * EAE_C + (ci0_) + 4 * a_q) * 4;                                 \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok_ ? lin_ : -1, 0, 0);              \
                a_reg[i] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),          \
                                       __uint_as_float(v_.w));                                                       \
            }                                                                                                        \
        }
#define EAE_Q_STAGE_A(buf_)                                                                                          \
        {                                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
                float* dst_ = wlds + (buf_) * ABUF + (a_m0 + i) * AS_STRIDE + 2 * a_q;                              \
                *reinterpret_cast<float2*>(dst_) = make_float2(a_reg[i].x, a_reg[i].z);                              \
                *reinterpret_cast<float2*>(dst_ + 16) = make_float2(a_reg[i].y, a_reg[i].w);                         \
            }                                                                                                        \
        }
        // byte offset of the weight slab of (tap index, 32-channel chunk); K order: chunk (outer), then tap
#define EAE_Q_SLAB(ti_, ch_) ((((EAE_Q_TAP(ti_) >> 16) * EAE_C + (ch_) * KC) * EAE_C) * 4)
#define EAE_Q_W_LOAD(dst_, slab_, kk_)                                                                               \
        {                                                                                                            \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane + (kk_) * 2 * EAE_C * 4, (slab_), 0); \
            dst_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z),                  \
                               __uint_as_float(v_.w));                                                               \
        }

        f32x16 acc[4];
        if (s0 == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        } else {
            // the tail of an interrupted tile: wait for its head (published long ago, normally), then continue the chain
            int gave_up = 0;
            if (lane == 0) {
                int spins = 0;
                while (__hip_atomic_load(flags + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > p.split_spin_limit) { gave_up = 1; break; }
                }
            }
            // No result on a timeout: the tile keeps whatever its head parked, the error word counts the tail, and the flag is
            // left alone (a head that publishes late would otherwise leave a stale 1 behind a reset). The host reads the
            // word in stream order (eae_hip_conv_workspace_collect), raises, and zeroes the workspace again.
            if (__builtin_amdgcn_readfirstlane(gave_up)) {
                if (lane == 0) atomicAdd(p.split_ws + SPLIT_ERROR_WORD, 1u);
                return;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // sc1 (aux 16): served from L2 / memory, never from this CU's L1; lanes outside the image read 0
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(out_rsrc, park, (32 * t + 8 * g) * 4, 16);
                    acc[t][4 * g + 0] = __uint_as_float(v.x); acc[t][4 * g + 1] = __uint_as_float(v.y);
                    acc[t][4 * g + 2] = __uint_as_float(v.z); acc[t][4 * g + 3] = __uint_as_float(v.w);
                }
            if (lane == 0) __hip_atomic_store(flags + d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
        }

        int tap_i = s0 % ntaps, chunk = s0 / ntaps;
        float4 ring[RING];
        {
            const int slab0 = EAE_Q_SLAB(tap_i, chunk);
#pragma unroll
            for (int i = 0; i < RING; ++i) EAE_Q_W_LOAD(ring[i], slab0, i)
        }
        EAE_Q_PREFETCH_A(EAE_Q_TAP(tap_i), chunk * KC)
        EAE_Q_STAGE_A(0)
        for (int step = s0; step < s1; ++step) {
            int nti = tap_i + 1, nch = chunk;
            if (nti == ntaps) { nti = 0; ++nch; }
            if (step + 1 >= s1) { nti = tap_i; nch = chunk; }      // the last step re-loads itself: loads stay unconditional
            const int slab_cur = EAE_Q_SLAB(tap_i, chunk);
            const int slab_nxt = EAE_Q_SLAB(nti, nch);
            EAE_Q_PREFETCH_A(EAE_Q_TAP(nti), nch * KC)
            const float4* a_rd = reinterpret_cast<const float4*>(wlds + ((step - s0) & 1) * ABUF + a_off);
            const float4 a0 = a_rd[0], a1 = a_rd[1], a2 = a_rd[2], a3 = a_rd[3];
            const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float4 wq = ring[kk % RING];
                acc[0] = mfma32(wq.x, av[kk], acc[0]);      // A = W^T[co = 32 t + lj][k], B = X^T[k][pos = lj]
                acc[1] = mfma32(wq.y, av[kk], acc[1]);
                acc[2] = mfma32(wq.z, av[kk], acc[2]);
                acc[3] = mfma32(wq.w, av[kk], acc[3]);
                if (kk + RING < KC / 2) { EAE_Q_W_LOAD(ring[kk % RING], slab_cur, kk + RING) }
                else { EAE_Q_W_LOAD(ring[kk % RING], slab_nxt, kk + RING - KC / 2) }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            EAE_Q_STAGE_A(((step - s0) + 1) & 1)
            tap_i = nti;
            chunk = nch;
        }

        if (s1 == T) {
            if constexpr (NORM >= NORM_LATENT) {
                // conv_3: bias_add, then the whole latent stage on the register tile (latent_body.h); the tile's pixels in
                // p.out (where a cut tile's accumulators waited) receive the decoder's input
                if (p.bias) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 bv = *reinterpret_cast<const float4*>(vec_lds + 32 * t + 8 * g + cbase);
                            acc[t][4 * g + 0] = acc[t][4 * g + 0] + bv.x;
                            acc[t][4 * g + 1] = acc[t][4 * g + 1] + bv.y;
                            acc[t][4 * g + 2] = acc[t][4 * g + 2] + bv.z;
                            acc[t][4 * g + 3] = acc[t][4 * g + 3] + bv.w;
                        }
                }
                wave_latent_body<NORM == NORM_LATENT, NORM == NORM_LATENT>(acc, vec_lds + EAE_C, p.gamma, p.gamma_out, p.latent, valid,
                                                                           (long)img, pr * p.wp + pc, p.hp * p.wp, lane);
            } else {
                wave_epilogue<NORM, EPI_RING>(acc, vec_lds, p.bias != nullptr, p.gamma, o, valid, lane);
            }
        } else {
            // a head: park the accumulators in the tile's own output pixels, write-through, and publish them. (These 16-byte
            // buffer stores carry MFMA results, which the compiler keeps apart from their readers whatever the store looks like.
            // The same store right behind VECTOR instructions that write its data registers is NOT kept apart when its soffset
            // is a register, and then stores stale lanes: DESIGN.md section 10, "a trap met on the way".)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x4 v = {__float_as_uint(acc[t][4 * g]), __float_as_uint(acc[t][4 * g + 1]),
                                     __float_as_uint(acc[t][4 * g + 2]), __float_as_uint(acc[t][4 * g + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, park, (32 * t + 8 * g) * 4, 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this wave has reached memory
            if (lane == 0 && !p.split_mute_heads) __hip_atomic_store(flags + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#undef EAE_Q_PREFETCH_A
#undef EAE_Q_STAGE_A
#undef EAE_Q_SLAB
#undef EAE_Q_TAP
#undef EAE_Q_W_LOAD
    }
}

}  // namespace

int eae_conv_gemm::launch_split(ConvGemmParams& p, hipStream_t stream, int cut) {
    const int cus = eae_compute_units();
    if (cus < 8 || (cus & 7)) return 1;
    p.tiles_r = (p.hp + QT_H - 1) / QT_H;
    p.tiles_c = (p.wp + QT_W - 1) / QT_W;
    const long nsp = (long)p.n * p.tiles_r * p.tiles_c;
    const long tiles = nsp * p.n_phases;
    const long simds = 4L * cus;
    // cut: -1 = decide here, 0 = whole tiles, 1..3 = cut, sized for that many resident waves per SIMD (tests)
    long k = tiles / simds;                     // whole tiles per SIMD
    if (k > 3) k = 3;
    if (k < 1) k = 1;
    // The hand-off between the two halves of a cut tile rests on what was validated on gfx950 only: workgroups dispatched in
    // grid order (a tail only waits for a wave dispatched before it), sc1 stores / loads meeting in the XCD's L2. On any
    // other device -- and on a partition of an MI355X (DPX / QPX / CPX: the sizing below assumes 8 XCDs per logical device) -- the
    // launch is never cut of its own accord (whole tiles: the same bits).
    const bool validated = eae_is_whole_mi355x();
    bool split = cut > 0;
    if (cut > 0) k = cut > 3 ? 3 : cut;
    if (cut < 0 && validated && p.split_ws && p.n_phases == 1 && tiles >= simds) {
        // Uncut, the launch takes as long as its busiest SIMD: ceil(tiles / SIMDs) tiles; cut, tiles / SIMDs plus ~2 % for
        // the hand-offs. Only the convolutions: the tiles of the transposed ones are short (16-36 K-steps against 100), their
        // last round costs little and cutting them was measured to lose (Kodak batch 24, bursts: transpose_conv_1 0.286 ms
        // uncut, 0.295-0.310 cut; _2 1.009 / 1.014-1.031; conv_2 0.995 / 0.917; conv_3 0.382 / 0.249).
        // Also when the tile count is a whole number of rounds, from three tiles per SIMD on: alone on the GPU a launch of
        // exactly 9 tiles per SIMD runs 0.6 % faster uncut (conv_2 of 48 images: 1.778 against 1.788 ms), but next to the coder's
        // long-lived waves some SIMDs fall behind and an uncut launch then waits for a whole tile on them (0.77 of peak in the
        // pipeline against 0.90 alone; cut: 0.83); the short items at the end of a cut launch absorb that. With exactly two
        // tiles per SIMD (conv_2 of 64 images of 256x256) every tile would be cut and the tails would be dispatched while
        // their heads still run: measured slower there (0.50 against 0.44 ms in the pipeline), so such launches stay whole.
        const long rounds = (tiles + simds - 1) / simds;
        split = (double)rounds * simds > 1.03 * (double)tiles || tiles >= 3 * simds;
    }
    if (split && !p.split_ws) return EAE_HIP_BAD_ARGUMENT;
    if (p.norm >= NORM_LATENT && k > 2) k = 2;       // that instance holds two waves per SIMD
    if ((cus / 8) * 4 * k > 1024) return 1;          // flag table: 1024 cut tiles per XCD
    // longest phase first (insertion sort of <= 4 descriptors): every CU works through the same mix of phases, and the last
    // tiles of the launch -- the ones that get cut -- are the short ones
    for (int i = 1; i < p.n_phases; ++i)
        for (int j = i; j > 0 && p.phase[j].ntaps > p.phase[j - 1].ntaps; --j) {
            const PhaseDesc tmp = p.phase[j]; p.phase[j] = p.phase[j - 1]; p.phase[j - 1] = tmp;
        }
    p.split = split ? 1 : 0;
    p.split_spin_limit = SPIN_LIMIT;
    p.split_mute_heads = 0;
    if (split) {
        // test hook (eae_hip_debug_set_split_mute, tests/test_gpu_conv_split.py): heads that never publish, tails that give up after ~1 ms
        if (g_eae_launch_options.split_mute) { p.split_mute_heads = 1; p.split_spin_limit = 1 << 12; }
    }
    p.split_resident_waves_per_xcd = (cus / 8) * 4 * (int)k;
    // one wave per item, blocks of 4 items, the 8 shares interleaved: the largest share decides the grid
    const long cnt_max = (nsp + 7) / 8 * p.n_phases;
    const long d_max = split ? (cnt_max < p.split_resident_waves_per_xcd ? cnt_max : p.split_resident_waves_per_xcd) : 0;
    // Blocks of four waves (the unit of dispatch). Next to the coder's long-lived waves a SIMD that hosts one runs its GEMM wave a
    // third slower, and a four-wave block then holds the slots of the CU's other three SIMDs until its slowest wave is through; with
    // one-wave blocks (EAE_HIP_SPLIT_WPB=1) only the hosting SIMD falls behind: conv_2 + GDN_2, the layer with the longest tiles, runs
    // 1.035 -> 0.997 ms in the kernel-by-kernel schedule (conv GEMM 0.778 -> 0.789 of peak), but the product mode -- three transform
    // streams already fill a launch's tails with the next batches' kernels, and four times as many blocks cost dispatch -- loses 0.6 %
    // (3,103 -> 3,085 Mpx/s), and conv_3 / the transposed convolutions lose 2-5 % by themselves (profiles/r04_waves_per_block.log).
    // So four it stays; the one-wave instance is kept behind the override and in the parity tests.
    const int wpb = g_eae_launch_options.split_wpb == 1 ? 1 : 4;
    const int grid = (int)((cnt_max + d_max + wpb - 1) / wpb) * 8;
#define EAE_LAUNCH_S(N_)                                                                                                  \
    {                                                                                                                     \
        if (wpb == 1) hipLaunchKernelGGL((conv_gemm_split_kernel<N_, 1>), dim3(grid), dim3(64), 0, stream, p);            \
        else hipLaunchKernelGGL((conv_gemm_split_kernel<N_, 4>), dim3(grid), dim3(256), 0, stream, p);                    \
    }
    if (p.norm == NORM_LATENT) EAE_LAUNCH_S(NORM_LATENT)
    else if (p.norm == NORM_LATENT_PLAIN) EAE_LAUNCH_S(NORM_LATENT_PLAIN)
    else if (p.norm == EAE_NORM_GDN) EAE_LAUNCH_S(EAE_NORM_GDN)
    else if (p.norm == EAE_NORM_IGDN) EAE_LAUNCH_S(EAE_NORM_IGDN)
    else EAE_LAUNCH_S(EAE_NORM_NONE)
#undef EAE_LAUNCH_S
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
