// coder_simd.hip -- the lossless coder reorganised for a 64-wide machine: the SERIAL part of a map's arithmetic coder cut down
// to the interval arithmetic itself (64 maps per wavefront, every lane in step), everything else data-parallel around it.
//
// A binary arithmetic coder is a serial chain per stream (lossless/c++/source/BinaryArithmeticCoder.cpp:144-252), so the only
// parallelism is across the feature maps of a batch, and what a batch costs is the LENGTH of the longest map's chain (the
// transforms of the batches in flight have to cover it) times the instructions per link (every one of them delays the MFMA waves
// the coder's waves share their SIMDs with). Round 2 ran ~75 (encoder) and ~105 (decoder) vector instructions per binary
// decision, with bit I/O, Exp-Golomb escapes and sign bits inside the chain, and a second pass for streams longer than an LDS
// window. Here the chain only does what is inherently serial (coder/lean_step.h, pinned to coder_core.h on the CPU by
// tests/test_lean_coder.py):
//
//   encode   (1) binarise_kernel, one wavefront per map, parallel over the symbols: UEG0 binarisation (LosslessCoder.cpp:232-252)
//                -> a list of (bit, context) decisions, one byte each, and the complete bypass stream (signs, Exp-Golomb
//                suffixes: bit offsets by wave prefix sums).
//            (2) bac_encode_core_kernel, 64 maps per wavefront: decision j of every lane's map through the interval update; the
//                renormalisation is closed-form and NOTHING is written to the stream: each decision leaves a 32-bit record (the 16
//                bits that may leave, the number of E1/E2 shifts, the number of E3 scalings), four records per 16-byte store.
//            (3) emit_kernel, one wavefront per map, parallel over the records: the position of every leaving bit is a prefix sum
//                of the emitted lengths, the pending-E3 queue a segmented sum of the scalings; tiles of 64 records are assembled
//                in LDS and stored as whole words. Also the flush (stop_encoding) and the capacity check.
//   decode   (4) bac_decode_core_kernel, 64 maps per wavefront: one decision per lane per step; the stream reaches the lanes through
//                a 32-word LDS ring per lane that is topped up from memory every 8 steps (16-byte loads issued a checkpoint ahead
//                of their use: no stream is "too long", nothing waits for memory), the code register and the interval are
//                top-aligned; the lane tracks the truncated-unary context (it selects the next probability) and stores ONE BYTE per
//                symbol, the unary prefix 0..L.
//            (5) debinarise_kernel, one wavefront per map, parallel over the symbols: signs and Exp-Golomb suffixes out of the
//                bypass stream (prefix sums; the escapes of a tile, whose lengths depend on the stream, in a short serial loop
//                over LDS), the symbols, and the comparison with the encoder's input (the assert of lossless/compression.py:146-153).
//
// Anything these kernels do not handle -- an error of any kind (their exact code and stage matter), L == 0 or L > 32, a pending
// E3 queue of thousands of bits -- marks the map RETRY, and the general per-lane kernel (the shared core of coder_core.h,
// statement for statement the reference) recodes that map from scratch. Results are therefore identical to the host library's in
// every case; tests/test_coder_device.py compares bytes, bit counts, symbols, statuses and stages.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../coder/lean_step.h"
#include "eae_hip.h"

// coder_device.hip
int eae_coder_generic_encode(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                             const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                             uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream);
int eae_coder_generic_decode(uint32_t n_maps, uint32_t map_size, int16_t* out, uint8_t L, const double* probs,
                             const int32_t* prob_row, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                             const uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream);

// gfx950: a 64-bit shift (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose shift amount sits in the LAST register of the wave's VGPR
// allocation gives wrong results whenever other waves share the SIMD (csrc/isa_guard.py rule 2, DESIGN.md section 5: the fault of
// round 3's first decoder core). These kernels shift 64-bit windows by computed amounts, so each of them names the last register of
// ITS OWN allocation in an empty asm: the allocator cannot give it to anybody, and the allocation stays what the kernel needs
// (24 / 40 / 48 registers). Reserving v63 in all of them was the first form of this: the wide passes (one wavefront per map) at 64
// registers instead of 24-48 took the room of the transforms' waves -- conv_2 +5 %, tconv3 +19 % in the one-stream leg
// (profiles/r04_tight_alloc.log). A kernel that outgrows its number moves to the next granule with its real last register
// unreserved: the guard checks the shipped ISA whatever this does (and tests/test_isa_guard.py pins the five allocations).
#ifndef EAE_RES_BINARISE
#define EAE_RES_BINARISE 47
#endif
#ifndef EAE_RES_ENCODE_CORE
#define EAE_RES_ENCODE_CORE 55
#endif
#ifndef EAE_RES_EMIT
#define EAE_RES_EMIT 39
#endif
#ifndef EAE_RES_DECODE_CORE
#define EAE_RES_DECODE_CORE 55
#endif
#ifndef EAE_RES_DEBINARISE
#define EAE_RES_DEBINARISE 23
#endif
#ifndef EAE_RES_DEBINARISE_STAGED
#define EAE_RES_DEBINARISE_STAGED 39
#endif
#ifndef EAE_RES_DECODE_CORE_CHUNKED
#define EAE_RES_DECODE_CORE_CHUNKED 63      // the resumable form of the decoder core (a parked state to load and store)
#endif
#ifndef EAE_DECODE_TOPUP_ZEROS
#define EAE_KEEP_VGPR_FREE_(n) asm volatile("; v" #n " reserved: the last register of the allocation holds no operand" ::: "v" #n)
#define EAE_KEEP_LAST_VGPR_FREE(n) EAE_KEEP_VGPR_FREE_(n)
#else      // the first decoder core is kept as it was built (40 of 40 registers): scratch/r04, tests/test_isa_guard.py
#define EAE_KEEP_LAST_VGPR_FREE(n)
#endif

namespace {

using namespace eae_core;
using namespace eae_lean;


#ifndef EAE_SIMD_PRIO
#define EAE_SIMD_PRIO 3
#endif

constexpr int32_t RETRY = -100;           // internal: recode this map with the general kernel
constexpr uint32_t kMaxFastL = 32;        // contexts staged in LDS: 33 x 64 lanes x 8 B = 17 KB
constexpr uint32_t kRecordPad = 12;       // records beyond a map's decisions: the stop record + the tail of the last 16-byte store
constexpr uint32_t kRing = 32;            // decoder: stream words per lane in LDS (a step takes <= 30 bits: 8 steps <= 8 words)
constexpr uint32_t kBypassStage = 512;    // debinarise: words of a map's bypass stream staged in LDS (a multiple of 512: 16,384 bits, several times a map's at 3 bpp)
constexpr uint32_t kDebinariseStage = 1024;      // debinarise: prefix bytes / expected symbols staged in LDS at a time (5.6 KB of LDS per wavefront in all)
constexpr uint32_t kEmitWords = 224;      // emit: 64-bit words of one tile of 256 records in LDS (56 bits per record on average: beyond that the general kernel)

struct SimdParams {
    uint32_t n_maps, map_size, L, dcap;   // dcap: bytes of decision storage per map (multiple of 8)
    uint32_t rcap;                        // records per map (dcap + kRecordPad, a multiple of 4)
    const int16_t* symbols;
    int16_t* decoded;
    const double* probs;
    const int32_t* prob_row;
    uint8_t* streams;
    uint64_t stride;
    uint32_t* bac_bits;
    uint32_t* bypass_bits;
    int32_t* status;
    int32_t* stage;
    uint8_t* decisions;                   // [group][j / 8][lane][8]
    uint32_t* ndec;                       // [n_maps]: decisions of a map (encode); 1 = handed to the general kernel (decode)
    uint32_t* records;                    // [n_maps][rcap]
    uint8_t* prefixes;                    // [n_maps][map_size] (decode; shares the records' memory -- its own in the chunked round trip)
    // the chunked round trip (eae_hip_coder_roundtrip_trailing): the serial chains cut into `nchunks` launches each, so that the
    // emit pass and the decoder of chunk c run while the encoder core is on chunk c + 1 (what crosses a launch is below)
    uint32_t chunk, nchunks;
    uint32_t* group_steps;                // [groups]: steps of the encoder core of a group of 64 maps, fixed by its first chunk
    uint2* enc_state;                     // [n_maps]: the interval between two chunks of the encoder core
    uint4* emit_state;                    // [n_maps]: bits emitted, pending E3 scalings, the partial 64-bit word
    uint32_t* avail_bits;                 // [n_maps]: stream bits that are in memory (whole 64-bit words) after the last emit chunk
    uint4* dec_state;                     // [n_maps]: interval, code register, stream bits taken
    uint2* dec_state2;                    // [n_maps]: unary count, symbol index
};

// Steps of one chunk of a chain of `steps` steps cut into `nchunks`: a multiple of 64 (whole record tiles, whole 8-step rounds).
__host__ __device__ inline uint32_t chunk_span(uint32_t steps, uint32_t nchunks) {
    const uint32_t span = ((steps + nchunks - 1u) / nchunks + 63u) & ~63u;
    return span ? span : 64u;
}

// Exclusive prefix sum over the 64 lanes, and the total. Data-parallel primitives, not cross-lane loads: four row_shr steps scan
// the rows of 16 lanes, row_bcast:15 / :31 carry the row totals on (the sequence LLVM's atomic optimiser emits for gfx9): 6 DPP
// additions where six __shfl_up rounds were 6 x (ds_bpermute + compare + add).
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t& total) {
    int inc = (int)v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);     // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);     // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);     // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);     // row_shr:8
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    total = (uint32_t)__builtin_amdgcn_readlane(inc, 63);
    return (uint32_t)inc - v;
}
// Exclusive running maximum over the lanes below (values >= -1; -1 where there is none): the same six DPP steps on max, applied
// to the input moved up by one lane (wave_shr:1).
__device__ __forceinline__ int wave_exclusive_max(int v) {
    int m = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);     // wave_shr:1
#define EAE_MAX_STEP(ctrl_, rows_)                                                        \
    {                                                                                     \
        const int o_ = __builtin_amdgcn_update_dpp(-1, m, ctrl_, rows_, 0xF, false);      \
        m = o_ > m ? o_ : m;                                                              \
    }
    EAE_MAX_STEP(0x111, 0xF)
    EAE_MAX_STEP(0x112, 0xF)
    EAE_MAX_STEP(0x114, 0xF)
    EAE_MAX_STEP(0x118, 0xF)
    EAE_MAX_STEP(0x142, 0xA)
    EAE_MAX_STEP(0x143, 0xC)
#undef EAE_MAX_STEP
    return m;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// (1) one wavefront per map: symbols -> decisions + bypass stream
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void binarise_kernel(const SimdParams p) {
    EAE_KEEP_LAST_VGPR_FREE(EAE_RES_BINARISE);
    __shared__ unsigned long long ybuf[40];            // one tile of bypass bits: carry word + 64 x (33 + 1) bits
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row < 0) {                                     // exception map: no streams (compression.py:68-75)
        if (lane == 0) { p.ndec[m] = 0; p.bac_bits[m] = 0; p.bypass_bits[m] = 0; p.status[m] = 0; if (p.stage) p.stage[m] = 0; }
        return;
    }
    const uint32_t L = p.L;
    const int16_t* in = p.symbols + (size_t)m * p.map_size;
    uint8_t* dec = p.decisions + (size_t)(m >> 6) * 64u * p.dcap + (size_t)(m & 63u) * 8u;
    unsigned long long* ystream = reinterpret_cast<unsigned long long*>(p.streams + (uint64_t)m * p.stride + p.stride / 2);
    const uint32_t size_bits = round_up_to_byte(required_bits(p.map_size, L));
    uint32_t jbase = 0, bbase = 0;
    bool overflow = false;
    if (lane < 40) ybuf[lane] = 0;
    for (uint32_t t = 0; t < p.map_size; t += 64) {
        const uint32_t i = t + lane;
        const bool valid = i < p.map_size;
        const int s = valid ? (int)in[i] : 0;
        const uint32_t a = (uint32_t)(s < 0 ? -s : s);
        const uint32_t ones = a < L ? a : L;
        const uint32_t nd = valid ? ones + (a < L ? 1u : 0u) : 0u;
        // bypass bits of this symbol, first bit in time at bit 0: Exp-Golomb of a - L (LosslessCoder.cpp:58-111), then the sign
        unsigned long long bb = 0;
        uint32_t nb = 0;
        if (valid && a >= L) {
            const uint32_t v = a - L + 1u;
            const uint32_t n = count_nb_bits(v) - 1u;
            bb = ((1ull << n) - 1ull) | ((unsigned long long)(n ? rev16(v - (1u << n)) >> (16u - n) : 0u) << (n + 1u));
            nb = 2u * n + 1u;
        }
        if (valid && s != 0) {
            bb |= (unsigned long long)(s > 0 ? 1u : 0u) << nb;   // LosslessCoder.cpp:22-37: 0 = negative
            nb++;
        }
        // both prefix sums in one scan: decisions (<= 33 per symbol) in the low half, bypass bits (<= 34) in the high half
        uint32_t both_tot;
        const uint32_t both = wave_exclusive_scan(nd | (nb << 16), both_tot);
        const uint32_t jo = both & 0xFFFFu, bo = both >> 16, jtot = both_tot & 0xFFFFu, btot = both_tot >> 16;
        // decisions: `ones` ones in contexts 0..ones-1, then a zero in context a when a < L (LosslessCoder.cpp:167-191)
        for (uint32_t q = 0; q < nd; q++) {
            const uint32_t j = jbase + jo + q;
            dec[(size_t)(j >> 3) * 512u + (j & 7u)] = (uint8_t)((q << 1) | (q < ones ? 1u : 0u));
        }
        // bypass: assemble the tile in LDS behind the carried partial word, flush the complete words
        if (nb) {
            const uint32_t pos = (bbase & 63u) + bo;
            atomicOr(&ybuf[pos >> 6], bb << (pos & 63u));
            if ((pos & 63u) + nb > 64u) atomicOr(&ybuf[(pos >> 6) + 1u], bb >> (64u - (pos & 63u)));
        }
        if (bbase + btot > size_bits) overflow = true;          // Bitstream.cpp:32-35 -> general kernel for the exact error
        if (overflow) break;
        const uint32_t nfull = ((bbase & 63u) + btot) >> 6;
        const unsigned long long mine = lane < 40 ? ybuf[lane] : 0ull;
        const unsigned long long carry = ybuf[nfull];
        if (lane < nfull) ystream[(bbase >> 6) + lane] = mine;
        if (lane < 40) ybuf[lane] = lane == 0 ? carry : 0ull;
        jbase += jtot;
        bbase += btot;
    }
    if (lane == 0) {
        if (!overflow && (bbase & 63u)) ystream[bbase >> 6] = ybuf[0];     // Bitstream flush of the partial word
        p.ndec[m] = jbase;
        p.bypass_bits[m] = bbase;
        p.bac_bits[m] = 0;
        p.status[m] = overflow ? RETRY : 0;
        if (p.stage) p.stage[m] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// (2) 64 maps per wavefront: decision j of every map through the interval update, in step; one record per decision
// ---------------------------------------------------------------------------------------------------------------------
extern __shared__ double lds_dyn[];

template <bool CHUNKED>
__global__ __launch_bounds__(64) void bac_encode_core_kernel(const SimdParams p) {
    EAE_KEEP_LAST_VGPR_FREE(EAE_RES_ENCODE_CORE);
    __builtin_amdgcn_s_setprio(EAE_SIMD_PRIO);
    const uint32_t lane = threadIdx.x;
    const uint32_t m = blockIdx.x * 64u + lane;
    const bool in_range = m < p.n_maps;
    const int32_t row = in_range ? (p.prob_row ? p.prob_row[m] : (int32_t)m) : -1;
    const bool live = in_range && row >= 0 && p.status[m] == 0;
    const uint32_t L = p.L;
    double* probs = lds_dyn;                            // [context][lane], scaled by 2^-16 (lean_step.h)
    bool retry = false;
    if (live)
        for (uint32_t k = 0; k < L; k++) {
            const double pk = p.probs[(size_t)row * L + k];
            probs[k * 64u + lane] = scale_probability(pk);
            // An invalid probability only matters if its context is coded (BinaryArithmeticCoder.cpp:146-153): the general
            // kernel sorts that out; here the whole map is handed over so that the steps below need no check.
            if (!(pk > 0. && pk < 1.)) retry = true;
        }
    const uint32_t nd = live && !retry ? p.ndec[m] : 0u;
    uint32_t steps = wave_max(nd);
    const uint32_t nd_all = ~wave_max(~(nd ? nd : 0xFFFFFFFFu));      // the fewest decisions of a lane that has any (all lanes idle: 2^32 - 1)
    const uint8_t* dec = p.decisions + (size_t)blockIdx.x * 64u * p.dcap + (size_t)lane * 8u;
    uint32_t* rec = p.records + (size_t)(in_range ? m : 0u) * p.rcap;
    Interval s = interval_init();
    uint32_t j_first = 0;
    if (CHUNKED) {
        // the group's chain length is fixed by the first chunk (a map that drops out later -- its emit pass gave up -- must not
        // move the chunk boundaries of the others); this chunk's share of it, and the interval where the last chunk left it
        if (p.chunk == 0u) { if (lane == 0u) p.group_steps[blockIdx.x] = steps; }
        else steps = p.group_steps[blockIdx.x];
        const uint32_t span = chunk_span(steps, p.nchunks);
        j_first = p.chunk * span < steps ? p.chunk * span : steps;
        steps = j_first + span < steps ? j_first + span : steps;
        if (p.chunk != 0u && in_range) { const uint2 st = p.enc_state[m]; s.lo = st.x; s.hc = st.y; }
    }
    // Memory in the loop, and what the loop must NOT wait for (round 6). A round reads eight decision bytes per lane and stores two
    // 16-byte groups of records; what the chain needs is the decisions of the round in hand, requested two rounds earlier. Every
    // load and store of the loop is issued unconditionally -- a round beyond the group's last reads the last round again, a lane
    // with nothing to store stores into the unused tail of its own record region (entries dcap + 8 .. dcap + 11: the stop record
    // is at most entry dcap, a group store reaches at most dcap + 3) -- because a memory instruction behind a branch makes the
    // compiler's `s_waitcnt vmcnt` a wait for EVERYTHING in flight: rounds 3-5 had the prefetch and the stores behind `if`s and so
    // waited at every round for the stores issued a moment earlier (the round as long as a store's way to memory, 1.2 us for
    // 0.67 us of arithmetic: one Kodak image's core 277 us where the chain is 150).
    const uint32_t last_round = (steps ? steps - 1u : 0u) >> 3;
    auto decisions_of = [&](uint32_t round) __attribute__((always_inline)) {
        return *reinterpret_cast<const uint2*>(dec + (size_t)(round < last_round ? round : last_round) * 512u);
    };
    uint32_t* const sink = rec + p.dcap + 8u;
    const uint32_t nd_here = CHUNKED ? (nd < steps ? nd : steps) : nd;      // (a chunk ends at `steps`; the rounds that pad a trip below must not go on)
    auto round_of_eight = [&](const uint2& d, uint32_t jb) __attribute__((always_inline)) {
        const unsigned long long d8 = (unsigned long long)d.x | ((unsigned long long)d.y << 32);
        // the eight probabilities first: they depend on the decisions only, so their LDS latency stays out of the interval's
        // dependency chain (a byte beyond the map's last decision may hold anything: its context is masked into the staged rows'
        // range, the value is never used)
        auto probability = [&](uint32_t q) __attribute__((always_inline)) {
            const uint32_t ctx = (uint32_t)(d8 >> (8u * q + 1u)) & 31u;
            return probs[(ctx < L ? ctx : 0u) * 64u + lane];
        };
        uint32_t r[8];
        if (jb + 8u <= nd_all && jb + 8u <= steps) {
            // every lane that codes at all has these eight decisions: no mask per step (a lane without a map computes on zeros and
            // stores nothing). Two thirds of the rounds of a Kodak batch: a map has at least one decision per symbol.
            double pq[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) pq[q] = probability(q);
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) r[q] = encode_step(s, pq[q], ((uint32_t)(d8 >> (8u * q)) & 1u) != 0u);
        } else {
            // the last rounds of the group's chain: a step only for the lanes that still have one
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                const Interval before = s;
                const uint32_t rec_q = encode_step(s, probability(q), ((uint32_t)(d8 >> (8u * q)) & 1u) != 0u);
                const bool has = jb + q < nd_here;
                s.lo = has ? s.lo : before.lo;
                s.hc = has ? s.hc : before.hc;
                r[q] = has ? rec_q : 0u;
            }
        }
        // four records per 16-byte store (up to three entries beyond the map's last decision: the pad of rcap)
        *reinterpret_cast<uint4*>(jb < nd_here ? rec + jb : sink) = make_uint4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<uint4*>(jb + 4u < nd_here ? rec + jb + 4u : sink) = make_uint4(r[4], r[5], r[6], r[7]);
    };
    // Three rounds per trip, three registers of decisions: each is asked for again the moment its round is through and is used
    // two rounds later (a register that is reloaded while its old value is still being used would be copied across the loop's back
    // edge, and a copy of a register in flight is a wait for it). A chain that is not a multiple of three rounds is padded with
    // empty ones (every step masked, the stores into the sink).
    uint2 da = decisions_of(j_first >> 3), db = decisions_of((j_first >> 3) + 1u), dc = decisions_of((j_first >> 3) + 2u);
    // (the first three arrive before the loop: what the loop's own waits count are then the steady state's six memory operations
    // between a request and its use, not the two of the first trip)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    for (uint32_t jb = j_first; jb < steps; jb += 24) {
        const uint32_t round = jb >> 3;
        round_of_eight(da, jb);
        da = decisions_of(round + 3u);
        round_of_eight(db, jb + 8u);
        db = decisions_of(round + 4u);
        round_of_eight(dc, jb + 16u);
        dc = decisions_of(round + 5u);
    }
    if (live) {
        if (retry) p.status[m] = RETRY;                 // the general kernel reproduces the exact code and stage
        else if (!CHUNKED || p.chunk + 1u == p.nchunks) rec[nd] = stop_record(s);      // BinaryArithmeticCoder.cpp:61-102; emitted by emit_kernel
        else p.enc_state[m] = make_uint2(s.lo, s.hc);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// (3) one wavefront per map: records -> the arithmetic-coded stream
// ---------------------------------------------------------------------------------------------------------------------
// Record j shifts n_j bits out (when n_j > 0): the first of them, then the pending-E3 queue as it stands (P_j copies of the
// complement, BinaryArithmeticCoder.cpp:322-337), then the other n_j - 1; afterwards the queue holds the record's own k_j
// scalings. So P_j = the k's of the records since (and including) the last one that shifted anything out, and the position of
// record j's bits is the sum of n + P over the records before it: two prefix sums and a running maximum per tile of 64 records.
template <bool CHUNKED>
__global__ __launch_bounds__(64) void emit_kernel(const SimdParams p) {
    EAE_KEEP_LAST_VGPR_FREE(EAE_RES_EMIT);
    __shared__ unsigned long long buf[kEmitWords + 2];
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row < 0 || p.status[m] != 0) return;
    uint32_t nrec = p.ndec[m] + 1u;                     // the stop record closes the stream
    const uint32_t* rec = p.records + (size_t)m * p.rcap;
    unsigned long long* out = reinterpret_cast<unsigned long long*>(p.streams + (uint64_t)m * p.stride);
    const uint32_t size_bits = round_up_to_byte(required_bits(p.map_size, p.L));
    for (uint32_t w = lane; w < kEmitWords + 2u; w += 64u) buf[w] = 0ull;
    uint32_t base = 0, pending = 0;                     // bits emitted so far; the queue in front of the tile
    bool give_up = false;
    uint32_t t_first = 0;
    const bool last_chunk = !CHUNKED || p.chunk + 1u == p.nchunks;
    if (CHUNKED) {
        // the records the encoder core's chunk has just left: [chunk * span, (chunk + 1) * span) of this map's decisions; the stop
        // record goes with the last chunk wherever the map's decisions end
        const uint32_t span = chunk_span(p.group_steps[m >> 6], p.nchunks);
        const uint32_t nd = nrec - 1u;
        t_first = p.chunk * span < nd ? p.chunk * span : nd;
        if (!last_chunk) nrec = t_first + span < nd ? t_first + span : nd;
        if (p.chunk != 0u) {
            const uint4 st = p.emit_state[m];
            base = st.x;
            pending = st.y;
            if (lane == 0) buf[0] = (unsigned long long)st.z | ((unsigned long long)st.w << 32);
        }
    }
    // A tile is 256 records, FOUR consecutive ones per lane (one 16-byte load): what a lane does for a record -- fields, the bit
    // string, two LDS ORs -- is per record whatever the tile, but the three wave scans, the flush and the bookkeeping are per TILE,
    // and they were three quarters of the pass with one record per lane (135 -> ~60 instructions per 64 records).
    for (uint32_t t = t_first; t < nrec; t += 4u * 64u) {
        const uint32_t j0 = t + 4u * lane;
        uint4 rr = make_uint4(0u, 0u, 0u, 0u);
        if (j0 < nrec) rr = *reinterpret_cast<const uint4*>(rec + j0);        // (the pad behind a map's records covers the tail of the load)
        uint32_t r[4] = {rr.x, rr.y, rr.z, rr.w};
        uint32_t n[4], k[4], kl[4];
        bool has[4];
        uint32_t ksum = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            if (j0 + q >= nrec) r[q] = 0u;
            n[q] = record_n(r[q]);
            k[q] = record_k(r[q]);
            has[q] = n[q] != 0u;
            kl[q] = ksum;                                   // scalings of this lane's records before record q
            ksum += k[q];
        }
        uint32_t ktot;
        const uint32_t e0 = wave_exclusive_scan(ksum, ktot);                             // scalings of the tile's records before this lane's
        // the running maximum of e over the records that shifted out = e of the LAST such record (e never decreases)
        int lane_last = -1;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) lane_last = has[q] ? (int)(e0 + kl[q]) : lane_last;
        const int below = wave_exclusive_max(lane_last);                                 // ... of the lanes below
        uint32_t c[4], cl[4], queue[4];
        uint32_t csum = 0;
        int since = below;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t e = e0 + kl[q];
            queue[q] = (since >= 0 ? e - (uint32_t)since : pending + e) + (record_is_stop(r[q]) ? 1u : 0u);
            c[q] = has[q] ? n[q] + queue[q] : 0u;
            cl[q] = csum;
            csum += c[q];
            since = has[q] ? (int)e : since;
        }
        uint32_t ctot;
        const uint32_t o0 = wave_exclusive_scan(csum, ctot);
        // the queue behind the tile: the k's from the last record that shifted out onwards
        const int tile_last = __builtin_amdgcn_readlane(since, 63);                      // lane 63's `since` has seen every record of the tile
        pending = tile_last >= 0 ? ktot - (uint32_t)tile_last : pending + ktot;
        const uint32_t start = base & 63u;              // bits of the carried partial word in buf[0]
        // the tile must fit the LDS buffer and the stream its capacity (Bitstream.cpp:32-35): otherwise the general kernel
        if (start + ctot > kEmitWords * 64u || base + ctot > size_bits) { give_up = true; break; }
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            if (has[q]) {
                const uint32_t lead = record_leaving(r[q]);
                const unsigned long long first = lead >> 31;
                // the other n - 1 leaving bits, first in time at bit 0
                const unsigned long long rest = (unsigned long long)(__builtin_bitreverse32(lead << 1) & ((1u << (n[q] - 1u)) - 1u));
                const uint32_t pos = start + o0 + cl[q];
                const uint32_t qu = queue[q];
                if (c[q] <= 64u) {
                    // first bit, `queue` complements, the rest: at most 64 bits in one piece (queue <= 63 here)
                    const unsigned long long run = first ? 0ull : (((qu < 63u ? (1ull << qu) : (1ull << 63)) - 1ull) | (qu == 63u ? (1ull << 62) : 0ull));
                    const unsigned long long v = first | (run << 1) | (qu + 1u < 64u ? rest << (qu + 1u) : 0ull);
                    const uint32_t sh = pos & 63u;
                    atomicOr(&buf[pos >> 6], v << sh);
                    if (sh + c[q] > 64u) atomicOr(&buf[(pos >> 6) + 1u], v >> (64u - sh));
                } else {
                    // a long queue (nearly dead maps under a very skewed first probability): the first bit, the run word by word, the rest
                    if (first) atomicOr(&buf[pos >> 6], 1ull << (pos & 63u));
                    if (!first) {
                        uint32_t b0 = pos + 1u;
                        const uint32_t b1 = pos + 1u + qu;            // ones over [b0, b1)
                        while (b0 < b1) {
                            const uint32_t sh = b0 & 63u;
                            const uint32_t cnt = (b1 - b0) < (64u - sh) ? (b1 - b0) : (64u - sh);
                            const unsigned long long ones = cnt == 64u ? ~0ull : (((1ull << cnt) - 1ull) << sh);
                            atomicOr(&buf[b0 >> 6], ones);
                            b0 += cnt;
                        }
                    }
                    const uint32_t pr = pos + 1u + qu, sh = pr & 63u;
                    if (rest) {
                        atomicOr(&buf[pr >> 6], rest << sh);
                        if (sh + (n[q] - 1u) > 64u) atomicOr(&buf[(pr >> 6) + 1u], rest >> (64u - sh));
                    }
                }
            }
        }
        // complete words to the stream, the partial one stays as the next tile's carry
        const uint32_t nfull = (start + ctot) >> 6;
        for (uint32_t w = lane; w < nfull; w += 64u) out[(base >> 6) + w] = buf[w];
        const unsigned long long carry = buf[nfull];
        for (uint32_t w = lane; w <= nfull; w += 64u) buf[w] = 0ull;
        if (lane == 0) buf[0] = carry;
        base += ctot;
    }
    if (lane == 0) {
        if (give_up) {
            p.status[m] = RETRY;
        } else if (last_chunk) {
            if (base & 63u) out[base >> 6] = buf[0];     // Bitstream flush of the partial word (zeros above the last bit)
            p.bac_bits[m] = base;
        } else {
            const unsigned long long carry = buf[0];
            p.emit_state[m] = make_uint4(base, pending, (uint32_t)carry, (uint32_t)(carry >> 32));
            p.avail_bits[m] = base & ~63u;               // the whole words are in memory: what a trailing decoder may read
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// (4) 64 maps per wavefront: decode the decisions; one prefix byte per symbol
// ---------------------------------------------------------------------------------------------------------------------
// LDS of a block: scaled probabilities [L + 1][64] doubles (one row beyond L: read ahead of an escape, never used), then the ring
// [kRing][64] words.
// (one row beyond the ring: row kRing mirrors row 0, so that words w and w + 1 are always rows r and r + 1 for the unchecked rounds)
constexpr size_t decode_lds_bytes(uint32_t L) { return ((size_t)L + 1u) * 64u * sizeof(double) + ((size_t)kRing + 1u) * 64u * sizeof(uint32_t); }

#ifdef EAE_HWID_PROBE      // scratch/r03_hwid_probe.py: does a long-lived coder wave ever resume on another CU / SIMD / slot (context save / restore)?
__device__ unsigned int g_hwid_probe[8];     // [0] waves, [1] waves whose HW_ID[15:0] or XCC_ID changed, [2..5] an example (before, after)
#endif

// CHUNKED (eae_hip_coder_roundtrip_trailing): the launch decodes as far as the stream has reached memory -- `avail_bits`, left by the
// emit pass of the encoder's latest chunk -- and parks every lane's state (interval, code register, bits taken, unary count, symbol
// index); the next launch rebuilds ring and window at that bit position and goes on. A step takes at most 30 bits, and the lane
// only steps while 46 more bits are known to be there (30 + the 16 the code register looks ahead), so nothing unwritten is ever
// taken; the last launch runs on the complete stream (`bac_bits`) exactly like the unchunked kernel.
constexpr uint32_t kNotStarted = 0xFFFFFFFFu;
constexpr uint32_t kStepMargin = 46u;

template <bool CHUNKED>
__global__ __launch_bounds__(64) void bac_decode_core_kernel(const SimdParams p) {
    if (CHUNKED) { EAE_KEEP_LAST_VGPR_FREE(EAE_RES_DECODE_CORE_CHUNKED); } else { EAE_KEEP_LAST_VGPR_FREE(EAE_RES_DECODE_CORE); }
    __builtin_amdgcn_s_setprio(EAE_SIMD_PRIO);
#ifdef EAE_HWID_PROBE
    const unsigned int probe_hw0 = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const unsigned int probe_xcc0 = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
#endif
    const uint32_t lane = threadIdx.x;
    const uint32_t m = blockIdx.x * 64u + lane;
    const bool in_range = m < p.n_maps;
    const int32_t row = in_range ? (p.prob_row ? p.prob_row[m] : (int32_t)m) : -1;
    const uint32_t L = p.L;
    double* probs = lds_dyn;
    uint32_t* ring = reinterpret_cast<uint32_t*>(lds_dyn + ((size_t)L + 1u) * 64u) + lane;      // row w of this lane: ring[(w & 31) * 64]
    const bool live = in_range && row >= 0 && p.status[m] == 0;
    const bool final_chunk = !CHUNKED || p.chunk + 1u == p.nchunks;
    const uint32_t nbac = live ? (final_chunk ? p.bac_bits[m] : p.avail_bits[m]) : 0u;
    bool retry = false;
    if (live && (nbac > p.stride * 4u || p.bypass_bits[m] > p.stride * 4u)) retry = true;       // beyond the buffer: not a stream of ours
    if (live) {
        for (uint32_t k = 0; k < L; k++) {
            const double pk = p.probs[(size_t)row * L + k];
            probs[k * 64u + lane] = scale_probability(pk);
            if (!(pk > 0. && pk < 1.)) retry = true;     // only an error if that context is decoded: general kernel
        }
        probs[L * 64u + lane] = scale_probability(0.5);
    }
    const uint32_t size = live && !retry ? p.map_size : 0u;
    const uint4* src = reinterpret_cast<const uint4*>(p.streams + (uint64_t)(in_range ? m : 0u) * p.stride);
    const uint32_t nwords = size ? (nbac + 31u) >> 5 : 0u;          // words that hold stream bits; everything beyond reads as zero
    // where this launch picks the stream up: bit 0, or where the last chunk parked the lane
    uint4 parked = make_uint4(0u, 0u, 0u, kNotStarted);
    uint2 parked2 = make_uint2(0u, 0u);
    if (CHUNKED && size) { parked = p.dec_state[m]; parked2 = p.dec_state2[m]; }
    const bool started = CHUNKED && parked.w != kNotStarted;
    const uint32_t taken0 = started ? parked.w : 0u;      // stream bits taken so far
    const uint32_t word0 = taken0 >> 5, wbase = word0 & ~3u;
#ifdef EAE_DECODE_TOPUP_ZEROS
    // The FIRST form of this kernel (commit 377df1b), kept buildable (never shipped: scratch/variant.sh, EAE_HIP_LIB) because it
    // passed every stand-alone test and derailed next to MFMA kernels: words beyond the end of the stream are
    // requested too and land as zeros in rows the lane has already consumed; bit reversal at the fetch, which is waited for at once.
    auto fetch = [&](uint32_t w0) {                                  // w0: a multiple of 4
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (w0 < nwords) v = src[w0 >> 2];
        v.x = w0 + 0u < nwords ? __builtin_bitreverse32(v.x) : 0u;
        v.y = w0 + 1u < nwords ? __builtin_bitreverse32(v.y) : 0u;
        v.z = w0 + 2u < nwords ? __builtin_bitreverse32(v.z) : 0u;
        v.w = w0 + 3u < nwords ? __builtin_bitreverse32(v.w) : 0u;
        return v;
    };
    auto land = [&](uint32_t w0, const uint4& v) {
        ring[((w0 + 0u) & (kRing - 1u)) * 64u] = v.x;
        ring[((w0 + 1u) & (kRing - 1u)) * 64u] = v.y;
        ring[((w0 + 2u) & (kRing - 1u)) * 64u] = v.z;
        ring[((w0 + 3u) & (kRing - 1u)) * 64u] = v.w;
    };
#pragma unroll
    for (uint32_t w0 = 0; w0 < kRing; w0 += 4u) land(w0, fetch(w0));
#else
    // Four stream words from memory (raw), requested a checkpoint ahead of their use; only groups that hold stream bits are ever
    // requested (w0 < nwords): what a lane reads beyond the end of its stream -- the tail of the last group, older words still in
    // the ring -- is never taken (`left` bounds every take).
    auto fetch = [&](uint32_t w0) {                                  // w0: a multiple of 4
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (w0 < nwords) v = src[w0 >> 2];
        return v;
    };
    // ... and into the ring, bit-reversed (the next bit in time most significant)
    auto land = [&](uint32_t w0, const uint4& v) {
        ring[((w0 + 0u) & (kRing - 1u)) * 64u] = __builtin_bitreverse32(v.x);
        ring[((w0 + 1u) & (kRing - 1u)) * 64u] = __builtin_bitreverse32(v.y);
        ring[((w0 + 2u) & (kRing - 1u)) * 64u] = __builtin_bitreverse32(v.z);
        ring[((w0 + 3u) & (kRing - 1u)) * 64u] = __builtin_bitreverse32(v.w);
        if ((w0 & (kRing - 1u)) == 0u) ring[kRing * 64u] = __builtin_bitreverse32(v.x);      // row kRing mirrors row 0
    };
    // the ring starts full: words wbase .. wbase + 31
    {
        uint4 first[kRing / 4u];
#pragma unroll
        for (uint32_t g = 0; g < kRing / 4u; g++) first[g] = fetch(wbase + 4u * g);
#pragma unroll
        for (uint32_t g = 0; g < kRing / 4u; g++) land(wbase + 4u * g, first[g]);
    }
#endif
    uint32_t loaded = wbase + kRing;      // words [.., loaded) have been in the ring
    // the window: the next `rcount` stream bits, left-aligned (the next bit in time at bit 63)
    unsigned long long rwin = (((unsigned long long)ring[(word0 & (kRing - 1u)) * 64u] << 32) |
                               (unsigned long long)ring[((word0 + 1u) & (kRing - 1u)) * 64u]) << (taken0 & 31u);
    uint32_t rcount = 64u - (taken0 & 31u), rword = word0 + 2u;
    uint32_t left = nbac > taken0 ? nbac - taken0 : 0u;      // stream bits not yet taken
    uint32_t code32 = parked.z;
    // Bac::start_decoding (BinaryArithmeticCoder.cpp:104-122): 16 bits, the last one repeated once the stream is exhausted
    bool begun = started;
    if (!started && (final_chunk || left >= kStepMargin)) {
        const uint32_t k = left < 16u ? left : 16u;
        uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - k));
        const uint32_t sticky = bits & 1u;
        bits = (bits << (16u - k)) | (sticky ? ((1u << (16u - k)) - 1u) : 0u);
        rwin <<= k;
        rcount -= k;
        left -= k;
        code32 = bits << 16;
        begun = true;
    }
    Interval s = interval_init();
    const double p0 = probs[lane];
    double pk = p0;
    uint32_t unary = 0, i = 0;
    if (started) {
        s.lo = parked.x;
        s.hc = parked.y;
        unary = parked2.x;
        i = parked2.y;
        pk = probs[unary * 64u + lane];
    }
    uint8_t* prefix = p.prefixes + (size_t)(in_range ? m : 0u) * p.map_size;
    const uint32_t steps_left_any = wave_max(size);       // 0: nothing to do in this block
    uint4 fa = make_uint4(0u, 0u, 0u, 0u), fb = fa;        // eight words on their way from memory
    bool flying = false;
    if (steps_left_any) {
        for (;;) {
            // ---- checkpoint, every 8 steps: the words requested at the last checkpoint enter the ring (their rows hold words this
            // lane has already moved to its window: it was at most 24 words behind `loaded` when they were requested), and the next
            // eight are requested when this lane is at most 24 words behind and the stream has words left. A step takes at most 30
            // bits, so the ring cannot run dry of stream words.
            if (flying) {
                land(loaded, fa);
                land(loaded + 4u, fb);
                loaded += 8u;
            }
#ifdef EAE_DECODE_TOPUP_ZEROS
            flying = loaded - rword <= kRing - 8u;
#else
            // (two words less than the ring allows: the unchecked rounds read words rword - 2 and rword - 1 from the ring again)
            flying = loaded - rword <= kRing - 10u && loaded < nwords;
#endif
            if (flying) {
                fa = fetch(loaded);
                fb = fetch(loaded + 4u);
            }
            // A round in which every decoding lane has eight symbols and eight steps' worth of stream left (a step takes at most 30
            // bits) runs without the per-step masks: no `i < size`, no end-of-stream extension, and the prefix byte is stored every
            // step -- the running count, overwritten until the symbol's last decision leaves the final value (a lane's stores to
            // one address stay in order). All but the last rounds of a map.
#ifdef EAE_DECODE_TOPUP_ZEROS
            bool fast_round = false;                        // the first form of this kernel stays as it was built (tests/test_isa_guard.py)
#else
            // (a step takes 30 bits at most and a fraction of a bit on average -- the whole stream of a nearly dead map is a few dozen
            // bits --: the round is run on the bet that the bits are there and undone if a lane ends it having taken more than its
            // stream holds, which happens at the end of streams only)
            bool fast_round = !__any(size != 0u && i + 8u > size) && (!CHUNKED || final_chunk || !__any(size != 0u && !begun));
#endif
            if (fast_round) {
                if (size) {
                    // The stream as a bit position instead of a window: the next 32 bits are words w and w + 1 of the ring (rows r, r + 1:
                    // one ds_read2st64) shifted by the position's low five bits -- no conditional top-up, no window to shift and
                    // count (8 instructions where the window takes 19). The window form is rebuilt behind the round for the
                    // checkpoint and the checked rounds.
                    const uint32_t bp0 = rword * 32u - rcount;
                    uint32_t bp = bp0;
                    const Interval s0 = s;
                    const uint32_t code0 = code32, unary0 = unary, i0 = i;
                    const double pk0 = pk;
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) {
                        const uint32_t row = (bp >> 5) & (kRing - 1u);
                        const uint32_t hi = ring[row * 64u], lo = ring[row * 64u + 64u];
                        const double pspec = probs[(unary + 1u) * 64u + lane];
                        const DecodeStep d = decode_step(s, code32, pk);
                        const uint32_t win = (uint32_t)(((((unsigned long long)hi << 32) | (unsigned long long)lo) << (bp & 31u)) >> 32);
                        const uint32_t bits = (win >> 1) >> (31u - d.take);                 // the step's d.take <= 30 bits, first in time on top
                        bp += d.take;
                        code32 = shift_code(code32, d, bits);
                        const uint32_t count = unary + (d.one ? 1u : 0u);       // a one: unary + 1 (= L when it ends the symbol); a zero: unary
                        prefix[i] = (uint8_t)count;
                        const bool done = !d.one || count == L;
                        i += done ? 1u : 0u;
                        unary = done ? 0u : count;
                        pk = done ? p0 : pspec;
                    }
                    if (__any(bp - bp0 > left)) {
                        // the bet is lost for some lane (the end of its stream): every lane goes back -- the window was not touched, it
                        // is only rebuilt below -- and the round runs step by step with the checks
                        s = s0; code32 = code0; unary = unary0; i = i0; pk = pk0;
                        fast_round = false;
                    } else {
                        left -= bp - bp0;
                        const uint32_t row = (bp >> 5) & (kRing - 1u);
                        rwin = (((unsigned long long)ring[row * 64u] << 32) | (unsigned long long)ring[row * 64u + 64u]) << (bp & 31u);
                        rcount = 64u - (bp & 31u);
                        rword = (bp >> 5) + 2u;
                    }
                }
            }
            if (!fast_round) {
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                if (i < size && (!CHUNKED || final_chunk || (begun && left >= kStepMargin))) {
                    // top the window up: 32 more bits once at most 32 are left (the read is unconditional, its use is not)
                    const uint32_t wnext = ring[(rword & (kRing - 1u)) * 64u];
                    const double pspec = probs[(unary + 1u) * 64u + lane];
                    const bool need = rcount <= 32u;
                    rwin |= (unsigned long long)(need ? wnext : 0u) << (need ? 32u - rcount : 0u);
                    rcount += need ? 32u : 0u;
                    rword += need ? 1u : 0u;
                    // Bac::decode (BinaryArithmeticCoder.cpp:124-134, 254-320), lean_step.h
                    const DecodeStep d = decode_step(s, code32, pk);
                    // the d.take stream bits of this step; beyond the end of the stream the last real bit of THIS step repeats,
                    // and no bit at all reads as 0 (the `storage` of rescale_decoding)
                    const uint32_t k = d.take < left ? d.take : left;
                    uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - k));
                    const uint32_t ext = d.take - k;
                    bits = (bits << ext) | ((bits & 1u) ? ((1u << ext) - 1u) : 0u);
                    rwin <<= k;
                    rcount -= k;
                    left -= k;
                    code32 = shift_code(code32, d, bits);
                    // binarisation state (LosslessCoder.cpp:193-230): a one advances the unary count up to L, a zero ends it
                    const bool done = !d.one || unary + 1u == L;
                    if (done) prefix[i] = (uint8_t)(d.one ? L : unary);
                    i += done ? 1u : 0u;
                    unary = done ? 0u : unary + 1u;
                    pk = done ? p0 : pspec;
                }
            }
            }
            if (!__any(i < size && (!CHUNKED || final_chunk || (begun && left >= kStepMargin)))) break;
        }
    }
    if (CHUNKED && size && !final_chunk) {
        p.dec_state[m] = make_uint4(s.lo, s.hc, code32, begun ? nbac - left : kNotStarted);
        p.dec_state2[m] = make_uint2(unary, i);
    }
    if (live && retry) p.status[m] = RETRY;
#ifdef EAE_HWID_PROBE
    {
        const unsigned int hw1 = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);
        const unsigned int xcc1 = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
        if (lane == 0) {
            atomicAdd(&g_hwid_probe[0], 1u);
            if (((hw1 ^ probe_hw0) & 0xFFFFu) != 0u || xcc1 != probe_xcc0) {
                atomicAdd(&g_hwid_probe[1], 1u);
                g_hwid_probe[2] = probe_hw0; g_hwid_probe[3] = probe_xcc0; g_hwid_probe[4] = hw1; g_hwid_probe[5] = xcc1;
            }
        }
    }
#endif
}

#ifdef EAE_HWID_PROBE
extern "C" int eae_hip_debug_hwid_probe(unsigned int* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_hwid_probe), 8 * sizeof(unsigned int));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// (5) one wavefront per map: prefixes + bypass stream -> symbols; compare with the encoder's input
// ---------------------------------------------------------------------------------------------------------------------
template <bool STAGED>
__global__ __launch_bounds__(64) void debinarise_kernel(const SimdParams p) {
    if (STAGED) { EAE_KEEP_LAST_VGPR_FREE(EAE_RES_DEBINARISE_STAGED); } else { EAE_KEEP_LAST_VGPR_FREE(EAE_RES_DEBINARISE); }
    __shared__ uint32_t ytile[80];                     // the bypass words a tile of 64 symbols can touch: 64 x 34 bits + alignment
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (lane == 0 && p.ndec) p.ndec[m] = 0u;           // 1: handed to the general kernel (compare_kernel looks at it afterwards)
    if (row < 0 || p.status[m] != 0) {
        if (row >= 0 && p.status[m] == RETRY && lane == 0 && p.ndec) p.ndec[m] = 1u;
        return;
    }
    const uint32_t L = p.L, size = p.map_size;
    const uint32_t nbyp = p.bypass_bits[m];
    const uint32_t* gbyp = reinterpret_cast<const uint32_t*>(p.streams + (uint64_t)m * p.stride + p.stride / 2);
    const uint32_t ywords = (nbyp + 31u) >> 5;
    const uint8_t* prefix = p.prefixes + (size_t)m * size;
    const int16_t* expected = p.symbols ? p.symbols + (size_t)m * size : nullptr;
    int16_t* out = p.decoded ? p.decoded + (size_t)m * size : nullptr;
    uint32_t ybase = 0;                                // bypass bits consumed so far
    bool bad = false, differ = false;
    // STAGED (small steps: eae_hip_coder_decode_batch): everything the tile loop reads comes through LDS -- the first kBypassStage
    // words of the bypass stream once (a map whose bypass stream is longer reads the rest tile by tile from memory), prefix bytes
    // and expected symbols 1,024 symbols at a time, loads clamped instead of masked, eight per lane in flight. With its loads inside
    // the tile loop a tile begins by waiting for them and -- the compiler can only wait for everything in flight -- for the stores
    // of the tile before: 1 us per tile, 25 us for the 24 tiles of a Kodak map where the staged form takes 16. For large batches
    // (thousands of these wavefronts at once, their latency hidden by each other) the plain form is kept: the staged one's 5.6 KB of
    // LDS and 40 registers cost conv_2 3 % when it runs beside it (profiles/r06_coder_cores.md).
    __shared__ uint32_t ystage[STAGED ? kBypassStage + 80u : 1u];
    __shared__ uint8_t pstage[STAGED ? kDebinariseStage : 1u];
    __shared__ int16_t estage[STAGED ? kDebinariseStage : 1u];
    if (STAGED) {
        const uint32_t have = ywords < kBypassStage ? ywords : kBypassStage, last = ywords ? ywords - 1u : 0u;
        for (uint32_t k = 0; k < have; k += 512u) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t w = k + u * 64u + lane;
                v[u] = gbyp[w < last ? w : last];
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t w = k + u * 64u + lane;
                if (w < kBypassStage) ystage[w] = w < ywords ? v[u] : 0u;
            }
        }
        // zero beyond the stream (what a tile's window may still read), up to the end of the staging area
        for (uint32_t w = have + lane; w < kBypassStage + 80u; w += 64u) ystage[w] = 0u;
    }
    for (uint32_t t = 0; t < size; t += 64u) {
        if (STAGED && (t & (kDebinariseStage - 1u)) == 0u) {
            const uint32_t last = size - 1u;
            __syncthreads();                                   // (one wavefront: the reads of the last chunk are before these writes)
            for (uint32_t k = 0; k < kDebinariseStage && t + k < size; k += 512u) {
                uint8_t pv[8];
                int16_t ev[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const uint32_t i = t + k + u * 64u + lane, at_i = i < last ? i : last;
                    pv[u] = prefix[at_i];
                    ev[u] = expected ? expected[at_i] : (int16_t)0;
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    pstage[k + u * 64u + lane] = pv[u];
                    estage[k + u * 64u + lane] = ev[u];
                }
            }
            __syncthreads();
        }
        // the window of the bypass stream this tile can touch, zero beyond the stream: inside the staged words, or (long streams)
        // loaded for this tile
        const uint32_t w0 = ybase >> 5;
        // (a window is words w0 .. w0 + 79: staged, or zero behind a stream that was staged whole)
        const bool in_stage = STAGED && (ywords <= kBypassStage ? w0 <= kBypassStage : w0 + 80u <= kBypassStage);
        if (!in_stage) {
            ytile[lane] = w0 + lane < ywords ? gbyp[w0 + lane] : 0u;
            if (lane < 16u) ytile[64u + lane] = w0 + 64u + lane < ywords ? gbyp[w0 + 64u + lane] : 0u;
        }
        const uint32_t* const window = in_stage ? ystage + w0 : ytile;
        auto bits_at = [&](uint32_t pos) {             // 32 stream bits from position `pos`, the first in time at bit 0
            const uint32_t rel = pos - (w0 << 5), q = rel >> 5, sh = rel & 31u;
            const unsigned long long two = (unsigned long long)window[q] | ((unsigned long long)window[q + 1u] << 32);
            return (uint32_t)(two >> sh);
        };
        const uint32_t i = t + lane;
        const bool valid = i < size;
        uint32_t a = valid ? (uint32_t)(STAGED ? pstage[i & (kDebinariseStage - 1u)] : prefix[i]) : 0u;
        const bool nonzero = a != 0u;
        uint32_t ntot;
        uint32_t at = wave_exclusive_scan(nonzero ? 1u : 0u, ntot);      // sign bits of the symbols below this lane ...
        uint32_t extra = 0;                                             // ... and Exp-Golomb codes up to and including this lane's
        unsigned long long escapes = __ballot(valid && a == L);
        uint32_t extra_tot = 0;
        while (escapes) {
            // one escape of the tile at a time, in symbol order (its position depends on the lengths of the ones before it);
            // every lane reads the same bits (LosslessCoder.cpp:113-165): nn ones, a zero, nn suffix bits (most significant first)
            const int e = __builtin_ctzll(escapes);
            escapes &= escapes - 1ull;
            const uint32_t pos = ybase + (uint32_t)__builtin_amdgcn_readlane((int)at, e) + extra_tot;
            const uint32_t w = bits_at(pos);
            const uint32_t nn = (uint32_t)__builtin_ctz(~w | 0x80000000u);
            if (nn > 16u || pos + 2u * nn + 1u > nbyp) { bad = true; break; }     // malformed or truncated: general kernel
            const uint32_t suffix = nn ? __builtin_bitreverse32(bits_at(pos + nn + 1u)) >> (32u - nn) : 0u;
            const uint32_t value = (L + ((suffix + (1u << nn) - 1u) & 0xFFFFu)) & 0xFFFFu;      // uint16 arithmetic of the reference
            const uint32_t len = 2u * nn + 1u;
            if ((int)lane == e) a = value;
            if ((int)lane >= e) extra += len;
            extra_tot += len;
        }
        if (bad) break;
        // the sign of a non-zero symbol follows its Exp-Golomb code (LosslessCoder.cpp:39-56, 254-276): 0 = negative
        const uint32_t spos = ybase + at + extra;
        if (nonzero && spos >= nbyp) bad = true;        // resource_error: the general kernel reports it
        if (__any(bad)) { bad = true; break; }
        int v = (int)(int16_t)a;
        if (nonzero && !(bits_at(spos) & 1u)) v = -v;
        if (valid) {
            if (out) out[i] = (int16_t)v;
            if (expected && (STAGED ? estage[i & (kDebinariseStage - 1u)] : expected[i]) != (int16_t)v) differ = true;
        }
        ybase += ntot + extra_tot;
    }
    const bool any_differ = __any(differ);
    if (lane == 0) {
        if (bad) { p.status[m] = RETRY; if (p.ndec) p.ndec[m] = 1u; }
        else if (expected && any_differ) p.status[m] = MISMATCH;
    }
}

// decoded == encoded for the maps the general kernel decoded (debinarise_kernel compares its own), one wavefront per map
__global__ __launch_bounds__(64) void compare_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    if (p.ndec[m] == 0u) return;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row < 0 || p.status[m] != 0) return;
    const int16_t* a = p.symbols + (size_t)m * p.map_size;
    const int16_t* b = p.decoded + (size_t)m * p.map_size;
    int differ = 0;
    for (uint32_t i = lane; i < p.map_size; i += 64) differ |= (a[i] != b[i]);
    if (__any(differ) && lane == 0) p.status[m] = MISMATCH;
}

// =====================================================================================================================
// EXPERIMENTAL, not in the product library (round 6): everything between here and the matching #endif, and the two round-trip
// entry points at the end of this file, is compiled only with -DEAE_EXPERIMENTAL_CODER (lib/libeae_hip_test.so, csrc/Makefile).
// Both forms are byte-exact and both lose to the plain encode_batch + decode_batch pair on this runtime (numbers below and in
// DESIGN.md section 5); they are kept, with their tests, as the measured record of why. The <CHUNKED = true> instances of the
// kernels above exist only where eae_hip_coder_roundtrip_trailing instantiates them, i.e. in that build.
// =====================================================================================================================
#ifdef EAE_EXPERIMENTAL_CODER
// ---------------------------------------------------------------------------------------------------------------------
// (6) the three serial stages as ONE workgroup per 64 maps, pipelined through LDS: encoder core -> bit writer -> decoder core
// ---------------------------------------------------------------------------------------------------------------------
// For one or two images the coder's cost is the LENGTH of its serial chains laid end to end (encoder core 0.29 ms, emit 0.02,
// decoder core 0.33 of the 1.2 ms one Kodak image takes). Cut into launches on three streams the chains do overlap, but a hop
// between streams costs 60-100 us on this runtime (eae_hip_coder_roundtrip_trailing, kept as an option). Here the three stages of
// a group of 64 maps are three wavefronts of one workgroup -- each a 64-lanes-in-step serial chain on a SIMD of its own -- and what
// passes between them passes through LDS:
//   wave 0, the encoder core of bac_encode_core_kernel: its records go into a ring of kPipeRecords per lane instead of to memory;
//   wave 1, the BIT WRITER: one map per lane like the cores, not one map per wavefront like emit_kernel (that pass is prefix sums
//           over 64 records of ONE map: 2.1 instructions per record and map, 135 per step of a group; in step over 64 maps the
//           same work is ~45): it appends a record's first bit, the pending E3 run and the rest to a 64-bit accumulator and sends
//           whole 32-bit words to the stream in memory AND, bit-reversed, into the decoder's ring;
//   wave 2, the decoder core of bac_decode_core_kernel, its ring filled by wave 1 instead of from memory: it steps while 46 more
//           bits are known to be there (as in the chunked form) and runs on the exact bit count once the writer has flushed.
// Flow control is three counters in LDS (records written; the slowest writer lane; per lane the words written / taken) read once
// per round of eight steps; no stage can wait in a circle (the encoder only waits for the slowest writer lane, a writer lane only
// for room in a ring whose reader is then not starved, a reader lane only for bits), every wait is bounded, and a wait that
// expires aborts the workgroup: its maps go to the general kernel like anything else these kernels do not finish (long pending
// runs, a stream that outgrows its region, an invalid probability). The records never see memory; emit_kernel's 64 wavefronts
// per group are one.
// Measured (round 5, one Kodak image's maps alone on the GPU, profiles/r05_fused_*.log; -DEAE_PIPE_PROBE counts cycles per role):
// same bytes in every test, and NOT faster than the kernels it replaces -- 0.97 ms against 0.64 at 0.19 bpp, 2.45 against 2.33 at
// 2 bpp. (a) At the headline's entropy a map's whole stream is 30-500 bits and the decoder looks 46 bits ahead: for the sparse
// maps of a group nothing can be decoded before the stream is complete, and 64 lanes in step wait for the sparsest (the decoder
// had done 27 of its 216 rounds when the writer finished). (b) At 2 bpp the stages do overlap (597 of 791 decoder rounds without
// per-step checks, not one lost bet) but the bit writer, ~100 vector and mask instructions per record = 700 cycles, is slower than
// either core and both wait for it (the encoder core 54 % of its time). So eae_hip_coder_roundtrip_fused is an entry point with
// its tests, not what BatchCodec launches; a writer at the cores' 300 cycles per record would make it 1.5 against 2.33 ms at 2 bpp.
constexpr uint32_t kPipeRecords = 64u;            // records per lane between encoder core and bit writer (a power of two)
constexpr uint32_t kPipeSpinLimit = 1u << 21;     // sleeps before a wait gives up (~1 s): a bug or a dead neighbour, never a normal wait
constexpr uint32_t kPipeDone = 0x80000000u, kPipeBad = 0x40000000u;

// Everything the three wavefronts tell each other lives in LDS, and the LDS serves the requests of a CU in the order they arrive,
// a wavefront's own in program order: a counter written after the data it covers is seen after that data by whoever reads the
// counter first and the data second. So no fence (a workgroup-scope fence also waits for every global store in flight -- the
// stream words, the prefix bytes: 1-2 us per round of eight steps, which made the first form of this kernel slower than the three
// kernels it replaces): only the compiler must keep the order, and the LDS counter must be drained where a wave spins on it.
#define EAE_PIPE_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr size_t pipe_lds_bytes(uint32_t L) {
    return ((size_t)L + 1u) * 64u * sizeof(double) + ((size_t)kPipeRecords + kRing) * 64u * sizeof(uint32_t) + (8u + 4u * 64u) * sizeof(uint32_t);
}

#ifndef EAE_RES_PIPE
#define EAE_RES_PIPE 63
#endif

#ifdef EAE_PIPE_PROBE      // scratch/r05/pipe_probe.py: where the three wavefronts of block 0 spend their cycles
__device__ unsigned long long g_pipe_probe[3][4];      // per role: cycles in all, cycles waiting, rounds, HW_ID
__device__ unsigned long long g_pipe_probe2[8];        // decoder: rounds and cycles when every stream was complete, fast rounds, lost bets, waits
#define PIPE_T0() const long long probe_t0 = clock64(); long long probe_wait = 0, probe_mark = 0; unsigned probe_rounds = 0
#define PIPE_WAIT_BEGIN() probe_mark = clock64()
#define PIPE_WAIT_END() probe_wait += clock64() - probe_mark
#define PIPE_ROUND() probe_rounds++
#define PIPE_REPORT(role_) if (blockIdx.x == 0 && lane == 0) { g_pipe_probe[role_][0] = clock64() - probe_t0; g_pipe_probe[role_][1] = probe_wait; \
    g_pipe_probe[role_][2] = probe_rounds; g_pipe_probe[role_][3] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4); }
#else
#define PIPE_T0()
#define PIPE_WAIT_BEGIN()
#define PIPE_WAIT_END()
#define PIPE_ROUND()
#define PIPE_REPORT(role_)
#endif

__global__ __launch_bounds__(256) void coder_pipe_kernel(const SimdParams p) {
    EAE_KEEP_LAST_VGPR_FREE(EAE_RES_PIPE);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      // 0 encoder core, 1 bit writer, 2 decoder core, 3 nothing
    const uint32_t m = blockIdx.x * 64u + lane;
    const bool in_range = m < p.n_maps;
    const int32_t row = in_range ? (p.prob_row ? p.prob_row[m] : (int32_t)m) : -1;
    const uint32_t L = p.L;
    double* probs = lds_dyn;                                                              // [L + 1][64], scaled (lean_step.h)
    uint32_t* recring = reinterpret_cast<uint32_t*>(lds_dyn + ((size_t)L + 1u) * 64u);     // [kPipeRecords][64]
    uint32_t* wring = recring + kPipeRecords * 64u;                                       // [kRing][64] stream words, bit-reversed
    volatile uint32_t* ctl = wring + kRing * 64u;       // [0] records every coding lane has, [1] stop records written, [2] slowest writer lane, [3] abort
    volatile uint32_t* lane_ok = ctl + 8;               // the lane codes a map here (valid probabilities, no earlier status)
    volatile uint32_t* wr_words = lane_ok + 64;         // stream words the writer has sent | kPipeDone (stream complete) | kPipeBad (handed over)
    volatile uint32_t* wr_bits = wr_words + 64;         // bits of a complete stream
    volatile uint32_t* rd_word = wr_bits + 64;          // first stream word the decoder has not moved into its window yet
    const bool live = in_range && row >= 0 && p.status[m] == 0;
    __builtin_amdgcn_s_setprio(EAE_SIMD_PRIO);
    if (role == 0u) {
        bool good = live;
        if (live) {
            for (uint32_t k = 0; k < L; k++) {
                const double pk = p.probs[(size_t)row * L + k];
                probs[k * 64u + lane] = scale_probability(pk);
                if (!(pk > 0. && pk < 1.)) good = false;       // the general kernel names the error if that context is ever coded
            }
            probs[L * 64u + lane] = scale_probability(0.5);
        }
        lane_ok[lane] = good ? 1u : 0u;
        wr_words[lane] = 0u;
        wr_bits[lane] = 0u;
        rd_word[lane] = 0u;
        if (lane < 8u) ctl[lane] = 0u;
    }
    __syncthreads();
    if (role == 3u) return;
    const bool good = lane_ok[lane] != 0u;
    const uint32_t nd = good ? p.ndec[m] : 0u;

    PIPE_T0();
    if (role == 0u) {
        // ---- encoder core: bac_encode_core_kernel, records into the ring
        const uint32_t steps = wave_max(nd);
        const uint32_t nd_all = ~wave_max(~(nd ? nd : 0xFFFFFFFFu));
        const uint8_t* dec = p.decisions + (size_t)blockIdx.x * 64u * p.dcap + (size_t)lane * 8u;
        Interval s = interval_init();
        uint2 ahead = steps ? *reinterpret_cast<const uint2*>(dec) : make_uint2(0, 0);
        bool gave_up = false;
        // room: records [.., upto) must not pass the slowest writer lane by more than the ring (the writer publishes that lane once
        // per round; lanes that only wait for their stop record do not count)
        auto wait_for_room = [&](uint32_t upto) {
            uint32_t spins = 0;
            for (;;) {
                EAE_PIPE_ORDER();
                const uint32_t slowest = ctl[2];
                if (slowest == 0xFFFFFFFFu || upto <= slowest + kPipeRecords) return true;
                if (ctl[3] != 0u || ++spins > kPipeSpinLimit) return false;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        for (uint32_t jb = 0; jb < steps; jb += 8) {
            PIPE_WAIT_BEGIN();
            const bool room_ok = wait_for_room(jb + 8u);
            PIPE_WAIT_END();
            PIPE_ROUND();
            if (!room_ok) { gave_up = true; break; }
            const unsigned long long d8 = (unsigned long long)ahead.x | ((unsigned long long)ahead.y << 32);
            if (jb + 8 < steps) ahead = *reinterpret_cast<const uint2*>(dec + (size_t)((jb >> 3) + 1u) * 512u);
            double pq[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                const uint32_t ctx = (uint32_t)(d8 >> (8u * q + 1u)) & 31u;
                pq[q] = probs[(ctx < L ? ctx : 0u) * 64u + lane];
            }
            uint32_t r[8];
            if (jb + 8u <= nd_all) {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) r[q] = encode_step(s, pq[q], ((uint32_t)(d8 >> (8u * q)) & 1u) != 0u);
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) {
                    r[q] = 0u;
                    if (jb + q < nd) r[q] = encode_step(s, pq[q], ((uint32_t)(d8 >> (8u * q)) & 1u) != 0u);
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; q++)
                if (jb + q < nd) recring[((jb + q) & (kPipeRecords - 1u)) * 64u + lane] = r[q];      // (a lane never writes beyond ITS records: the slot of its stop record stays free)
            EAE_PIPE_ORDER();
            if (lane == 0u) ctl[0] = jb + 8u;
        }
        if (!gave_up && !wait_for_room(steps + 1u)) gave_up = true;       // the stop records: slot nd of every lane
        if (!gave_up) {
            if (nd) recring[(nd & (kPipeRecords - 1u)) * 64u + lane] = stop_record(s);       // BinaryArithmeticCoder.cpp:61-102
            EAE_PIPE_ORDER();
            if (lane == 0u) ctl[1] = 1u;
        } else if (lane == 0u) {
            ctl[3] = 1u;
        }
        if (live && !good) p.status[m] = RETRY;          // the general kernel reproduces the exact code and stage
        PIPE_REPORT(0);
        return;
    }

    if (role == 1u) {
        // ---- bit writer: emit_kernel's stream, one map per lane
        uint32_t* out = reinterpret_cast<uint32_t*>(p.streams + (uint64_t)(in_range ? m : 0u) * p.stride);
        const uint32_t size_bits = round_up_to_byte(required_bits(p.map_size, L));
        uint32_t j = 0, cnt = 0, nwords = 0, pending = 0, total = 0;
        unsigned long long acc = 0ull;
        bool fin = nd == 0u, bad = false;
        uint32_t spins = 0;
        bool gave_up = false;
        auto flush = [&]() {
            const uint32_t word = (uint32_t)acc;
            wring[(nwords & (kRing - 1u)) * 64u + lane] = __builtin_bitreverse32(word);
            out[nwords] = word;
            acc >>= 32;
            cnt -= 32u;
            nwords++;
        };
        for (;;) {
            EAE_PIPE_ORDER();
            const uint32_t have = ctl[0], stopped = ctl[1], taken_words = rd_word[lane];
            // how many records this lane takes this round: what the encoder has left for it (its stop record once the encoder is
            // through), at most eight, and only as many as the decoder's ring has room for -- one word per record on the short path,
            // three spare for a long pending run, after which the lane rests until the next round
            const uint32_t ordinary = j < nd ? (have < nd ? have : nd) - j : 0u;
            const uint32_t ready = ordinary + ((j + ordinary == nd && stopped != 0u) ? 1u : 0u);
            const uint32_t slots = taken_words + kRing - nwords;
            uint32_t limit = ready < 8u ? ready : 8u;
            limit = limit + 3u <= slots ? limit : (slots > 3u ? slots - 3u : 0u);
            if (fin || bad) limit = 0u;
            const bool progressed = limit != 0u;
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                // integer masks, not booleans: the compiler turns the latter into SGPR-pair logic, an issue slot each for a lone wave
                const uint32_t gom = q < limit ? 0xFFFFFFFFu : 0u;                       // this lane takes a record in this step
                const uint32_t r = recring[(j & (kPipeRecords - 1u)) * 64u + lane];
                const uint32_t n = record_n(r), k = record_k(r), stop = (r >> 4) & 1u;
                const uint32_t queue = pending + stop;
                const uint32_t hasm = n != 0u ? gom : 0u;                                 // ... and the record shifts bits out
                const uint32_t first = r >> 31;
                // the other n - 1 leaving bits, first in time at bit 0 (what r << 1 drags in from the record's low half lands above them)
                const uint32_t rest = __builtin_amdgcn_ubfe(__builtin_bitreverse32(r << 1), 0u, n - 1u);
                // unusual, anywhere in the wave: a pending run beyond 15 bits (nearly dead maps under a very skewed first probability),
                // a stream about to outgrow its region (Bitstream.cpp:32-35)
                const uint32_t odd = (queue > 15u || total + n + queue > size_bits) ? hasm : 0u;
                if (__any(odd != 0u)) {
                    if (odd != 0u && (total + n + queue > size_bits || queue > 64u)) bad = true;       // the general kernel's
                    if (hasm != 0u && !bad) {
                        acc |= (unsigned long long)first << cnt;
                        cnt += 1u;
                        if (cnt >= 32u) flush();
                        for (uint32_t left_run = queue; left_run != 0u;) {
                            const uint32_t t = left_run < 16u ? left_run : 16u;
                            acc |= (unsigned long long)(first ? 0u : ((1u << t) - 1u)) << cnt;
                            cnt += t;
                            if (cnt >= 32u) flush();
                            left_run -= t;
                        }
                        acc |= (unsigned long long)rest << cnt;
                        cnt += n - 1u;
                        if (cnt >= 32u) flush();
                        total += n + queue;
                        if (queue > 15u) limit = q + 1u;          // it may have sent three words: the room was counted for one
                    }
                } else {
                    // the first leaving bit, `queue` copies of its complement, then the other n - 1: at most 31 bits, one append
                    const uint32_t run = (first - 1u) & ((1u << (queue & 31u)) - 1u);
                    const uint32_t v = (first | (run << 1) | (rest << ((queue + 1u) & 31u))) & hasm;
                    const uint32_t c = (n + queue) & hasm;
                    acc |= (unsigned long long)v << cnt;
                    cnt += c;
                    total += c;
                    if (cnt >= 32u) flush();
                }
                pending = gom != 0u ? (n != 0u ? k : pending + k) : pending;
                j -= gom;                                                                  // + 1 where the lane took the record
                if (__any((gom & stop) != 0u)) {
                    if ((gom & stop) != 0u && !bad) {
                        if (cnt) { cnt = 32u; flush(); }              // Bitstream flush of the partial word (zeros above the last bit)
                        fin = true;
                        p.bac_bits[m] = total;
                    }
                }
            }
            wr_bits[lane] = total;
            EAE_PIPE_ORDER();
            wr_words[lane] = nwords | (fin ? kPipeDone : 0u) | (bad ? kPipeBad : 0u);
            const uint32_t mine = (!fin && !bad && j < nd) ? j : 0xFFFFFFFFu;
            const uint32_t slowest = ~wave_max(~mine);
            if (lane == 0u) ctl[2] = slowest;
            if (!__any(!fin && !bad)) break;
            PIPE_ROUND();
            if (__any(progressed)) spins = 0;
            else {
                if (ctl[3] != 0u || ++spins > kPipeSpinLimit) { gave_up = true; break; }
                PIPE_WAIT_BEGIN();
                __builtin_amdgcn_s_sleep(2);
                PIPE_WAIT_END();
            }
        }
        PIPE_REPORT(1);
        if (gave_up && lane == 0u) ctl[3] = 1u;
        if (good && (bad || gave_up || ctl[3] != 0u)) p.status[m] = RETRY;
        return;
    }

    // ---- decoder core: bac_decode_core_kernel, its ring filled by the bit writer
    uint32_t* ring = wring + lane;
    const uint32_t size = good ? p.map_size : 0u;
    Interval s = interval_init();
    const double p0 = probs[lane];
    double pk = p0;
    uint32_t unary = 0, i = 0, taken = 0, code32 = 0, rcount = 0, rword = 0;
    unsigned long long rwin = 0ull;
    bool begun = false, dropped = size == 0u;
#ifdef EAE_PIPE_PROBE
    bool probe_seen_complete = false;
    unsigned probe_fast = 0, probe_lost = 0, probe_waits = 0;
#endif
    uint8_t* prefix = p.prefixes + (size_t)(in_range ? m : 0u) * p.map_size;
    uint32_t spins = 0;
    for (;;) {
        EAE_PIPE_ORDER();
        const uint32_t w = wr_words[lane];
        const bool complete = (w & kPipeDone) != 0u;
        if (w & kPipeBad) dropped = true;
        EAE_PIPE_ORDER();
        const uint32_t avail = complete ? wr_bits[lane] : (w & 0x3FFFFFFFu) * 32u;      // stream bits in the ring so far (all of them once complete)
        bool progressed = false;
#ifdef EAE_PIPE_PROBE
        if (blockIdx.x == 0 && !probe_seen_complete && !__any(!dropped && !complete)) {
            probe_seen_complete = true;
            if (lane == 0) { g_pipe_probe2[0] = probe_rounds; g_pipe_probe2[1] = clock64() - probe_t0; }
        }
#endif
        // A round costs the wavefront the same whether one lane steps or all of them: wait until every lane that still decodes has
        // the 46 bits a step needs and two words beyond (or its whole stream) -- lanes are fed at the pace of their own maps, and
        // stepping whenever any of them could ran 2.6 rounds per round of the encoder.
        // ... unless some lane's ring is half full: its writer lane must never be left without room while this wave waits for
        // another lane (the encoder stops within a ring of records of its slowest writer lane, and would starve that other lane).
        const bool fed = complete || avail - taken >= kStepMargin + 64u;
        const bool crowded = !dropped && i < size && avail - taken >= 16u * 32u;
        if (__any(!dropped && i < size && !fed) && !__any(crowded)) {
            if (ctl[3] != 0u || ++spins > kPipeSpinLimit) { if (lane == 0u) ctl[3] = 1u; break; }
            PIPE_WAIT_BEGIN();
            __builtin_amdgcn_s_sleep(4);
            PIPE_WAIT_END();
            continue;
        }
        spins = 0;
        if (!dropped && !begun && (complete || avail >= 64u)) {
            // Bac::start_decoding (BinaryArithmeticCoder.cpp:104-122): 16 bits, the last one repeated once the stream is exhausted
            rwin = ((unsigned long long)ring[0] << 32) | (unsigned long long)ring[64];
            rcount = 64u;
            rword = 2u;
            const uint32_t k = avail < 16u ? avail : 16u;
            uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - k));
            const uint32_t sticky = bits & 1u;
            bits = (bits << (16u - k)) | (sticky ? ((1u << (16u - k)) - 1u) : 0u);
            rwin <<= k;
            rcount -= k;
            taken = k;
            code32 = bits << 16;
            begun = true;
            progressed = true;
        }
        // A round in which every lane that still decodes has begun, has eight symbols left and -- its stream complete -- eight
        // steps' worth of bits, or -- its stream still growing -- the 46 bits a step needs, runs WITHOUT per-step checks, like the
        // fast rounds of bac_decode_core_kernel. For a growing stream that is a bet (a step takes 30 bits at most, 0.3-1.5 on
        // average): the lane's state is parked in registers first, the window only takes words the writer has sent, and if any
        // lane ends the round having taken more bits than its stream had, every lane goes back to the parked state and the
        // round is run again step by step with the checks.
        const uint32_t sent = complete ? 0xFFFFFFFFu : (w & 0x3FFFFFFFu);                 // words the window may take
        bool fast_round = !__any(!dropped && i < size && (!begun || i + 8u > size || (complete ? avail - taken < 8u * 30u : avail - taken < kStepMargin)));
        if (fast_round) {
            const Interval s0 = s;
            const uint32_t code0 = code32, rcount0 = rcount, rword0 = rword, taken0 = taken, unary0 = unary, i0 = i;
            const unsigned long long rwin0 = rwin;
            const double pk0 = pk;
            if (!dropped && i < size) {
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) {
                    const uint32_t wnext = ring[(rword & (kRing - 1u)) * 64u];
                    const double pspec = probs[(unary + 1u) * 64u + lane];
                    const bool need = rcount <= 32u && rword < sent;
                    rwin |= (unsigned long long)(need ? wnext : 0u) << (need ? 32u - rcount : 0u);
                    rcount += need ? 32u : 0u;
                    rword += need ? 1u : 0u;
                    const DecodeStep d = decode_step(s, code32, pk);
                    const uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - d.take));
                    rwin <<= d.take;
                    rcount -= d.take;
                    taken += d.take;
                    code32 = shift_code(code32, d, bits);
                    const uint32_t count = unary + (d.one ? 1u : 0u);
                    prefix[i] = (uint8_t)count;
                    const bool done = !d.one || count == L;
                    i += done ? 1u : 0u;
                    unary = done ? 0u : count;
                    pk = done ? p0 : pspec;
                }
            }
            if (__any(!dropped && taken > avail)) {            // the bet is lost: back to the parked state, and step by step below
                s = s0; code32 = code0; rcount = rcount0; rword = rword0; taken = taken0; unary = unary0; i = i0; rwin = rwin0; pk = pk0;
                fast_round = false;
#ifdef EAE_PIPE_PROBE
                probe_lost++;
#endif
            } else {
                progressed = true;
#ifdef EAE_PIPE_PROBE
                probe_fast++;
#endif
            }
        }
        if (!fast_round) {
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            if (!dropped && begun && i < size && (complete || avail - taken >= kStepMargin)) {
                const uint32_t wnext = ring[(rword & (kRing - 1u)) * 64u];
                const double pspec = probs[(unary + 1u) * 64u + lane];
                const bool need = rcount <= 32u;
                rwin |= (unsigned long long)(need ? wnext : 0u) << (need ? 32u - rcount : 0u);
                rcount += need ? 32u : 0u;
                rword += need ? 1u : 0u;
                const DecodeStep d = decode_step(s, code32, pk);
                const uint32_t left = avail - taken;
                const uint32_t k = d.take < left ? d.take : left;
                uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - k));
                const uint32_t ext = d.take - k;
                bits = (bits << ext) | ((bits & 1u) ? ((1u << ext) - 1u) : 0u);
                rwin <<= k;
                rcount -= k;
                taken += k;
                code32 = shift_code(code32, d, bits);
                const bool done = !d.one || unary + 1u == L;
                if (done) prefix[i] = (uint8_t)(d.one ? L : unary);
                i += done ? 1u : 0u;
                unary = done ? 0u : unary + 1u;
                pk = done ? p0 : pspec;
                progressed = true;
            }
        }
        }
        rd_word[lane] = rword >= 2u ? rword - 2u : 0u;      // the window holds words rword - 2 and rword - 1 at most: everything below has left the ring
        EAE_PIPE_ORDER();
        if (!__any(!dropped && i < size)) break;
        PIPE_ROUND();
        if (__any(progressed)) spins = 0;
        else {
            if (ctl[3] != 0u || ++spins > kPipeSpinLimit) { if (lane == 0u) ctl[3] = 1u; break; }
            PIPE_WAIT_BEGIN();
            __builtin_amdgcn_s_sleep(2);
            PIPE_WAIT_END();
        }
    }
    PIPE_REPORT(2);
#ifdef EAE_PIPE_PROBE
    if (blockIdx.x == 0 && lane == 0) { g_pipe_probe2[2] = probe_fast; g_pipe_probe2[3] = probe_lost; }
#endif
    if (!dropped && i < size) p.status[m] = RETRY;      // an aborted workgroup: what was not decoded here is the general kernel's
}

#ifdef EAE_PIPE_PROBE
extern "C" int eae_hip_debug_pipe_probe(unsigned long long* out12) {
    return (int)hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_pipe_probe), 12 * sizeof(unsigned long long));
}
extern "C" int eae_hip_debug_pipe_probe2(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pipe_probe2), 8 * sizeof(unsigned long long));
}
#endif

// which maps the pipeline handed to the general kernel: their decoder is the general kernel's too (the pipeline left no prefixes)
__global__ void pipe_note_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.n_maps) p.avail_bits[m] = p.status[m] == RETRY ? 1u : 0u;
}
__global__ void pipe_restore_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < p.n_maps && p.avail_bits[m] != 0u && p.status[m] == 0) p.status[m] = RETRY;
}

#endif  // EAE_EXPERIMENTAL_CODER

// status 0 -> RETRY for the coded maps: hands every map that has not failed to the general kernel
__global__ void mark_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= p.n_maps) return;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row >= 0 && p.status[m] == 0) p.status[m] = RETRY;
    if (p.ndec) p.ndec[m] = 1u;
}

uint32_t decision_capacity(uint32_t map_size, uint8_t L) { return ((map_size * ((uint32_t)L + 1u)) + 7u) & ~7u; }
bool fast_applies(uint8_t L) { return L >= 1 && L <= kMaxFastL; }

}  // namespace

extern "C" {

// workspace: [per-map decision counts][decisions of an encode | decoded symbols of a verify][records of an encode | prefix bytes of
// a decode], 256-byte aligned pieces
static uint64_t round256(uint64_t v) { return (v + 255u) & ~(uint64_t)255u; }
static uint64_t piece_a_bytes(uint32_t n_maps, uint32_t map_size, uint8_t L) {
    const uint64_t groups = ((uint64_t)n_maps + 63u) / 64u;
    const uint64_t decisions = fast_applies(L) ? groups * 64u * decision_capacity(map_size, L) : 0u;
    const uint64_t decoded = (uint64_t)n_maps * map_size * sizeof(int16_t);
    return round256(decisions > decoded ? decisions : decoded);
}
static uint64_t piece_b_bytes(uint32_t n_maps, uint32_t map_size, uint8_t L) {
    const uint64_t records = fast_applies(L) ? (uint64_t)n_maps * (decision_capacity(map_size, L) + kRecordPad) * sizeof(uint32_t) : 0u;
    const uint64_t prefixes = (uint64_t)n_maps * map_size;
    return round256(records > prefixes ? records : prefixes);
}
uint64_t eae_hip_coder_workspace_bytes(uint32_t n_maps, uint32_t map_size, uint8_t L) {
    return 256u + round256((uint64_t)n_maps * sizeof(uint32_t)) + piece_a_bytes(n_maps, map_size, L) + piece_b_bytes(n_maps, map_size, L);
}

static SimdParams make_params(uint32_t n_maps, uint32_t map_size, uint8_t L, const int16_t* symbols, const double* probs,
                              const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                              uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace) {
    SimdParams p{};
    p.n_maps = n_maps; p.map_size = map_size; p.L = L; p.dcap = decision_capacity(map_size, L); p.rcap = p.dcap + kRecordPad;
    p.symbols = symbols; p.probs = probs; p.prob_row = prob_row; p.streams = streams; p.stride = stride;
    p.bac_bits = bac_bits; p.bypass_bits = bypass_bits; p.status = status; p.stage = stage;
    if (workspace) {
        uint8_t* ws = reinterpret_cast<uint8_t*>(round256(reinterpret_cast<uintptr_t>(workspace)));
        p.ndec = reinterpret_cast<uint32_t*>(ws);
        p.decisions = ws + round256((uint64_t)n_maps * sizeof(uint32_t));
        p.decoded = reinterpret_cast<int16_t*>(p.decisions);
        p.records = reinterpret_cast<uint32_t*>(p.decisions + piece_a_bytes(n_maps, map_size, L));
        p.prefixes = reinterpret_cast<uint8_t*>(p.records);
    }
    return p;
}

static int check_simd_layout(uint32_t map_size, uint8_t L, const uint8_t* streams, uint64_t stride) {
    const uint64_t half = stride / 2;
    return (half < (uint64_t)(round_up_to_byte(required_bits(map_size, L)) >> 3) + 16 || (stride & 15u) ||
            (reinterpret_cast<uintptr_t>(streams) & 15u)) ? 1 : 0;
}

int eae_hip_coder_encode_batch(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                               const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                               uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace,
                               uint64_t workspace_bytes, void* stream) {
    if (!symbols || !probs || !streams || !bac_bits || !bypass_bits || !status || !workspace) return -1;
    if (check_simd_layout(map_size, L, streams, stride)) return 1;
    if (workspace_bytes < eae_hip_coder_workspace_bytes(n_maps, map_size, L)) return 1;
    if (n_maps == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (!fast_applies(L) || map_size == 0) {
        (void)hipMemsetAsync(status, 0, (size_t)n_maps * sizeof(int32_t), s);
        return eae_coder_generic_encode(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                        status, stage, 0, s);
    }
    const SimdParams p = make_params(n_maps, map_size, L, symbols, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                     status, stage, workspace);
    hipLaunchKernelGGL(binarise_kernel, dim3(n_maps), dim3(64), 0, s, p);
    hipLaunchKernelGGL(bac_encode_core_kernel<false>, dim3((n_maps + 63u) / 64u), dim3(64), (size_t)L * 64u * sizeof(double), s, p);
    hipLaunchKernelGGL(emit_kernel<false>, dim3(n_maps), dim3(64), 0, s, p);
    const int rc = eae_coder_generic_encode(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits,
                                            bypass_bits, status, stage, RETRY, s);
    return rc ? rc : (int)hipGetLastError();
}

int eae_hip_coder_decode_batch(uint32_t n_maps, uint32_t map_size, int16_t* symbols_out, const int16_t* expected, uint8_t L,
                               const double* probs, const int32_t* prob_row, const uint8_t* streams, uint64_t stride,
                               const uint32_t* bac_bits, const uint32_t* bypass_bits, int32_t* status, int32_t* stage,
                               void* workspace, uint64_t workspace_bytes, void* stream) {
    if (!probs || !streams || !bac_bits || !bypass_bits || !status) return -1;
    if (!symbols_out && (!expected || !workspace)) return -1;
    const bool have_ws = workspace && workspace_bytes >= eae_hip_coder_workspace_bytes(n_maps, map_size, L);
    if (!symbols_out && !have_ws) return 1;
    if (n_maps == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    SimdParams p = make_params(n_maps, map_size, L, expected, probs, prob_row, const_cast<uint8_t*>(streams), stride,
                               const_cast<uint32_t*>(bac_bits), const_cast<uint32_t*>(bypass_bits), status, stage,
                               have_ws ? workspace : nullptr);
    if (symbols_out) p.decoded = symbols_out;
    if (!expected) (void)hipMemsetAsync(status, 0, (size_t)n_maps * sizeof(int32_t), s);   // a pure decode starts from a clean slate
    // the 64-maps-per-wavefront kernels need the workspace (prefix bytes) and 16-byte aligned streams; otherwise, or for L == 0 or
    // L > 32, the general kernel decodes every map
    const bool fast = fast_applies(L) && map_size && have_ws && !check_simd_layout(map_size, L, streams, stride);
    if (fast) {
        hipLaunchKernelGGL(bac_decode_core_kernel<false>, dim3((n_maps + 63u) / 64u), dim3(64), decode_lds_bytes(L), s, p);
        // (small steps: the staged form, whose tile loop waits for no memory; see the kernel)
        if (n_maps <= 256u) hipLaunchKernelGGL(debinarise_kernel<true>, dim3(n_maps), dim3(64), 0, s, p);
        else hipLaunchKernelGGL(debinarise_kernel<false>, dim3(n_maps), dim3(64), 0, s, p);
    } else {
        hipLaunchKernelGGL(mark_kernel, dim3((n_maps + 255u) / 256u), dim3(256), 0, s, p);
    }
    // whatever a map reported that only the general kernel can name (errors of any kind), or every map
    const int rc = eae_coder_generic_decode(n_maps, map_size, p.decoded, L, probs, prob_row, streams, stride, bac_bits,
                                            bypass_bits, status, stage, RETRY, s);
    if (rc) return rc;
    if (expected) {
        if (p.ndec) hipLaunchKernelGGL(compare_kernel, dim3(n_maps), dim3(64), 0, s, p);
        else return 1;
    }
    return (int)hipGetLastError();
}

#if defined(EAE_EXPERIMENTAL_CODER) && !defined(EAE_DECODE_TOPUP_ZEROS)
// ---- the chunked round trip: decoder and emit pass trailing the encoder core ------------------------------------------------
// For a batch of one or two images the coder is two to four wavefronts and its time is the LENGTH of its serial chains: binarise
// -> encoder core -> emit -> decoder core -> debinarise, one after the other (0.72 ms of the 1.2 ms one Kodak image takes, of
// which the two cores are 0.29 + 0.33). Cut into chunks, the chains overlap: while the encoder core runs chunk c + 1, the emit
// pass turns the records of chunk c into stream words and the decoder works through the words of chunk c - 1. Three streams,
// nothing polled: launch order and events carry the dependencies, so the schedule can be captured into a hipGraph like any other.
uint64_t eae_hip_coder_trailing_workspace_bytes(uint32_t n_maps, uint32_t map_size, uint8_t L) {
    const uint64_t groups = ((uint64_t)n_maps + 63u) / 64u;
    return eae_hip_coder_workspace_bytes(n_maps, map_size, L) + round256((uint64_t)n_maps * map_size) + round256(groups * 4u) +
           round256((uint64_t)n_maps * 8u) + round256((uint64_t)n_maps * 16u) + round256((uint64_t)n_maps * 4u) +
           round256((uint64_t)n_maps * 16u) + round256((uint64_t)n_maps * 8u) + 256u;
}

namespace {
struct TrailingStreams {
    hipStream_t emit = nullptr, decode = nullptr;
    hipEvent_t encoded[16] = {}, emitted[16] = {}, start = nullptr, done = nullptr;
    bool complete = false;
};
// a few sets, handed out in turn: calls that overlap in time (one per batch in flight) then do not share their side streams.
// Hand-out and first-use creation are under one mutex (callers on different threads, e.g. one per codec); a set whose creation
// failed half way is never handed out (`complete`).
TrailingStreams* trailing_streams() {
    static TrailingStreams sets[16][4];
    static unsigned next[16] = {0};
    static std::mutex guard;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> hold(guard);
    TrailingStreams& t = sets[dev][next[dev]++ & 3u];
    if (!t.complete) {
        if (!t.emit && hipStreamCreateWithFlags(&t.emit, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (!t.decode && hipStreamCreateWithFlags(&t.decode, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < 16; i++) {
            if (!t.encoded[i] && hipEventCreateWithFlags(&t.encoded[i], hipEventDisableTiming) != hipSuccess) return nullptr;
            if (!t.emitted[i] && hipEventCreateWithFlags(&t.emitted[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (!t.start && hipEventCreateWithFlags(&t.start, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) return nullptr;
        t.complete = true;
    }
    return &t;
}
// Whatever was put on the side streams, the caller's stream waits for it: also on the error returns, so that a caller who frees
// or reuses the buffers behind a failed call does not race the launches that did go out.
int trailing_join(TrailingStreams* t, hipStream_t s, int rc) {
    (void)hipEventRecord(t->emitted[0], t->emit);
    (void)hipStreamWaitEvent(t->decode, t->emitted[0], 0);
    (void)hipEventRecord(t->done, t->decode);
    (void)hipStreamWaitEvent(s, t->done, 0);
    return rc;
}
}  // namespace

int eae_hip_coder_roundtrip_trailing(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                                     const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                                     uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace,
                                     uint64_t workspace_bytes, uint32_t chunks, void* stream) {
    if (!symbols || !probs || !streams || !bac_bits || !bypass_bits || !status || !workspace) return -1;
    if (check_simd_layout(map_size, L, streams, stride)) return 1;
    if (workspace_bytes < eae_hip_coder_trailing_workspace_bytes(n_maps, map_size, L)) return 1;
    if (n_maps == 0) return 0;
    if (chunks > 16u) chunks = 16u;
    if (chunks < 2u || !fast_applies(L) || map_size == 0) {        // nothing to overlap: the two calls one after the other
        const int rc = eae_hip_coder_encode_batch(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                                  status, stage, workspace, workspace_bytes, stream);
        if (rc) return rc;
        return eae_hip_coder_decode_batch(n_maps, map_size, nullptr, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                          status, stage, workspace, workspace_bytes, stream);
    }
    TrailingStreams* t = trailing_streams();
    if (!t) return (int)hipErrorUnknown;
    hipStream_t s = (hipStream_t)stream;
    SimdParams p = make_params(n_maps, map_size, L, symbols, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage, workspace);
    {   // the pieces only the chunked form needs, behind the standard ones: prefix bytes of their own (the records are still being
        // read while the decoder writes them), the group chain lengths, and what crosses the launches
        uint8_t* at = reinterpret_cast<uint8_t*>(p.records) + piece_b_bytes(n_maps, map_size, L);
        const uint64_t groups = ((uint64_t)n_maps + 63u) / 64u;
        p.prefixes = at; at += round256((uint64_t)n_maps * map_size);
        p.group_steps = reinterpret_cast<uint32_t*>(at); at += round256(groups * 4u);
        p.enc_state = reinterpret_cast<uint2*>(at); at += round256((uint64_t)n_maps * 8u);
        p.emit_state = reinterpret_cast<uint4*>(at); at += round256((uint64_t)n_maps * 16u);
        p.avail_bits = reinterpret_cast<uint32_t*>(at); at += round256((uint64_t)n_maps * 4u);
        p.dec_state = reinterpret_cast<uint4*>(at); at += round256((uint64_t)n_maps * 16u);
        p.dec_state2 = reinterpret_cast<uint2*>(at);
    }
    const dim3 per_map(n_maps), per_group((n_maps + 63u) / 64u), wave(64);
    // every lane starts "not started" (0xFF..): a decoder chunk that finds too few bits leaves it so
    (void)hipMemsetAsync(p.dec_state, 0xFF, (size_t)n_maps * 16u, s);
    (void)hipMemsetAsync(p.avail_bits, 0, (size_t)n_maps * 4u, s);
    hipLaunchKernelGGL(binarise_kernel, per_map, wave, 0, s, p);
    (void)hipEventRecord(t->start, s);
    (void)hipStreamWaitEvent(t->emit, t->start, 0);
    (void)hipStreamWaitEvent(t->decode, t->start, 0);
    p.nchunks = chunks;
    for (uint32_t c = 0; c < chunks; c++) {
        p.chunk = c;
        hipLaunchKernelGGL(bac_encode_core_kernel<true>, per_group, wave, (size_t)L * 64u * sizeof(double), s, p);
        (void)hipEventRecord(t->encoded[c], s);
        (void)hipStreamWaitEvent(t->emit, t->encoded[c], 0);
        hipLaunchKernelGGL(emit_kernel<true>, per_map, wave, 0, t->emit, p);
        (void)hipEventRecord(t->emitted[c], t->emit);
        if (c >= 1u && c + 1u < chunks) {
            // decoder chunk c - 1, on what the emit pass of chunk c has put in memory (the last decoder chunk follows below)
            SimdParams d = p;
            d.chunk = c - 1u;
            d.nchunks = chunks - 1u;
            (void)hipStreamWaitEvent(t->decode, t->emitted[c], 0);
            hipLaunchKernelGGL(bac_decode_core_kernel<true>, per_group, wave, decode_lds_bytes(L), t->decode, d);
        }
    }
    // whatever the fast encoder could not finish, recoded by the general kernel (same bytes), then the decoder's last chunk on the
    // complete streams, the data-parallel rest of the decoder and the comparison: the tail of eae_hip_coder_decode_batch
    int rc = eae_coder_generic_encode(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage,
                                      RETRY, t->emit);
    if (rc) return trailing_join(t, s, rc);
    (void)hipEventRecord(t->emitted[chunks - 1u], t->emit);
    (void)hipStreamWaitEvent(t->decode, t->emitted[chunks - 1u], 0);
    {
        SimdParams d = p;
        d.chunk = chunks - 2u;
        d.nchunks = chunks - 1u;
        hipLaunchKernelGGL(bac_decode_core_kernel<true>, per_group, wave, decode_lds_bytes(L), t->decode, d);
        hipLaunchKernelGGL(debinarise_kernel<false>, per_map, wave, 0, t->decode, d);
        rc = eae_coder_generic_decode(n_maps, map_size, d.decoded, L, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage,
                                      RETRY, t->decode);
        if (rc) return trailing_join(t, s, rc);
        hipLaunchKernelGGL(compare_kernel, per_map, wave, 0, t->decode, d);
    }
    (void)hipEventRecord(t->done, t->decode);
    (void)hipStreamWaitEvent(s, t->done, 0);
    return (int)hipGetLastError();
}
// ---- the round trip with the three serial stages of a group in ONE workgroup (coder_pipe_kernel) -----------------------------
int eae_hip_coder_roundtrip_fused(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                                  const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                                  uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace,
                                  uint64_t workspace_bytes, void* stream) {
    if (!symbols || !probs || !streams || !bac_bits || !bypass_bits || !status || !workspace) return -1;
    if (check_simd_layout(map_size, L, streams, stride)) return 1;
    if (workspace_bytes < eae_hip_coder_trailing_workspace_bytes(n_maps, map_size, L)) return 1;
    if (n_maps == 0) return 0;
    if (!fast_applies(L) || map_size == 0) {
        const int rc = eae_hip_coder_encode_batch(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                                  status, stage, workspace, workspace_bytes, stream);
        if (rc) return rc;
        return eae_hip_coder_decode_batch(n_maps, map_size, nullptr, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits,
                                          status, stage, workspace, workspace_bytes, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    SimdParams p = make_params(n_maps, map_size, L, symbols, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage, workspace);
    p.avail_bits = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(p.records) + piece_b_bytes(n_maps, map_size, L));      // one flag per map
    const dim3 per_map(n_maps), per_group((n_maps + 63u) / 64u), wave(64), flat((n_maps + 255u) / 256u);
    hipLaunchKernelGGL(binarise_kernel, per_map, wave, 0, s, p);
    hipLaunchKernelGGL(coder_pipe_kernel, per_group, dim3(256), pipe_lds_bytes(L), s, p);
    hipLaunchKernelGGL(pipe_note_kernel, flat, dim3(256), 0, s, p);
    int rc = eae_coder_generic_encode(n_maps, map_size, symbols, L, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage, RETRY, s);
    if (rc) return rc;
    hipLaunchKernelGGL(pipe_restore_kernel, flat, dim3(256), 0, s, p);
    hipLaunchKernelGGL(debinarise_kernel<false>, per_map, wave, 0, s, p);
    rc = eae_coder_generic_decode(n_maps, map_size, p.decoded, L, probs, prob_row, streams, stride, bac_bits, bypass_bits, status, stage, RETRY, s);
    if (rc) return rc;
    hipLaunchKernelGGL(compare_kernel, per_map, wave, 0, s, p);
    return (int)hipGetLastError();
}
#endif

}  // extern "C"
