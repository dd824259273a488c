// coder_simd.hip -- the lossless coder reorganised for a 64-wide machine: 64 maps per wavefront, every lane in step.
//
// A binary arithmetic coder is a serial chain per stream (lossless/c++/source/BinaryArithmeticCoder.cpp:144-252), so the
// only parallelism is across the feature maps of a batch. The per-lane kernels of coder_device.hip run the reference's
// control flow in every lane; lanes diverge at every symbol, and the many long-lived waves cost the transform kernels
// that run next to them about a tenth of their rate. Here the work is split so that the serial part is the same
// instruction stream for all 64 lanes:
//
//   encode   (1) binarise_kernel, one wavefront per map, fully parallel over the symbols: UEG0 binarisation
//                (LosslessCoder.cpp:232-252) -> a list of (bit, context) decisions, one byte each, and the complete
//                bypass stream (signs, Exp-Golomb suffixes: bit offsets by wave prefix sums).
//            (2) bac_encode_kernel, 64 maps per wavefront: step j feeds decision j of every lane's map to the interval
//                update; the renormalisation is closed-form and the pending-bit queue is emitted in one 64-bit put, so
//                the step is branch-light and the lanes stay converged.
//   decode   (3) bac_decode_kernel, 64 maps per wavefront: both streams of the 64 maps are staged in LDS; each step decodes
//                one decision per lane and advances a three-register binarisation state (unary count, symbol index).
//            (4) compare_kernel: decoded == encoded symbols (the assert of lossless/compression.py:146-153).
//
// A stream longer than the decoder's LDS windows (64 / 16 words; 192 / 48 for maps of more than 4096 symbols) is decoded by
// a second launch of the same kernel that reads the words beyond the window from memory. Anything else the fast kernels do
// not handle -- an error of any kind (their exact code and stage matter), L == 0 or L > 32 -- marks the map RETRY, and the general
// per-lane kernel (the shared core of coder_core.h, statement for statement the reference) recodes that map from
// scratch. Results are therefore identical to the host library's in every case; tests/test_coder_device.py compares
// bytes, bit counts, symbols, statuses and stages.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../coder/coder_core.h"
#include "eae_hip.h"

// coder_device.hip
int eae_coder_generic_encode(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                             const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                             uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream);
int eae_coder_generic_decode(uint32_t n_maps, uint32_t map_size, int16_t* out, uint8_t L, const double* probs,
                             const int32_t* prob_row, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                             const uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream);

namespace {

using namespace eae_core;

#ifndef EAE_SIMD_PRIO
#define EAE_SIMD_PRIO 3
#endif

constexpr int32_t RETRY = -100;           // internal: recode this map with the general kernel
constexpr uint32_t kMaxFastL = 32;        // contexts staged in LDS: 32 x 64 lanes x 8 B = 16 KB
// LDS windows per lane of the decoder, in dwords (arithmetic-coded stream, bypass stream). First pass: 2048 + 512 bits,
// 20 KB per block, small enough to sit next to the transform kernels' blocks. Second pass, only for the maps the first
// one found too long: 14336 + 3072 bits, 136 KB per block (one block per CU).
constexpr uint32_t kBacWindowWords = 64, kBypassWindowWords = 16;
// maps of more than kMediumMapSize symbols (latents of images beyond about 1 Mpixel) start with wider windows: their streams
// would mostly overflow the small ones and the second pass costs a full serial decode of its own
constexpr uint32_t kBacWindowWordsMedium = 192, kBypassWindowWordsMedium = 48, kMediumMapSize = 4096;
// dynamic LDS of bac_decode_kernel<WB, WY>: probabilities [L + 1][64] doubles, windows [WB + 3][64] and [WY + 1][64] words
constexpr size_t decode_lds_bytes(uint32_t L, uint32_t wb, uint32_t wy) {
    return ((size_t)L + 1u) * 64u * sizeof(double) + ((size_t)wb + 3u + wy + 1u) * 64u * sizeof(uint32_t);
}

struct SimdParams {
    uint32_t n_maps, map_size, L, dcap;   // dcap: bytes of decision storage per map (multiple of 8)
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
    uint32_t* ndec;                       // [n_maps]
    int32_t* perm;                        // decode, sorted form: [0..3] header (short blocks, long blocks), then 64 maps per block
};

// Exclusive prefix sum over the 64 lanes, and the total. Data-parallel primitives, not cross-lane loads: four row_shr steps scan
// the rows of 16 lanes, row_bcast:15 / :31 carry the row totals on (the sequence LLVM's atomic optimiser emits for gfx9): 6 DPP
// additions where six __shfl_up rounds were 6 x (ds_bpermute + compare + add) -- this runs once per 64 symbols of every map,
// next to the transforms.
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
// (2) 64 maps per wavefront: decision j of every map through the interval update, in step
// ---------------------------------------------------------------------------------------------------------------------
extern __shared__ double lds_dyn[];

__global__ __launch_bounds__(64) void bac_encode_kernel(const SimdParams p) {
    __builtin_amdgcn_s_setprio(EAE_SIMD_PRIO);
    const uint32_t lane = threadIdx.x;
    const uint32_t m = blockIdx.x * 64u + lane;
    const bool in_range = m < p.n_maps;
    const int32_t row = in_range ? (p.prob_row ? p.prob_row[m] : (int32_t)m) : -1;
    bool live = in_range && row >= 0 && p.status[m] == 0;
    const uint32_t L = p.L;
    double* probs = lds_dyn;                            // [context][lane]
    bool retry = false;
    if (live)
        for (uint32_t k = 0; k < L; k++) {
            const double pk = p.probs[(size_t)row * L + k];
            probs[k * 64u + lane] = pk;
            // An invalid probability only matters if its context is coded (BinaryArithmeticCoder.cpp:146-153): the general
            // kernel sorts that out; here the whole map is handed over so that the steps below need no check.
            if (!(pk > 0. && pk < 1.)) retry = true;
        }
    uint32_t nd = live && !retry ? p.ndec[m] : 0u;     // set to 0 when the map leaves for the general kernel: its steps stop
    const uint32_t steps = wave_max(nd);
    uint32_t err = 0;                                  // a condition only the general kernel reports (kept out of the branches)
    Bac bac;
    bac.init();
    bac.bs.init_writer(p.streams + (uint64_t)(in_range ? m : 0u) * p.stride, required_bits(p.map_size, L));
    const uint8_t* dec = p.decisions + (size_t)blockIdx.x * 64u * p.dcap + (size_t)lane * 8u;
    uint32_t low = 0, high = kRangeMax, e3 = 0;
    uint2 ahead = steps ? *reinterpret_cast<const uint2*>(dec) : make_uint2(0, 0);
    for (uint32_t jb = 0; jb < steps; jb += 8) {
        const unsigned long long d8 = (unsigned long long)ahead.x | ((unsigned long long)ahead.y << 32);
        if (jb + 8 < steps) ahead = *reinterpret_cast<const uint2*>(dec + (size_t)((jb >> 3) + 1u) * 512u);
        // eight steps emit at most the pending E3 bits + 8 x (16 leaving bits + 15 new E3 bits): near the end of the stream's
        // capacity (Bitstream.cpp:32-35) the map goes to the general kernel, which reproduces the exact point of failure
        if (bac.bs.write_index + e3 + 8u * 31u > bac.bs.size_bits) { err = 1u; nd = 0u; }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t j = jb + q;
            if (j < nd) {
                const uint32_t d = (uint32_t)(d8 >> (8u * q)) & 0xFFu;
                const uint32_t bit = d & 1u;
                const double pk = probs[(d >> 1) * 64u + lane];
                // Bac::update_middle + encode_bit (BinaryArithmeticCoder.cpp:144-180)
                const uint32_t mid = low + (uint32_t)(pk * (double)(high - low));
                uint32_t nl = bit ? mid + 1u : low;
                uint32_t nh = bit ? high : mid;
                err |= nl > kRangeMax ? 1u : 0u;                     // precision_error (cannot happen with 0 < p < 1)
                // E1/E2 in closed form (as Bac::encode): n leading equal bits leave, with the pending E3 bits behind the first
                const uint32_t diff = (nl ^ nh) & 0xFFFFu;
                const uint32_t n = diff ? (uint32_t)__builtin_clz(diff) - 16u : 16u;
                if (n) {
                    const uint32_t out = rev16(nh);
                    const unsigned long long first = out & 1u;
                    Bitstream& bs = bac.bs;
                    auto put = [&](unsigned long long bits, uint32_t cnt) {     // cnt <= 63 bits, first in time at bit 0
                        const uint32_t sh = bs.write_index & 63u;
                        bs.acc |= bits << sh;
                        if (sh + cnt >= 64u) {
                            store64(bs.data + ((bs.write_index >> 6) << 3), bs.acc);
                            bs.acc = sh ? bits >> (64u - sh) : 0ull;
                        }
                        bs.write_index += cnt;
                    };
                    if (e3 > 47u) {
                        // a long run of pending E3 bits (maps that are almost all zeros): the leaving bit, then the run in
                        // pieces, so that the single put below again holds at most 1 + 47 + 15 bits
                        put(first, 1u);
                        while (e3 > 32u) { put(first ? 0ull : 0xFFFFFFFFull, 32u); e3 -= 32u; }
                        const unsigned long long run = first ? 0ull : ((1ull << e3) - 1ull);
                        const unsigned long long rest = (out >> 1) & ((1u << (n - 1u)) - 1u);
                        put(run | (rest << e3), n - 1u + e3);
                    } else {
                        // the leaving bit followed by e3 complements, first in time at bit 0: 1, or 0 then e3 ones = 2^(e3+1) - 2
                        const unsigned long long head = first ? 1ull : ((2ull << e3) - 2ull);
                        const unsigned long long rest = (out >> 1) & ((1u << (n - 1u)) - 1u);
                        put(head | (rest << (1u + e3)), n + e3);
                    }
                    e3 = 0;
                    nl = (nl << n) & 0xFFFFu;
                    nh = ((nh << n) & 0xFFFFu) | ((1u << n) - 1u);
                }
                // E3 (BinaryArithmeticCoder.cpp:238-245) in closed form, see bac_decode_kernel: the number of scalings is the run
                // of positions below the top bit where `low` has a 1 and `high` a 0, capped where `high` would reach 0xBFFE
                {
                    const uint32_t e3_run = (uint32_t)__builtin_clz(~(((nl & ~nh) & 0x7FFFu) << 17));
                    const uint32_t e3_cap = 14u - (uint32_t)__builtin_ctz(~nh);
                    const uint32_t k3 = nh > kRangeThreeQuarters || nl <= kRangeQuarter ? 0u : (e3_run < e3_cap ? e3_run : e3_cap);
                    nl = (((nl - 0x8000u) << k3) + 0x8000u) & kRangeMax;
                    nh = (((nh - 0x8000u) << k3) + 0x8000u + ((1u << k3) - 1u)) & kRangeMax;
                    e3 += k3;
                }
                low = nl;
                high = nh;
            }
        }
    }
    if (err) retry = true;
    if (live) {
        int s = OK;
        if (!retry) {
            bac.low = low;
            bac.high = high;
            bac.nb_e3 = e3;
            s = bac.stop_encoding();                    // BinaryArithmeticCoder.cpp:61-102, flushes the stream
        }
        if (retry || s) p.status[m] = RETRY;            // the general kernel reproduces the exact code and stage
        else p.bac_bits[m] = bac.bs.write_index;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// (3) 64 maps per wavefront: decode, streams staged in LDS
// ---------------------------------------------------------------------------------------------------------------------
// MODE 0 / 1: first / second pass over the maps in their own order (maps 64 b .. 64 b + 63 in block b). MODE 2: ONE launch
// over the maps sorted by sort_maps_kernel into blocks of short maps (streams fit the windows) and blocks of long ones: a block
// is either all first-pass or all second-pass work, so no map is walked twice and no wavefront runs both loops. At the rates
// of trained models (~1 bit per pixel) most maps are long and nearly every group of 64 held both kinds: the two passes then
// cost two full serial decodes (1.94 ms per Kodak batch at 0.9 bpp against 0.57 ms at 0.19 bpp).
template <uint32_t WB, uint32_t WY, int MODE, bool SECOND>
__device__ __forceinline__ void bac_decode_body(const SimdParams& p, const uint32_t m) {
    const uint32_t lane = threadIdx.x;
    const bool in_range = m < p.n_maps;
    const int32_t row = in_range ? (p.prob_row ? p.prob_row[m] : (int32_t)m) : -1;
    const uint32_t L = p.L;
    // LDS: probabilities [L + 1][lane], arithmetic-coded window [WB + 3][lane], bypass window [WY + 1][lane]. The extra rows
    // let the step below read one context / up to three words / one word beyond the valid ones without a guard (see
    // decode_lds_bytes): a zeroed word is what the reference reads beyond the end of a stream anyway.
    double* probs = lds_dyn;
    uint32_t* wbac = reinterpret_cast<uint32_t*>(lds_dyn + ((size_t)L + 1u) * 64u);
    uint32_t* wbyp = wbac + (WB + 3u) * 64u;
    // First pass (SECOND == false): the streams of the 64 maps must fit the LDS windows of WB / WY words; a map whose streams
    // do not is marked RETRY. Second pass (same windows, so the same modest LDS request: a launch that asks for most of a
    // CU's LDS waits for a CU to drain, 0.2 ms next to the transforms even when it has nothing to do): only the RETRY maps,
    // and a lane whose stream is longer than the window fetches the words beyond it from memory itself, one load per word
    // (the wave waits for that load: slower, but exact and unbounded).
    bool live = in_range && row >= 0 && p.status[m] == (MODE == 1 ? RETRY : 0);
    if (MODE == 1 && !__any(live)) return;                  // nothing was handed on to this pass in this group of 64 maps
    const uint32_t nbac = live ? p.bac_bits[m] : 0u;
    const uint32_t nbyp = live ? p.bypass_bits[m] : 0u;
    const uint32_t* gbac = reinterpret_cast<const uint32_t*>(p.streams + (uint64_t)(in_range ? m : 0u) * p.stride);
    const uint32_t* gbyp = reinterpret_cast<const uint32_t*>(p.streams + (uint64_t)(in_range ? m : 0u) * p.stride + p.stride / 2);
    bool retry = false;
    if (live && (nbac > p.stride * 4u || nbyp > p.stride * 4u)) retry = true;    // beyond the buffer: not a stream of ours
    if (!SECOND && live && (nbac > WB * 32u || nbyp > WY * 32u)) retry = true;   // longer than the window: second pass
    if (live) {
        for (uint32_t k = 0; k < L; k++) {
            const double pk = p.probs[(size_t)row * L + k];
            probs[k * 64u + lane] = pk;
            if (!(pk > 0. && pk < 1.)) retry = true;     // only an error if that context is decoded: general kernel
        }
        probs[L * 64u + lane] = 0.5;                     // read ahead of an escape, never used
    }
    // stage the streams of the 64 maps: the wave copies one map per iteration, coalesced
    for (uint32_t l = 0; l < 64u; l++) {
        const uint32_t ml = __shfl(m, l, 64);
        if (ml >= p.n_maps) { if (MODE == 2) continue; else break; }
        const uint32_t bits_b = __shfl(nbac, l, 64), bits_y = __shfl(nbyp, l, 64);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(p.streams + (uint64_t)ml * p.stride);
        for (uint32_t w = lane; w < WB && w * 32u < bits_b; w += 64u) wbac[w * 64u + l] = src[w];
        const uint32_t* srcy = reinterpret_cast<const uint32_t*>(p.streams + (uint64_t)ml * p.stride + p.stride / 2);
        for (uint32_t w = lane; w < WY && w * 32u < bits_y; w += 64u) wbyp[w * 64u + l] = srcy[w];
    }
    {   // zeros behind the last word of this lane's stream: at most three window words are loaded beyond it (a refill happens
        // at 32 buffered bits or fewer and only stream bits are ever consumed: refills <= bits / 32 + 2)
        const uint32_t nwords = retry ? 0u : (nbac + 31u) >> 5;
        if (nwords <= WB)
            for (uint32_t t = 0; t < 3u; t++) wbac[(nwords + t) * 64u + lane] = 0u;
    }
    __syncthreads();
    int16_t* out = p.decoded + (size_t)(in_range ? m : 0u) * p.map_size;
    uint32_t low = 0, high = kRangeMax, code = 0, ridx = 0, yidx = 0;
    // rwin: the next rcount bits of the arithmetic-coded stream, LEFT-aligned with the next bit in time at bit 63, so that
    // "the next k bits, first in time most significant" (what the 16-bit code register wants) is one shift, not a bit reversal
    unsigned long long rwin = 0;
    uint32_t rcount = 0, rword = 0;       // rword: next dword of the window to load
    auto refill = [&]() {                 // 32 more bits once at most 32 are left, zeros beyond the stream
        const bool need = rcount <= 32u;
        const bool have = rword < WB && rword * 32u < nbac;                  // priming reads word 0 only
        const uint32_t w = wbac[(have ? rword : 0u) * 64u + lane];
        rwin |= (unsigned long long)(need && have ? __builtin_bitreverse32(w) : 0u) << (need ? 32u - rcount : 0u);
        rcount += need ? 32u : 0u;
        rword += need ? 1u : 0u;
    };
    auto take = [&](uint32_t k) {         // k <= 16 bits off the window (k == 0: nothing, returns 0)
        const uint32_t bits = (uint32_t)((rwin >> 1) >> (63u - k));
        rwin <<= k;
        rcount -= k;
        return bits;
    };
    bool active = live && !retry;
    if (active) {
        // Bac::start_decoding (BinaryArithmeticCoder.cpp:104-122): 16 bits, the last one repeated once the stream is exhausted
        refill();
        const uint32_t k = nbac < 16u ? nbac : 16u;
        uint32_t bits = take(k);
        const uint32_t sticky = bits & 1u;
        bits = (bits << (16u - k)) | (sticky ? ((1u << (16u - k)) - 1u) : 0u);
        code = bits;
        ridx = k;
    }
    uint32_t unary = 0, i = 0;
    const uint32_t size = p.map_size;
    if (size == 0) active = false;
    // One decision per lane per iteration. The body is written without lane-divergent branches except for the rare events
    // (E3 scalings, Exp-Golomb escapes): a lone wavefront issues an instruction every 6-7 cycles, so the length of this loop
    // IS the decoder's speed, and every divergent `if` costs a handful of scalar mask instructions on top of its body.
    // The three LDS reads of a step are issued one step ahead of their use (the window word of the next refill, the
    // probability of the context the next decision will be in if this one is a one, the bypass word holding the next sign
    // bit): none of their latencies sits in the decision -> decision dependency chain.
    const double p0 = active ? probs[lane] : 0.5;
    double pk = p0;
    uint32_t gw = 0, gw_ahead = 0, gw_row = 0xFFFFFFF0u, gy = 0, gy_row = 0xFFFFFFFFu;   // the words held from beyond the windows
    // (macros, not lambdas: with the held words captured by reference the compiler kept them in scratch memory and turned
    // the LDS read into a flat load selecting between LDS and scratch, waited for in every step)
#define EAE_WINDOW_WORD(dst_)                                                                                         \
    {                                                                                                                 \
        dst_ = wbac[(!SECOND || rword < WB + 2u ? rword : WB + 2u) * 64u + lane];                                     \
        if (SECOND && rword >= WB) {                                                                                  \
            /* beyond the window: the word comes from memory. The word after it is requested at the same time and has */  \
            /* the ~30 steps it takes to use up 32 bits to arrive (a load per word on demand stalled the wave for a   */  \
            /* memory round trip every time any of its 64 lanes crossed a word boundary)                              */  \
            if (rword != gw_row) {                                                                                    \
                gw = rword == gw_row + 1u ? gw_ahead : (rword * 32u < nbac ? gbac[rword] : 0u);                       \
                gw_row = rword;                                                                                       \
                gw_ahead = (rword + 1u) * 32u < nbac ? gbac[rword + 1u] : 0u;                                         \
            }                                                                                                         \
            dst_ = gw;                                                                                                \
        }                                                                                                             \
    }
#define EAE_BYPASS_WORD(dst_, yrow_)                                                                                  \
    {                                                                                                                 \
        const uint32_t yr_ = (yrow_);                                                                                 \
        dst_ = wbyp[(yr_ < WY ? yr_ : WY) * 64u + lane];                                                              \
        if (SECOND && yr_ >= WY) {                                                                                    \
            if (yr_ != gy_row) { gy = yr_ * 32u < nbyp ? gbyp[yr_] : 0u; gy_row = yr_; }                              \
            dst_ = gy;                                                                                                \
        }                                                                                                             \
    }
    uint32_t wnext = 0u;
    if (active) EAE_WINDOW_WORD(wnext)
    uint32_t err = 0;
    while (active) {
        {   // refill: 32 more bits once at most 32 are left, zeros beyond the stream
            const bool need = rcount <= 32u;
            rwin |= (unsigned long long)(need ? __builtin_bitreverse32(wnext) : 0u) << (need ? 32u - rcount : 0u);
            rcount += need ? 32u : 0u;
            rword += need ? 1u : 0u;
        }
        uint32_t wload;
        EAE_WINDOW_WORD(wload)
        const double pspec = probs[(unary + 1u) * 64u + lane];
        uint32_t yword;
        EAE_BYPASS_WORD(yword, yidx >> 5)
        // Bac::decode (BinaryArithmeticCoder.cpp:124-134, 254-320) with the closed-form renormalisation of coder_core.h
        const uint32_t mid = low + (uint32_t)(pk * (double)(high - low));
        const uint32_t bit = code > mid ? 1u : 0u;
        uint32_t nl = bit ? mid + 1u : low;
        uint32_t nh = bit ? high : mid;
        const uint32_t diff = (nl ^ nh) & 0xFFFFu;
        const uint32_t n = diff ? (uint32_t)__builtin_clz(diff) - 16u : 16u;
        nl = (nl << n) & kRangeMax;
        nh = ((nh << n) & kRangeMax) | ((1u << n) - 1u);
        // E3 in closed form too. After E1/E2 the intervals' top bits are 0 / 1; one E3 scaling (BinaryArithmeticCoder.cpp:
        // 238-245, 300-318) deletes the bit below them when it is 1 in `low` and 0 in `high`, so the loop runs once per
        // leading position where that holds (e3_run) -- except that the reference compares `high` with 3 * 0x3FFF = 0xBFFD,
        // not 0xBFFF: it also stops as soon as `high` has become 0xBFFE / 0xBFFF, i.e. after 14 - (trailing ones of high)
        // scalings. k scalings map v to 2^k (v - 2^15) + 2^15 (+ 2^k - 1 for `high`, + the k new stream bits for the code
        // register), all modulo 2^16 like the loop's masks.
        const uint32_t e3_run = (uint32_t)__builtin_clz(~(((nl & ~nh) & 0x7FFFu) << 17));
        const uint32_t e3_cap = 14u - (uint32_t)__builtin_ctz(~nh);
        const uint32_t k3 = nh > kRangeThreeQuarters ? 0u : (e3_run < e3_cap ? e3_run : e3_cap);
        {
            // the n bits that E1/E2 shift in and the k3 bits of the E3 scalings leave the window together (at most 30 of the
            // 33 or more buffered); beyond the end of the stream the last real bit repeats, and no bit at all reads as 0
            const uint32_t total = n + k3;
            const uint32_t avail = nbac - ridx;
            const uint32_t k = total < avail ? total : avail;
            uint32_t bits = take(k);                                       // first in time most significant
            const uint32_t sticky = bits & 1u;
            const uint32_t ext = total - k;
            bits = (bits << ext) | (sticky ? ((1u << ext) - 1u) : 0u);
            ridx += k;
            code = ((code << n) & kRangeMax) | (bits >> k3);
            code = (((code - 0x8000u) << k3) + 0x8000u + (bits & ((1u << k3) - 1u))) & kRangeMax;
            nl = (((nl - 0x8000u) << k3) + 0x8000u) & kRangeMax;
            nh = (((nh - 0x8000u) << k3) + 0x8000u + ((1u << k3) - 1u)) & kRangeMax;
        }
        low = nl;
        high = nh;
        // binarisation state (LosslessCoder.cpp:193-230, 254-276): a one advances the unary count up to L, a zero ends it
        const bool escape = bit && unary + 1u == L;
        const bool done = !bit || escape;
        uint32_t a = bit ? L : unary;
        bool bad = false;
        if (escape) {
            // Exp-Golomb suffix from the bypass stream (LosslessCoder.cpp:113-165)
            uint32_t nn = 0;
            for (;;) {
                if (yidx >= nbyp) { bad = true; break; }
                uint32_t yw;
                EAE_BYPASS_WORD(yw, yidx >> 5)
                const uint32_t b = (yw >> (yidx & 31u)) & 1u;
                yidx++;
                if (!b) break;
                nn++;
                if (nn > 16u) { bad = true; break; }
            }
            uint32_t suffix = 0;
            for (uint32_t q = 0; q < nn && !bad; q++) {
                if (yidx >= nbyp) { bad = true; break; }
                uint32_t yw;
                EAE_BYPASS_WORD(yw, yidx >> 5)
                suffix = (suffix << 1) | ((yw >> (yidx & 31u)) & 1u);
                yidx++;
            }
            a = (L + ((suffix + (1u << nn) - 1u) & 0xFFFFu)) & 0xFFFFu;    // uint16 arithmetic of the reference
            EAE_BYPASS_WORD(yword, yidx >> 5)                              // the sign now sits further on
            if (bad) err = 1u;
        }
        int v = (int)(int16_t)a;
        // sign bit of a non-zero symbol (LosslessCoder.cpp:39-56). A sign missing from the bypass stream (resource_error) is
        // noticed after the loop (yidx > nbyp): the lane runs on over zero bits and the general kernel reports the exact status
        const bool nonzero = done && v != 0;
        const uint32_t sign = (yword >> (yidx & 31u)) & 1u;
        v = nonzero && !sign ? -v : v;
        yidx += nonzero ? 1u : 0u;
        if (done) out[i] = (int16_t)v;
        i += done ? 1u : 0u;
        unary = done ? 0u : unary + 1u;
        pk = done ? p0 : pspec;
        wnext = wload;
        active = i < size && err == 0u;
    }
    if (live && !retry && (err != 0u || yidx > nbyp || i < size)) retry = true;   // something only the general kernel reports
    if (live) p.status[m] = retry ? RETRY : 0;
#undef EAE_WINDOW_WORD
#undef EAE_BYPASS_WORD
}

template <uint32_t WB, uint32_t WY, int MODE>
__global__ __launch_bounds__(64) void bac_decode_kernel(const SimdParams p) {
    __builtin_amdgcn_s_setprio(EAE_SIMD_PRIO);
    const uint32_t lane = threadIdx.x;
    if (MODE == 2) {
        const uint32_t short_blocks = (uint32_t)p.perm[0], long_blocks = (uint32_t)p.perm[1];
        if (blockIdx.x >= short_blocks + long_blocks) return;
        const uint32_t m = (uint32_t)p.perm[4u + blockIdx.x * 64u + lane];          // 0xFFFFFFFF pads the last block of a kind
        // two copies of the loop, each with its windows' bounds known at compile time (a block-uniform flag inside ONE copy
        // cost the short maps 18 %: three more compares and branches in a step of ~105 instructions)
        if (blockIdx.x >= short_blocks) bac_decode_body<WB, WY, MODE, true>(p, m);
        else bac_decode_body<WB, WY, MODE, false>(p, m);
    } else {
        bac_decode_body<WB, WY, MODE, MODE == 1>(p, blockIdx.x * 64u + lane);
    }
}

// Sorts the maps of a launch into blocks of 64 for bac_decode_kernel<.., 2>: first the maps whose streams fit the windows of
// WB / WY words (and the ones there is nothing to decode for: skipped, failed), in their own order, padded to a multiple of 64
// with 0xFFFFFFFF; then the long ones, padded likewise. perm[0] / perm[1] = number of blocks of each kind. One block of 1024
// threads walks the maps in chunks (a few thousand maps: microseconds).
__global__ __launch_bounds__(1024) void sort_maps_kernel(const SimdParams p, uint32_t wb_bits, uint32_t wy_bits) {
    __shared__ uint32_t wave_sums[2][16];
    __shared__ uint32_t base[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int32_t* slots = p.perm + 4;
    // pass A: how many long maps there are, i.e. where the long blocks start
    uint32_t n_long = 0;
    for (uint32_t m = tid; m < p.n_maps; m += 1024u) {
        const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
        if (row >= 0 && p.status[m] == 0 && (p.bac_bits[m] > wb_bits || p.bypass_bits[m] > wy_bits)) n_long++;
    }
    for (int off = 32; off > 0; off >>= 1) n_long += __shfl_down(n_long, off, 64);
    if (lane == 0) wave_sums[0][wave] = n_long;
    __syncthreads();
    if (tid == 0) {
        uint32_t total_long = 0;
        for (int w = 0; w < 16; w++) total_long += wave_sums[0][w];
        const uint32_t total_short = p.n_maps - total_long;
        const uint32_t short_blocks = (total_short + 63u) / 64u, long_blocks = (total_long + 63u) / 64u;
        p.perm[0] = (int32_t)short_blocks; p.perm[1] = (int32_t)long_blocks; p.perm[2] = (int32_t)total_short; p.perm[3] = (int32_t)total_long;
        base[0] = 0u;
        base[1] = short_blocks * 64u;
    }
    __syncthreads();
    const uint32_t total_short = (uint32_t)p.perm[2], total_long = (uint32_t)p.perm[3];
    const uint32_t short_end = ((total_short + 63u) / 64u) * 64u, long_start = short_end, long_end = long_start + ((total_long + 63u) / 64u) * 64u;
    // padding slots
    for (uint32_t i = total_short + tid; i < short_end; i += 1024u) slots[i] = -1;
    for (uint32_t i = long_start + total_long + tid; i < long_end; i += 1024u) slots[i] = -1;
    // pass B: stable placement, chunk by chunk (exclusive scans over the 1024 threads of a chunk)
    for (uint32_t m0 = 0; m0 < p.n_maps; m0 += 1024u) {
        const uint32_t m = m0 + tid;
        uint32_t is_long = 0, is_short = 0;
        if (m < p.n_maps) {
            const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
            is_long = (row >= 0 && p.status[m] == 0 && (p.bac_bits[m] > wb_bits || p.bypass_bits[m] > wy_bits)) ? 1u : 0u;
            is_short = 1u - is_long;
        }
        // both exclusive scans inside the wavefront at once: short maps in the low half, long ones in the high half
        uint32_t both_tot;
        const uint32_t both = wave_exclusive_scan(is_short | (is_long << 16), both_tot);
        const uint32_t off_s = both & 0xFFFFu, off_l = both >> 16, tot_s = both_tot & 0xFFFFu, tot_l = both_tot >> 16;
        if (lane == 0) { wave_sums[0][wave] = tot_s; wave_sums[1][wave] = tot_l; }
        __syncthreads();
        uint32_t before_s = 0, before_l = 0;
        for (uint32_t w = 0; w < wave; w++) { before_s += wave_sums[0][w]; before_l += wave_sums[1][w]; }
        if (m < p.n_maps) {
            if (is_long) slots[base[1] + before_l + off_l] = (int32_t)m;
            else slots[base[0] + before_s + off_s] = (int32_t)m;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cs = 0, cl = 0;
            for (int w = 0; w < 16; w++) { cs += wave_sums[0][w]; cl += wave_sums[1][w]; }
            base[0] += cs;
            base[1] += cl;
        }
        __syncthreads();
    }
}

// (4) decoded == encoded, one wavefront per map
__global__ __launch_bounds__(64) void compare_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x, lane = threadIdx.x;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row < 0 || p.status[m] != 0) return;
    const int16_t* a = p.symbols + (size_t)m * p.map_size;
    const int16_t* b = p.decoded + (size_t)m * p.map_size;
    int differ = 0;
    for (uint32_t i = lane; i < p.map_size; i += 64) differ |= (a[i] != b[i]);
    if (__any(differ) && lane == 0) p.status[m] = MISMATCH;
}

// status 0 -> RETRY for the coded maps: hands every map that has not failed to the general kernel
__global__ void mark_kernel(const SimdParams p) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= p.n_maps) return;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (row >= 0 && p.status[m] == 0) p.status[m] = RETRY;
}

uint32_t decision_capacity(uint32_t map_size, uint8_t L) { return ((map_size * ((uint32_t)L + 1u)) + 7u) & ~7u; }
bool fast_applies(uint8_t L) { return L >= 1 && L <= kMaxFastL; }

}  // namespace

extern "C" {

// workspace: [per-map decision counts][decisions of an encode | decoded symbols of a verify], 256-byte aligned pieces
static uint64_t round256(uint64_t v) { return (v + 255u) & ~(uint64_t)255u; }
uint64_t eae_hip_coder_workspace_bytes(uint32_t n_maps, uint32_t map_size, uint8_t L) {
    const uint64_t groups = ((uint64_t)n_maps + 63u) / 64u;
    const uint64_t decisions = fast_applies(L) ? groups * 64u * decision_capacity(map_size, L) : 0u;
    const uint64_t decoded = (uint64_t)n_maps * map_size * sizeof(int16_t);
    return 256u + round256((uint64_t)n_maps * sizeof(uint32_t)) + round256(decisions > decoded ? decisions : decoded) +
           round256(((uint64_t)n_maps + 128u + 4u) * sizeof(int32_t));        // + the decoder's sorted map order
}

static SimdParams make_params(uint32_t n_maps, uint32_t map_size, uint8_t L, const int16_t* symbols, const double* probs,
                              const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                              uint32_t* bypass_bits, int32_t* status, int32_t* stage, void* workspace) {
    SimdParams p{};
    p.n_maps = n_maps; p.map_size = map_size; p.L = L; p.dcap = decision_capacity(map_size, L);
    p.symbols = symbols; p.probs = probs; p.prob_row = prob_row; p.streams = streams; p.stride = stride;
    p.bac_bits = bac_bits; p.bypass_bits = bypass_bits; p.status = status; p.stage = stage;
    if (workspace) {
        uint8_t* ws = reinterpret_cast<uint8_t*>(round256(reinterpret_cast<uintptr_t>(workspace)));
        p.ndec = reinterpret_cast<uint32_t*>(ws);
        p.decisions = ws + round256((uint64_t)n_maps * sizeof(uint32_t));
        p.decoded = reinterpret_cast<int16_t*>(p.decisions);
        const uint64_t groups = ((uint64_t)n_maps + 63u) / 64u;
        const uint64_t decisions = fast_applies(L) ? groups * 64u * decision_capacity(map_size, L) : 0u;
        const uint64_t decoded = (uint64_t)n_maps * map_size * sizeof(int16_t);
        p.perm = reinterpret_cast<int32_t*>(p.decisions + round256(decisions > decoded ? decisions : decoded));
    }
    return p;
}

static int check_simd_layout(uint32_t map_size, uint8_t L, const uint8_t* streams, uint64_t stride) {
    const uint64_t half = stride / 2;
    return (half < (uint64_t)(round_up_to_byte(required_bits(map_size, L)) >> 3) + 16 || (stride & 15u) ||
            (reinterpret_cast<uintptr_t>(streams) & 7u)) ? 1 : 0;
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
    hipLaunchKernelGGL(bac_encode_kernel, dim3((n_maps + 63u) / 64u), dim3(64), (size_t)L * 64u * sizeof(double), s, p);
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
    if (!symbols_out && workspace_bytes < eae_hip_coder_workspace_bytes(n_maps, map_size, L)) return 1;
    if (n_maps == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    SimdParams p = make_params(n_maps, map_size, L, expected, probs, prob_row, const_cast<uint8_t*>(streams), stride,
                               const_cast<uint32_t*>(bac_bits), const_cast<uint32_t*>(bypass_bits), status, stage, workspace);
    if (symbols_out) p.decoded = symbols_out;
    if (!expected) (void)hipMemsetAsync(status, 0, (size_t)n_maps * sizeof(int32_t), s);   // a pure decode starts from a clean slate
    if (fast_applies(L) && map_size) {
        const dim3 grid((n_maps + 63u) / 64u);
        // with a workspace (large enough for the sorted map order): ONE launch over blocks of short maps and blocks of long
        // maps; without: first pass over every group of 64, second pass over the maps the first one found too long
        const bool sorted = workspace && workspace_bytes >= eae_hip_coder_workspace_bytes(n_maps, map_size, L);
        const dim3 sorted_grid((n_maps + 63u) / 64u + 1u);
        if (map_size > kMediumMapSize) {
            const size_t lds = decode_lds_bytes(L, kBacWindowWordsMedium, kBypassWindowWordsMedium);
            static const hipError_t medium_ok = [] {
                hipError_t e = hipSuccess;
                const void* kernels[3] = {
                    reinterpret_cast<const void*>(&bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 0>),
                    reinterpret_cast<const void*>(&bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 1>),
                    reinterpret_cast<const void*>(&bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 2>)};
                for (const void* k : kernels) {
                    e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                    if (e != hipSuccess) return e;
                }
                return e;
            }();
            if (medium_ok != hipSuccess) return (int)medium_ok;
            if (sorted) {
                hipLaunchKernelGGL(sort_maps_kernel, dim3(1), dim3(1024), 0, s, p, kBacWindowWordsMedium * 32u, kBypassWindowWordsMedium * 32u);
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 2>), sorted_grid, dim3(64), lds, s, p);
            } else {
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 0>), grid, dim3(64), lds, s, p);
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWordsMedium, kBypassWindowWordsMedium, 1>), grid, dim3(64), lds, s, p);
            }
        } else {
            const size_t lds = decode_lds_bytes(L, kBacWindowWords, kBypassWindowWords);
            if (sorted) {
                hipLaunchKernelGGL(sort_maps_kernel, dim3(1), dim3(1024), 0, s, p, kBacWindowWords * 32u, kBypassWindowWords * 32u);
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWords, kBypassWindowWords, 2>), sorted_grid, dim3(64), lds, s, p);
            } else {
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWords, kBypassWindowWords, 0>), grid, dim3(64), lds, s, p);
                hipLaunchKernelGGL((bac_decode_kernel<kBacWindowWords, kBypassWindowWords, 1>), grid, dim3(64), lds, s, p);
            }
        }
        // whatever a map reported that only the general kernel can name (errors of any kind)
        const int rc = eae_coder_generic_decode(n_maps, map_size, p.decoded, L, probs, prob_row, streams, stride, bac_bits,
                                                bypass_bits, status, stage, RETRY, s);
        if (rc) return rc;
    } else {
        // general kernel over every map that has not failed already
        hipLaunchKernelGGL(mark_kernel, dim3((n_maps + 255u) / 256u), dim3(256), 0, s, p);
        const int rc = eae_coder_generic_decode(n_maps, map_size, p.decoded, L, probs, prob_row, streams, stride, bac_bits,
                                                bypass_bits, status, stage, RETRY, s);
        if (rc) return rc;
    }
    if (expected) hipLaunchKernelGGL(compare_kernel, dim3(n_maps), dim3(64), 0, s, p);
    return (int)hipGetLastError();
}

}  // extern "C"
