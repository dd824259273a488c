// coder_device.hip -- the lossless coder on the GPU: one feature map per lane.
//
// Replaces the 127 compress_lossless calls per image of lossless/compression.py:76-81 (one per non-exception map) by ONE
// launch over every map of a batch of images, symbols and streams resident in HBM. The coder of one map is a serial
// bit-exact state machine (16-bit interval + pending-bit counter), so the parallelism is across maps: 3048 independent
// streams for a batch of 24 Kodak images. Lanes of a wavefront run different maps; `lanes` (<= 64) lanes per 64-thread
// block are used so that the launch spreads over many CUs and a wave only waits for its slowest few maps.
// The arithmetic is coder_core.h, the same source the host library compiles (interval update in IEEE double, -ffp-contract=off).
// This is integer/bit work: no MFMA, no LDS; HBM traffic is the symbols (2 B each, read twice) and the streams.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../coder/coder_core.h"
#include "eae_hip.h"

// gfx950: a 64-bit shift (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose shift amount sits in the LAST register of the wave's VGPR
// allocation gives wrong results whenever other waves share the SIMD (csrc/isa_guard.py rule 2, DESIGN.md section 5: the fault of
// round 3's first decoder core). These kernels shift 64-bit windows by computed amounts, so each instantiation names a register just
// above what it needs in an empty asm at kernel entry (see coder_simd.hip for why not v63 everywhere). That only PADS the allocation
// (next_free_vgpr becomes n + 1 and vn is dead across that one point): it does not forbid the allocator to use vn later, and it
// works because the kernels' pressure stays below n. The ENFORCEMENT is csrc/isa_guard.py rule 2 on the shipped ISA -- run by
// build() and, whatever build() was told to skip, by tests/test_isa_guard.py::test_the_shipped_library_is_clean on the library that
// ships; tests/test_isa_guard.py also pins the allocations, so that one more live value shows up as a red test, not as wrong bits.
#ifndef EAE_DECODE_TOPUP_ZEROS
#define EAE_KEEP_VGPR_FREE(n) asm volatile("; v" #n " reserved: the last register of the allocation holds no operand" ::: "v" #n)
#else      // the first decoder core is kept as it was built (40 of 40 registers): scratch/r04, tests/test_isa_guard.py
#define EAE_KEEP_VGPR_FREE(n)
#endif

namespace {

using namespace eae_core;

#ifndef EAE_CODER_PRIO
#define EAE_CODER_PRIO 3
#endif

struct CoderParams {
    uint32_t n_maps, map_size, L, lanes;
    const int16_t* symbols;       // [n_maps][map_size]
    int16_t* reconstruction;      // nullable
    const double* probs;          // [rows][L]
    const int32_t* prob_row;      // nullable: row per map, < 0 = skip
    uint8_t* streams;             // [n_maps][stride]: BAC at +0, bypass at +stride/2
    uint64_t stride;
    uint32_t* bac_bits;
    uint32_t* bypass_bits;
    int32_t* status;
    int32_t* stage;               // nullable
    int32_t only_status;          // non-zero: code only the maps whose status currently holds this value (fallback pass)
};

// Eight int16 symbols per 16-byte load when the map is 16-byte aligned (every Kodak-sized map is), else one by one. The
// load of the NEXT eight is issued when the current eight are handed out, so its latency hides behind their coding.
struct SymbolReader {
    const int16_t* base;
    uint64_t lo, hi;
    uint4 ahead;
    uint32_t size;
    bool wide;
    __device__ __forceinline__ void init(const int16_t* p, uint32_t n) {
        base = p;
        size = n;
        wide = ((n & 7u) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0);
        lo = hi = 0;
        ahead = make_uint4(0, 0, 0, 0);
        if (wide && n) ahead = *reinterpret_cast<const uint4*>(p);
    }
    __device__ __forceinline__ int16_t next(uint32_t i) {
        if (wide) {
            if ((i & 7u) == 0) {
                lo = (uint64_t)ahead.x | ((uint64_t)ahead.y << 32);
                hi = (uint64_t)ahead.z | ((uint64_t)ahead.w << 32);
                if (i + 8u < size) ahead = *reinterpret_cast<const uint4*>(base + i + 8u);
            }
            const int16_t s = (int16_t)(lo & 0xFFFFu);
            lo = (lo >> 16) | (hi << 48);
            hi >>= 16;
            return s;
        }
        return base[i];
    }
};

// Most symbols of a trained (or random) model at its operating points are zero. A zero is ONE arithmetic-coder decision,
// bit 0 with probabilities[0] (LosslessCoder.cpp:167-191 with input 0; no suffix, no sign), and most of the time that
// decision neither shifts bits out nor grows the E3 queue. These two helpers do exactly that case in a dozen
// instructions and touch nothing when it does not apply; the caller then runs the general code of coder_core.h on
// the untouched state. Bit-exactness with the general path is covered by tests/test_coder_device.py (every stream is
// compared with the host library's, which has no such shortcut).
__device__ __forceinline__ bool encode_zero_fast(Bac& b, double p0) {
    const uint32_t mid = b.low + (uint32_t)(p0 * (double)(b.high - b.low));   // Bac::update_middle
    // top bits of (low, new high) differ -> no E1/E2; and no E3 (BinaryArithmeticCoder.cpp:238)
    if (!((b.low ^ mid) & 0x8000u) || (b.low > kRangeQuarter && mid <= kRangeThreeQuarters)) return false;
    b.middle = mid;
    b.high = mid;
    return true;
}
__device__ __forceinline__ bool decode_zero_fast(Bac& b, double p0) {
    const uint32_t mid = b.low + (uint32_t)(p0 * (double)(b.high - b.low));
    if (!(b.code >= b.low && b.code <= mid)) return false;                    // the decision is a one
    if (!((b.low ^ mid) & 0x8000u) || (b.low > kRangeQuarter && mid <= kRangeThreeQuarters)) return false;
    b.middle = mid;
    b.high = mid;
    return true;
}

// Probabilities of the lanes of a block, interleaved in LDS ([i][lane]: conflict-free, one ds_read per decision instead of a
// global load on the serial chain). Each lane reads back only what it wrote itself.
extern __shared__ double lds_probabilities[];
template <bool UNIFORM>
__device__ __forceinline__ void stage_probabilities(LosslessCoder& c, const CoderParams& p, int32_t row) {
    const uint32_t stride = UNIFORM ? 1u : p.lanes;
    double* mine = lds_probabilities + (UNIFORM ? 0u : threadIdx.x);
    const double* src = p.probs + (size_t)row * p.L;
    for (uint32_t i = 0; i < p.L; i++) mine[i * stride] = src[i];
    c.L = p.L;
    c.prob_stride = stride;
    c.probabilities = mine;
}

// Which map a thread codes. UNIFORM = one wavefront per map, every lane computing the same thing: all coder state is
// then wave-uniform, the compiler keeps it in SGPRs, runs the integer state machine on the scalar unit and branches
// without touching the exec mask; a wave needs few VGPRs, so it fits next to the transform kernels' waves without
// lowering their occupancy. Otherwise `lanes` maps per 64-thread block, one per lane.
template <bool UNIFORM>
__device__ __forceinline__ bool map_of_thread(const CoderParams& p, uint32_t& m) {
    if (UNIFORM) {
        m = blockIdx.x;
        return true;
    }
    if (threadIdx.x >= p.lanes) return false;
    m = blockIdx.x * p.lanes + threadIdx.x;
    return m < p.n_maps;
}

// MODE: 0 = encode + decode into `reconstruction`; 1 = encode only; 2 = encode + decode + compare (status MISMATCH).
template <int MODE, bool UNIFORM>
__global__ __launch_bounds__(64) void coder_maps_kernel(const CoderParams p) {
    if constexpr (UNIFORM) { EAE_KEEP_VGPR_FREE(23); } else { EAE_KEEP_VGPR_FREE(55); }      // allocations of 24 / 56
    // a handful of latency-bound waves next to the transforms' MFMA waves: let them issue whenever they are ready
    __builtin_amdgcn_s_setprio(EAE_CODER_PRIO);
    uint32_t m;
    if (!map_of_thread<UNIFORM>(p, m)) return;
    if (p.only_status && p.status[m] != p.only_status) return;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    const int16_t* in = p.symbols + (size_t)m * p.map_size;
    int st = STAGE_NONE, s = OK;
    uint32_t nbac = 0, nbyp = 0;
    if (row < 0) {  // exception map: passed through (compression.py:68-75), costed by the caller from its histogram
        if (MODE == 0)
            for (uint32_t i = 0; i < p.map_size; i++) p.reconstruction[(size_t)m * p.map_size + i] = in[i];
    } else {
        const uint32_t req = required_bits(p.map_size, p.L);
        LosslessCoder c;
        c.bac.init();
        c.bac.bs.init_writer(p.streams + (uint64_t)m * p.stride, req);
        c.bypass.init_writer(p.streams + (uint64_t)m * p.stride + p.stride / 2, req);
        stage_probabilities<UNIFORM>(c, p, row);
        SymbolReader rd;
        rd.init(in, p.map_size);
        // LosslessCoder::encode_map with the wide symbol reader
        const double p0 = p.L ? c.probability(0) : 0.;
        const bool fast = p0 > 0. && p0 < 1.;          // else the general path reports the error
        for (uint32_t i = 0; i < p.map_size; i++) {
            const int16_t v = rd.next(i);
            if (fast && v == 0 && encode_zero_fast(c.bac, p0)) continue;
            s = c.write_signed_ueg0(v);
            if (s) { st = STAGE_ENCODING; break; }
        }
        if (!s) {
            s = c.bac.stop_encoding();
            if (s) st = STAGE_STOP;
            else c.bypass.flush();
        }
        if (!s) {
            nbac = c.bac.bs.write_index;
            nbyp = c.bypass.write_index;
            if (MODE != 1) {
                __threadfence();   // this lane's own stores, read back below through byte loads
                s = c.bac.start_decoding();
                if (s) st = STAGE_START;
                else {
                    rd.init(in, p.map_size);
                    int mismatch = 0;
                    for (uint32_t i = 0; i < p.map_size; i++) {
                        int16_t v = 0;
                        if (!(fast && decode_zero_fast(c.bac, p0))) {
                            s = c.read_signed_ueg0(v);
                            if (s) { st = STAGE_DECODING; break; }
                        }
                        if (MODE == 0) p.reconstruction[(size_t)m * p.map_size + i] = v;
                        else mismatch |= (v != rd.next(i));
                    }
                    if (!s && mismatch) s = MISMATCH;
                }
            }
        }
    }
    p.bac_bits[m] = nbac;
    p.bypass_bits[m] = nbyp;
    p.status[m] = s;
    if (p.stage) p.stage[m] = st;
}

// The decoder side on its own: streams + bit lengths -> symbols (COMPARE = false), or -> a comparison with the symbols
// that were encoded (COMPARE = true: nothing is stored, status MISMATCH on a difference, maps whose status is already
// non-zero -- a failed encode -- are left alone).
template <bool COMPARE, bool UNIFORM>
__global__ __launch_bounds__(64) void decoder_maps_kernel(const CoderParams p) {
    if constexpr (UNIFORM) { if constexpr (COMPARE) { EAE_KEEP_VGPR_FREE(31); } else { EAE_KEEP_VGPR_FREE(23); } }      // allocations of 32 / 24,
    else { if constexpr (COMPARE) { EAE_KEEP_VGPR_FREE(55); } else { EAE_KEEP_VGPR_FREE(47); } }                        // 56 / 48
    __builtin_amdgcn_s_setprio(EAE_CODER_PRIO);
    uint32_t m;
    if (!map_of_thread<UNIFORM>(p, m)) return;
    if (p.only_status && p.status[m] != p.only_status) return;
    const int32_t row = p.prob_row ? p.prob_row[m] : (int32_t)m;
    if (COMPARE && !p.only_status && p.status[m] != 0) return;
    int st = STAGE_NONE, s = OK;
    if (row >= 0) {
        LosslessCoder c;
        c.bac.init();
        c.bac.bs.init_reader(p.streams + (uint64_t)m * p.stride, p.bac_bits[m]);
        c.bypass.init_reader(p.streams + (uint64_t)m * p.stride + p.stride / 2, p.bypass_bits[m]);
        stage_probabilities<UNIFORM>(c, p, row);
        const double p0 = p.L ? c.probability(0) : 0.;
        const bool fast = p0 > 0. && p0 < 1.;
        s = c.bac.start_decoding();
        if (s) st = STAGE_START;
        else {
            SymbolReader rd;
            if (COMPARE) rd.init(p.symbols + (size_t)m * p.map_size, p.map_size);
            int mismatch = 0;
            for (uint32_t i = 0; i < p.map_size; i++) {
                int16_t v = 0;
                if (!(fast && decode_zero_fast(c.bac, p0))) {
                    s = c.read_signed_ueg0(v);
                    if (s) { st = STAGE_DECODING; break; }
                }
                if (COMPARE) mismatch |= (v != rd.next(i));
                else p.reconstruction[(size_t)m * p.map_size + i] = v;
            }
            if (!s && mismatch) s = MISMATCH;
        }
    }
    p.status[m] = s;
    if (p.stage) p.stage[m] = st;
}

int check_layout(uint32_t map_size, uint8_t L, const uint8_t* streams, uint64_t stride) {
    const uint64_t half = stride / 2;
    if (half < (uint64_t)(round_up_to_byte(required_bits(map_size, L)) >> 3) + 16 || (stride & 15u) ||
        (reinterpret_cast<uintptr_t>(streams) & 7u))
        return 1;  // EAE_CAPACITY_ERROR of eae_coder_encode_maps
    return 0;
}

// lanes_per_wave: 1..64 = that many maps per 64-thread block, one per lane; <= 0 = one wavefront per map (uniform).
struct Geometry {
    bool uniform;
    uint32_t lanes;
    dim3 grid, block;
    size_t lds;
};
Geometry geometry(uint32_t n_maps, uint8_t L, int lanes_per_wave) {
    Geometry g;
    g.uniform = lanes_per_wave <= 0;
    g.lanes = g.uniform ? 1u : (uint32_t)(lanes_per_wave > 64 ? 64 : lanes_per_wave);
    g.grid = dim3((n_maps + g.lanes - 1) / g.lanes);
    g.block = dim3(64);
    g.lds = (size_t)g.lanes * L * sizeof(double);
    return g;
}

}  // namespace

// Fallback passes of coder_simd.hip: the general per-lane kernels over the maps whose status holds `only_status`.
int eae_coder_generic_encode(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                             const int32_t* prob_row, uint8_t* streams, uint64_t stride, uint32_t* bac_bits,
                             uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream) {
    const Geometry g = geometry(n_maps, L, 64);
    CoderParams p{n_maps, map_size, L, g.lanes, symbols, nullptr, probs, prob_row, streams, stride,
                  bac_bits, bypass_bits, status, stage, only_status};
    hipLaunchKernelGGL((coder_maps_kernel<1, false>), g.grid, g.block, g.lds, stream, p);
    return (int)hipGetLastError();
}
int eae_coder_generic_decode(uint32_t n_maps, uint32_t map_size, int16_t* out, uint8_t L, const double* probs,
                             const int32_t* prob_row, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                             const uint32_t* bypass_bits, int32_t* status, int32_t* stage, int only_status, hipStream_t stream) {
    const Geometry g = geometry(n_maps, L, 64);
    CoderParams p{n_maps, map_size, L, g.lanes, nullptr, out, probs, prob_row, const_cast<uint8_t*>(streams), stride,
                  const_cast<uint32_t*>(bac_bits), const_cast<uint32_t*>(bypass_bits), status, stage, only_status};
    hipLaunchKernelGGL((decoder_maps_kernel<false, false>), g.grid, g.block, g.lds, stream, p);
    return (int)hipGetLastError();
}

extern "C" {

uint64_t eae_hip_coder_stream_stride_bytes(uint32_t map_size, uint8_t L) {
    const uint64_t half = (uint64_t)(round_up_to_byte(required_bits(map_size, L)) >> 3) + 16;
    return 2 * ((half + 15u) & ~(uint64_t)15u);
}

int eae_hip_coder_compress_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, int16_t* reconstruction,
                                uint8_t L, const double* probs, const int32_t* prob_row, uint8_t* streams, uint64_t stride,
                                uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status, int32_t* stage, int mode,
                                int lanes_per_wave, void* stream) {
    if (!symbols || !probs || !streams || !bac_bits || !bypass_bits || !status) return -1;
    if (mode < 0 || mode > 2 || (mode == 0 && !reconstruction)) return -1;
    if (check_layout(map_size, L, streams, stride)) return 1;
    if (n_maps == 0) return 0;
    const Geometry g = geometry(n_maps, L, lanes_per_wave);
    CoderParams p{n_maps, map_size, L, g.lanes, symbols, reconstruction, probs, prob_row, streams, stride,
                  bac_bits, bypass_bits, status, stage, 0};
    hipStream_t s = (hipStream_t)stream;
    // the scalar-cache loads of the one-wave-per-map form are not coherent with the stores of the same launch: only the
    // encode-only mode uses it; the combined modes read their own streams back through the vector path
    if (mode == 1 && g.uniform) hipLaunchKernelGGL((coder_maps_kernel<1, true>), g.grid, g.block, g.lds, s, p);
    else if (mode == 0) hipLaunchKernelGGL((coder_maps_kernel<0, false>), g.grid, g.block, g.lds, s, p);
    else if (mode == 1) hipLaunchKernelGGL((coder_maps_kernel<1, false>), g.grid, g.block, g.lds, s, p);
    else hipLaunchKernelGGL((coder_maps_kernel<2, false>), g.grid, g.block, g.lds, s, p);
    return (int)hipGetLastError();
}

int eae_hip_coder_decode_maps(uint32_t n_maps, uint32_t map_size, int16_t* symbols_out, uint8_t L, const double* probs,
                              const int32_t* prob_row, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                              const uint32_t* bypass_bits, int32_t* status, int32_t* stage, int lanes_per_wave, void* stream) {
    if (!symbols_out || !probs || !streams || !bac_bits || !bypass_bits || !status) return -1;
    if (n_maps == 0) return 0;
    const Geometry g = geometry(n_maps, L, lanes_per_wave);
    CoderParams p{n_maps, map_size, L, g.lanes, nullptr, symbols_out, probs, prob_row,
                  const_cast<uint8_t*>(streams), stride, const_cast<uint32_t*>(bac_bits), const_cast<uint32_t*>(bypass_bits),
                  status, stage, 0};
    if (g.uniform) hipLaunchKernelGGL((decoder_maps_kernel<false, true>), g.grid, g.block, g.lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((decoder_maps_kernel<false, false>), g.grid, g.block, g.lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int eae_hip_coder_verify_maps(uint32_t n_maps, uint32_t map_size, const int16_t* expected, uint8_t L, const double* probs,
                              const int32_t* prob_row, const uint8_t* streams, uint64_t stride, const uint32_t* bac_bits,
                              const uint32_t* bypass_bits, int32_t* status, int32_t* stage, int lanes_per_wave, void* stream) {
    if (!expected || !probs || !streams || !bac_bits || !bypass_bits || !status) return -1;
    if (n_maps == 0) return 0;
    const Geometry g = geometry(n_maps, L, lanes_per_wave);
    CoderParams p{n_maps, map_size, L, g.lanes, expected, nullptr, probs, prob_row,
                  const_cast<uint8_t*>(streams), stride, const_cast<uint32_t*>(bac_bits), const_cast<uint32_t*>(bypass_bits),
                  status, stage, 0};
    if (g.uniform) hipLaunchKernelGGL((decoder_maps_kernel<true, true>), g.grid, g.block, g.lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((decoder_maps_kernel<true, false>), g.grid, g.block, g.lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

}  // extern "C"
