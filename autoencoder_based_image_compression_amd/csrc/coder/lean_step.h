// lean_step.h -- the SERIAL core of the binary arithmetic coder, stripped to what is inherently serial, in the form the
// 64-maps-per-wavefront kernels of hip/coder_simd.hip run it (and, compiled by g++, the CPU model lean_sim.cpp that pins the
// arithmetic to coder_core.h before any kernel runs).
//
// coder_core.h follows kodak_tensorflow/lossless/c++/source/BinaryArithmeticCoder.cpp statement for statement: 16-bit `low`,
// `high`, `code` in the low halves of 32-bit registers, masks after every shift, bit I/O inside the interval update. On a 64-wide
// machine the arithmetic coder of a feature map is ONE dependent instruction chain (about a hundred vector instructions per
// binary decision in round 2's kernels), and its length -- not any throughput -- is what the transforms of the batches in flight
// have to cover. So the chain is cut down to the interval arithmetic itself:
//
//   * the interval lives in the TOP halves of two registers, `low << 16` and `(~high) << 16`: a left shift then drops the leaving
//     bits and refills `low` with zeros and `high` with ones by itself (no masks), the number of leading equal bits of low and high
//     is clz(~(lo ^ hc)) and cannot exceed 16 (the empty low halves differ in every bit), and the decoder's comparison
//     `code > middle` is an unsigned comparison of top-aligned words;
//   * floor(p * (high - low)) (BinaryArithmeticCoder.cpp:154) is evaluated as floor((p * 2^-16) * (double)((high - low) << 16)):
//     scaling by a power of two changes no mantissa bit, so the product is the reference's, and the scaled probability is
//     prepared once per map;
//   * the E3 loop (BinaryArithmeticCoder.cpp:238-245, 300-318) is the closed form of coder_core.h / round 2 in this representation:
//     k scalings delete the k bits below the top bit of low and high (ones in low, zeros in high), and set the new top bits to
//     0 / 1; for the code register: shift by k, then flip the top bit (2^k (v - 2^15) + 2^15 = (v << k) ^ 2^15 modulo 2^16, k >= 1);
//   * the ENCODER does no bit I/O at all: per decision it leaves a 32-bit record (the 16 bits that may leave, the number n of
//     E1/E2 shifts, the number k of E3 scalings), and a data-parallel pass assembles the stream from the records (prefix sums of
//     the emitted lengths; the pending-E3 queue is a segmented sum of the k's): see emit_record / LeanEmitter below;
//   * the DECODER core only decodes decisions: it tracks the truncated-unary context (needed for the next probability) and stores
//     one byte per symbol (the unary prefix, 0..L); signs, Exp-Golomb suffixes (LosslessCoder.cpp:39-56, 113-165) and the
//     comparison with the encoder's input are a data-parallel pass over those bytes and the bypass stream.
//
// Everything here is plain integer / IEEE-double arithmetic on values, no memory access: the same source runs in a lane of a
// wavefront and in the CPU model. Bit-exactness against coder_core.h (hence against the reference build, whose streams
// coder_core.h reproduces byte for byte) is tested on the CPU by tests/test_lean_coder.py and on the GPU by the byte-level tests of
// tests/test_coder_device.py.
#pragma once
#include <stdint.h>

#include "coder_core.h"

namespace eae_lean {

using eae_core::kRangeMax;

EAE_HD uint32_t clz32(uint32_t x) { return (uint32_t)__builtin_clz(x); }   // x != 0
EAE_HD uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }   // x != 0

// The coding interval [low, high] of BinaryArithmeticCoder.h:9-117, top-aligned: lo = low << 16, hc = (~high & 0xFFFF) << 16.
struct Interval {
    uint32_t lo, hc;
};
EAE_HD Interval interval_init() { return Interval{0u, 0u}; }                 // low = 0, high = 0xFFFF (BinaryArithmeticCoder.cpp:35-42)
EAE_HD uint32_t interval_low16(const Interval& s) { return s.lo >> 16; }
EAE_HD uint32_t interval_high16(const Interval& s) { return (~s.hc) >> 16; }

// p * 2^-16, prepared once per context: exact for every p that can matter (a p below 2^-1006 gives floor(...) = 0 either way)
EAE_HD double scale_probability(double p) { return p * (1.0 / 65536.0); }

// middle = low + floor(p * (high - low)) (update_middle, BinaryArithmeticCoder.cpp:144-156), top-aligned. 0 < p < 1 => middle < high.
EAE_HD uint32_t middle32(const Interval& s, double p_scaled) {
    const uint32_t range32 = 0xFFFF0000u - s.hc - s.lo;               // (high - low) << 16
    const uint32_t t = (uint32_t)(p_scaled * (double)range32);        // floor of a non-negative double
    return s.lo + (t << 16);
}

// encode_bit / the decoder's choice (BinaryArithmeticCoder.cpp:158-180, 254-273): a one keeps (middle, high], a zero [low, middle]
EAE_HD void narrow(Interval& s, uint32_t mid32, bool one) {
    const uint32_t lo_one = mid32 + 0x10000u;
    const uint32_t hc_zero = 0xFFFF0000u - mid32;                     // (~middle & 0xFFFF) << 16
    s.lo = one ? lo_one : s.lo;
    s.hc = one ? s.hc : hc_zero;
}

struct Renorm {
    uint32_t n;        // E1/E2 shifts: the leading bits low and high have in common (0..16)
    uint32_t k;        // E3 scalings that follow (0..14)
    uint32_t leaving;  // encoder: the interval's top 16 bits BEFORE the shifts, top-aligned; its n leading bits leave, first in time first
};

// rescale_encoding / rescale_decoding (BinaryArithmeticCoder.cpp:182-252, 275-320) on the interval, in closed form.
EAE_HD Renorm renormalise(Interval& s) {
    Renorm r;
    r.leaving = s.lo;
    r.n = clz32(~(s.lo ^ s.hc));                                      // lo and hc differ exactly where low and high agree
    const uint32_t a = s.lo << r.n, b = s.hc << r.n;                  // top bits now 0 / 0 (low: 0, high: 1)
    // E3 fires once per leading position below the top where low has a one and high a zero (a and b both have a one) ...
    const uint32_t run = clz32(~(a & b) & 0x7FFFFFFFu) - 1u;          // the argument's low half is all ones: never 0
    // ... but the reference compares high with 3 * 0x3FFF = 0xBFFD, not 0xBFFF: no scaling at all when high is 0xBFFE / 0xBFFF,
    // and the loop also stops once high has BECOME one of them: after 14 - (trailing ones of high) scalings
    const uint32_t cap = 30u - ctz32(b | 0x80000000u);                // b == 0 (high = 0xFFFF): 0xFFFFFFFF, and run is 0 anyway
    const bool eligible = b >= 0x40020000u;                           // high <= 0xBFFD (with bit 14 of high clear, which `run` needs)
    r.k = eligible ? (run < cap ? run : cap) : 0u;
    s.lo = (a << r.k) & 0x7FFFFFFFu;
    s.hc = (b << r.k) & 0x7FFFFFFFu;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoder: one decision -> one record. Bits 31..16: `leaving`; bits 12..8: n; bits 3..0: k.
// ---------------------------------------------------------------------------------------------------------------------
EAE_HD uint32_t encode_step(Interval& s, double p_scaled, bool one) {
    narrow(s, middle32(s, p_scaled), one);
    const Renorm r = renormalise(s);
    return (r.leaving & 0xFFFF0000u) | (r.n << 8) | r.k;
}
EAE_HD uint32_t record_n(uint32_t rec) { return (rec >> 8) & 31u; }
EAE_HD uint32_t record_k(uint32_t rec) { return rec & 15u; }
EAE_HD uint32_t record_leaving(uint32_t rec) { return rec & 0xFFFF0000u; }
// The record of stop_encoding (BinaryArithmeticCoder.cpp:61-102): one more pending bit, then the bit `low >= 0x3FFF` with the
// whole queue behind it -- i.e. a decision that shifts ONE bit out, in front of which the queue has grown by one. The emitter
// adds that one to the pending count when it meets the flag (bit 4).
EAE_HD uint32_t stop_record(const Interval& s) {
    const uint32_t b = interval_low16(s) < eae_core::kRangeQuarter ? 0u : 1u;
    return (b << 31) | (1u << 8) | 16u;
}
EAE_HD bool record_is_stop(uint32_t rec) { return (rec & 16u) != 0u; }

// ---------------------------------------------------------------------------------------------------------------------
// Decoder: the code register is top-aligned like the interval (code << 16); `window` delivers the next stream bits.
// ---------------------------------------------------------------------------------------------------------------------
struct DecodeStep {
    bool one;          // the decoded decision
    uint32_t take;     // n + k: stream bits this decision consumes
    uint32_t flip;     // k != 0: the code register's top bit is flipped after the shift
};
// The reference's decode_bit (BinaryArithmeticCoder.cpp:254-273) narrows the interval only when low <= code <= high and otherwise
// leaves interval and decision as they were; this step always narrows. The two agree on EVERY input, damaged streams included,
// because the code register cannot leave the interval: it starts inside (any 16 bits lie in [0, 0xFFFF]); a decision keeps the
// half that holds it; E1/E2 shift out a bit that low, code and high share and E3 subtracts the same quarter from all three; and
// the bit that enters at the bottom (0 for low, 1 for high, the stream's or the repeated last one for the code) cannot break
// low <= code <= high. `code_inside` states the invariant; the CPU model checks it on every step of every damaged stream the
// tests feed it (tests/test_lean_coder.py), and the GPU test decodes such streams against the host library (tests/test_coder_device.py).
EAE_HD bool code_inside(const Interval& s, uint32_t code32) { return code32 >= s.lo && code32 <= ~s.hc; }   // ~hc = (high << 16) | 0xFFFF
EAE_HD DecodeStep decode_step(Interval& s, uint32_t code32, double p_scaled) {
    const uint32_t mid = middle32(s, p_scaled);
    DecodeStep d;
    d.one = code32 >= mid + 0x10000u;                                 // code > middle (BinaryArithmeticCoder.cpp:263-272)
    narrow(s, mid, d.one);
    const Renorm r = renormalise(s);
    d.take = r.n + r.k;
    d.flip = r.k ? 0x80000000u : 0u;
    return d;
}
// `bits`: the `take` stream bits, first in time most significant, right-aligned (beyond the end of the stream: the last bit read
// in THIS step repeated, zeros if none was read -- the `storage` of rescale_decoding, BinaryArithmeticCoder.cpp:275-277, 303-310).
EAE_HD uint32_t shift_code(uint32_t code32, const DecodeStep& d, uint32_t bits) {
    return (((code32 << d.take) | (bits << 16)) ^ d.flip);            // take <= 30; bits that would pass the top have left anyway
}

}  // namespace eae_lean
