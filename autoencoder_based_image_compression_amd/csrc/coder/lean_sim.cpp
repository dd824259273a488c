// lean_sim.cpp -- CPU model of the lean serial core (lean_step.h), one map at a time: TEST INFRASTRUCTURE for the arithmetic
// that hip/coder_simd.hip runs in every lane (built into lib/libeae_lean_sim.so; tests/test_lean_coder.py holds it against the
// host library, i.e. against coder_core.h, i.e. against the byte streams of the reference build).
//
// What is modelled exactly as the kernels do it: the per-decision step functions (encode_step / decode_step / shift_code), the
// record format, the rule by which records become stream bits (first leaving bit, the pending-E3 queue, the other leaving bits;
// the stop record), the decoder's bit window with the reference's end-of-stream behaviour, and the truncated-unary context
// tracking that yields one prefix byte per symbol. What is NOT modelled: the data-parallel organisation (wave scans, LDS tiles)
// of the emit / prefix passes -- those are checked on the GPU against bytes.
#include <stdint.h>

#include <cstddef>
#include <vector>

#include "lean_step.h"

using namespace eae_lean;
using eae_core::Bitstream;

extern "C" {

// One map: symbols -> arithmetic-coded stream (bytes LSB-first like Bitstream.cpp:30-59) + its length in bits. Returns 0, or
// eae_core::CAPACITY when the stream outgrows `capacity_bits` (the kernels hand such maps to the general kernel), or
// eae_core::PROBABILITY for a probability outside (0, 1) that is used.
int eae_lean_sim_encode(uint32_t size, const int16_t* in, uint32_t L, const double* probabilities, uint8_t* bac_bytes,
                        uint32_t capacity_bits, uint32_t* bac_bits, uint32_t* nb_decisions, uint32_t* records_out) {
    std::vector<uint32_t> records;
    Interval s = interval_init();
    for (uint32_t i = 0; i < size; i++) {
        const int v = (int)in[i];
        const uint32_t a = (uint32_t)(v < 0 ? -v : v);
        const uint32_t ones = a < L ? a : L;
        for (uint32_t q = 0; q < ones + (a < L ? 1u : 0u); q++) {
            const double p = probabilities[q];
            if (!(p > 0. && p < 1.)) return eae_core::PROBABILITY;
            records.push_back(encode_step(s, scale_probability(p), q < ones));
        }
    }
    *nb_decisions = (uint32_t)records.size();
    records.push_back(stop_record(s));
    if (records_out)
        for (std::size_t j = 0; j < records.size(); j++) records_out[j] = records[j];
    // records -> bits, sequentially (the emit pass computes the same positions with prefix sums)
    Bitstream bs;
    bs.init_writer(bac_bytes, capacity_bits);
    uint32_t pending = 0;
    for (uint32_t rec : records) {
        const uint32_t n = record_n(rec);
        if (record_is_stop(rec)) pending++;
        if (n) {
            const uint32_t lead = record_leaving(rec);
            const uint32_t first = lead >> 31;
            if (bs.put(first, 1)) return eae_core::CAPACITY;
            if (bs.put_run(first ^ 1u, pending)) return eae_core::CAPACITY;
            pending = 0;
            for (uint32_t t = 1; t < n; t++)
                if (bs.put((lead >> (31u - t)) & 1u, 1)) return eae_core::CAPACITY;
        }
        pending += record_k(rec);
    }
    bs.flush();
    *bac_bits = bs.write_index;
    return 0;
}

// One map: arithmetic-coded stream -> the truncated-unary prefix (0..L) of every symbol. Returns 0, an error code of the core, or
// eae_lean_sim_outside if the code register ever left the interval -- which no stream, however damaged, can cause (lean_step.h).
enum { eae_lean_sim_outside = -100 };
int eae_lean_sim_decode_prefixes(uint32_t size, uint32_t L, const double* probabilities, const uint8_t* bac_bytes,
                                 uint32_t bac_bits, uint8_t* prefixes) {
    // the stream as an array of bits with the reference's end-of-stream rule applied per step
    auto bit_at = [&](uint32_t i) { return (uint32_t)(bac_bytes[i >> 3] >> (i & 7u)) & 1u; };
    uint32_t pos = 0;
    auto take = [&](uint32_t count) {       // `count` bits, first in time most significant; sticky beyond the end
        uint32_t bits = 0, sticky = 0;
        for (uint32_t t = 0; t < count; t++) {
            if (pos < bac_bits) sticky = bit_at(pos++);
            bits = (bits << 1) | sticky;
        }
        return bits;
    };
    Interval s = interval_init();
    uint32_t code32 = take(16) << 16;       // start_decoding (BinaryArithmeticCoder.cpp:104-122)
    std::vector<double> scaled(L);
    for (uint32_t q = 0; q < L; q++) scaled[q] = scale_probability(probabilities[q]);
    for (uint32_t i = 0; i < size; i++) {
        uint32_t unary = 0;
        for (;;) {
            const double p = probabilities[unary];
            if (!(p > 0. && p < 1.)) return eae_core::PROBABILITY;
            if (!code_inside(s, code32)) return eae_lean_sim_outside;       // cannot happen (lean_step.h: code_inside); checked on every step
            const DecodeStep d = decode_step(s, code32, scaled[unary]);
            code32 = shift_code(code32, d, take(d.take));
            if (!d.one) break;
            unary++;
            if (unary == L) break;
        }
        prefixes[i] = (uint8_t)unary;
    }
    return 0;
}

}  // extern "C"
