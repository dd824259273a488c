/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" view over the REAL reference coder classes, compiled from the sources where they lie under
 * /root/reference/kodak_tensorflow/lossless/c++/source (never copied into this repo): see oracle/Makefile, target
 * _ref/libref_coder.so. This file is ours; it only (a) includes the reference headers, (b) opens the private members so
 * that the byte streams the reference never returns (compression.cpp:27-64) can be dumped as golden vectors, and
 * (c) maps C++ exceptions to integer codes. The function set mirrors oracle/coder_oracle.c one-to-one so the same
 * Python test drives both.
 */
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
/* every std header the reference headers pull in, BEFORE the keyword macros below (include guards then skip them) */
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <vector>

#define private public
#define class struct
#include "Bitstream.h"
#include "BinaryArithmeticCoder.h"
#include "LosslessCoder.h"
#undef class
#undef private
#include "compression.h"

extern "C" {

uint8_t ref_count_nb_bits(uint32_t input) { return count_nb_bits(input); }

void* ref_new(uint32_t required_size_in_bits, uint8_t L, const double* probabilities) {
    return new (std::nothrow) LosslessCoder(required_size_in_bits, L, probabilities);
}
void ref_free(void* c) { delete static_cast<LosslessCoder*>(c); }
#define LC(c) (static_cast<LosslessCoder*>(c))
uint32_t ref_occupancy_in_bits_bac(void* c) { return LC(c)->occupancy_in_bits_bac(); }
uint32_t ref_occupancy_in_bits_bypass(void* c) { return LC(c)->occupancy_in_bits_bypass(); }
uint32_t ref_written_bits_bac(void* c) { return LC(c)->m_bac.m_bitstream.m_write_index; }
uint32_t ref_written_bits_bypass(void* c) { return LC(c)->m_bitstream_bypass.m_write_index; }
const uint8_t* ref_bytes_bac(void* c) { return LC(c)->m_bac.m_bitstream.m_data.data(); }
const uint8_t* ref_bytes_bypass(void* c) { return LC(c)->m_bitstream_bypass.m_data.data(); }

/* std::out_of_range from m_probabilities.at(i) -> 5 (same convention as coder_oracle.c) */
#define GUARD(expr) try { return (int)(expr); } catch (const std::out_of_range&) { return 5; }
int ref_write_sign(void* c, int16_t v) { GUARD(LC(c)->write_sign(v)) }
int ref_read_sign(void* c, int16_t* v) { GUARD(LC(c)->read_sign(*v)) }
int ref_write_eg0(void* c, uint16_t v) { GUARD(LC(c)->write_eg0(v)) }
int ref_read_eg0(void* c, uint16_t* v) { GUARD(LC(c)->read_eg0(*v)) }
int ref_write_truncated_unary(void* c, uint16_t v) { GUARD(LC(c)->write_truncated_unary(v)) }
int ref_read_truncated_unary(void* c, uint16_t* v) { GUARD(LC(c)->read_truncated_unary(*v)) }
int ref_write_signed_ueg0(void* c, int16_t v) { GUARD(LC(c)->write_signed_ueg0(v)) }
int ref_read_signed_ueg0(void* c, int16_t* v) { GUARD(LC(c)->read_signed_ueg0(*v)) }
int ref_stop_bac_encoding(void* c) { GUARD(LC(c)->stop_bac_encoding()) }
int ref_start_bac_decoding(void* c) { GUARD(LC(c)->start_bac_decoding()) }
int ref_bac_encoding(void* c, uint8_t bit, double p) { GUARD(LC(c)->m_bac.encoding(bit, p)) }
int ref_bac_decoding(void* c, uint8_t* storage, double p) { GUARD(LC(c)->m_bac.decoding(*storage, p)) }

/* The reference entry point itself (compression.cpp:3-65). Returns 0 and *nb_bits on success; on exception returns
 * -1 (invalid_argument), 5 (out_of_range) or the "Error of type N" code, and copies what() into msg. */
int ref_compress_lossless(uint32_t size, const int16_t* in, int16_t* out, uint8_t L, const double* probs,
                          uint32_t* nb_bits, char* msg, uint32_t msg_cap) {
    try {
        *nb_bits = compress_lossless(size, in, out, L, probs);
        return 0;
    } catch (const std::invalid_argument& e) {
        if (msg) { strncpy(msg, e.what(), msg_cap - 1); msg[msg_cap - 1] = 0; }
        return -1;
    } catch (const std::out_of_range& e) {
        if (msg) { strncpy(msg, e.what(), msg_cap - 1); msg[msg_cap - 1] = 0; }
        return 5;
    } catch (const std::runtime_error& e) {
        if (msg) { strncpy(msg, e.what(), msg_cap - 1); msg[msg_cap - 1] = 0; }
        const char* p = strstr(e.what(), "Error of type ");
        return p ? atoi(p + 14) : -3;
    }
}

}  // extern "C"
