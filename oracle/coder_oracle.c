/*
 * oracle/coder_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or called from the product path).
 *
 * Plain-C, bit-serial restatement of the reference lossless coder
 *   kodak_tensorflow/lossless/c++/source/{utils,Bitstream,BinaryArithmeticCoder,LosslessCoder,compression}.cpp
 * Each function cites the reference lines it follows (paths relative to that directory).
 *
 * Parity pin: tests/test_oracle_coder.py checks this file against
 *   (1) the reference's own known-answer cases (tests.cpp:69-376, test_lossless.py:96-101), and
 *   (2) byte streams dumped from the REAL reference classes compiled into oracle/_ref (see oracle/ref_shim.cpp),
 *       committed as tests/golden/coder_golden.npz by oracle/gen_golden.py.
 *
 * Deliberately slow and literal: one bit at a time, same state machine, same error codes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utils.h:12-19 */
enum { ORC_SUCCESS = 0, ORC_CAPACITY = 1, ORC_RESOURCE = 2, ORC_PRECISION = 3, ORC_PROBABILITY = 4,
       ORC_OUT_OF_RANGE = 5 /* std::out_of_range from m_probabilities.at(i), LosslessCoder.cpp:173,189,208 */ };

/* BinaryArithmeticCoder.cpp:3-33 */
#define PRECISION 16u
#define RANGE_MAX ((((uint32_t)1) << PRECISION) - 1u)      /* 0xFFFF */
#define RANGE_HALF (RANGE_MAX >> 1)                         /* 0x7FFF */
#define RANGE_QUARTER (RANGE_HALF >> 1)                     /* 0x3FFF */
#define RANGE_THREE_QUARTERS (3u * RANGE_QUARTER)           /* 49149, NOT 0xBFFF */
#define MASK (((uint32_t)1) << (PRECISION - 1u))            /* 0x8000 */

/* utils.cpp:3-11 */
static uint32_t create_divisible(uint32_t input, uint32_t divisor) {
    const uint32_t remainder = input % divisor;
    if (remainder) input += divisor - remainder;
    return input;
}

/* utils.cpp:13-28 : floor(log2((double)x)) + 1, and 1 for x == 0 */
uint8_t orc_count_nb_bits(uint32_t input) {
    if (input) return (uint8_t)(floor(log2((double)input)) + 1);
    return 1;
}

/* Bitstream.h:11-64 */
typedef struct {
    uint32_t read_index, write_index;
    uint32_t nb_bytes;
    uint8_t* data;
} bitstream_t;

/* Bitstream.cpp:3-7 */
static int bs_init(bitstream_t* b, uint32_t required_size_in_bits) {
    b->read_index = 0;
    b->write_index = 0;
    b->nb_bytes = create_divisible(required_size_in_bits, 8) >> 3;
    b->data = (uint8_t*)calloc(b->nb_bytes ? b->nb_bytes : 1, 1);
    return b->data != NULL;
}
/* Bitstream.cpp:9-18 */
static uint32_t bs_size_in_bits(const bitstream_t* b) { return b->nb_bytes << 3; }
/* Bitstream.cpp:20-23 */
static uint32_t bs_occupancy(const bitstream_t* b) { return b->write_index - b->read_index; }
/* Bitstream.cpp:25-28 */
static int bs_entirely_read(const bitstream_t* b) { return b->write_index == b->read_index; }
/* Bitstream.cpp:30-59 : bit i -> byte i>>3, bit position i%8 (LSB first) */
static int bs_write_bit(bitstream_t* b, uint8_t byte_storing_bit) {
    if (b->write_index + 1 > bs_size_in_bits(b)) return ORC_CAPACITY;
    const uint32_t destination_byte = b->write_index >> 3;
    const uint8_t destination_bit = (uint8_t)(b->write_index % 8);
    b->data[destination_byte] &= (uint8_t)~(0x1 << destination_bit);
    b->data[destination_byte] |= (uint8_t)((byte_storing_bit & 0x1) << destination_bit);
    b->write_index++;
    return ORC_SUCCESS;
}
/* Bitstream.cpp:61-79 */
static int bs_read_bit(bitstream_t* b, uint8_t* storage) {
    if (b->read_index >= b->write_index) return ORC_RESOURCE;
    const uint32_t source_byte = b->read_index >> 3;
    const uint8_t source_bit = (uint8_t)(b->read_index % 8);
    *storage &= 0xFE;
    *storage |= (uint8_t)((b->data[source_byte] >> source_bit) & 0x1);
    b->read_index++;
    return ORC_SUCCESS;
}

/* BinaryArithmeticCoder.h:9-117 */
typedef struct {
    bitstream_t bitstream;
    uint32_t low, middle, high, nb_e3, code;
} bac_t;

/* LosslessCoder.h:12-169 */
typedef struct orc_coder {
    bac_t bac;
    bitstream_t bypass;
    uint8_t truncated_unary_length;
    double probabilities[256];
} orc_coder_t;

/* BinaryArithmeticCoder.cpp:35-42 */
static int bac_init(bac_t* c, uint32_t required_size_in_bits) {
    c->low = 0; c->middle = RANGE_HALF; c->high = RANGE_MAX; c->nb_e3 = 0; c->code = 0;
    return bs_init(&c->bitstream, required_size_in_bits);
}
/* BinaryArithmeticCoder.cpp:136-142 (m_code is NOT reset) */
static void bac_reset(bac_t* c) { c->low = 0; c->middle = RANGE_HALF; c->high = RANGE_MAX; c->nb_e3 = 0; }

/* BinaryArithmeticCoder.cpp:144-156 */
static int bac_update_middle(bac_t* c, double probability) {
    if (isnan(probability)) return ORC_PROBABILITY;
    if ((probability <= 0.) || (probability >= 1.)) return ORC_PROBABILITY;
    c->middle = c->low + (uint32_t)(floor(probability * (c->high - c->low)));
    return ORC_SUCCESS;
}
/* BinaryArithmeticCoder.cpp:322-337 */
static int bac_clear_e3_queue(bac_t* c, uint8_t value) {
    const uint8_t inverted_bit = !(value & 0x1);
    for (uint32_t i = 0; i < c->nb_e3; i++) {
        int state = bs_write_bit(&c->bitstream, inverted_bit);
        if (state) return state;
    }
    c->nb_e3 = 0;
    return ORC_SUCCESS;
}
/* BinaryArithmeticCoder.cpp:158-180 */
static int bac_encode_bit(bac_t* c, uint8_t input, double probability) {
    int state = bac_update_middle(c, probability);
    if (state) return state;
    if (input & 0x1) c->low = c->middle + 1; else c->high = c->middle;
    return ORC_SUCCESS;
}
/* BinaryArithmeticCoder.cpp:182-252 */
static int bac_rescale_encoding(bac_t* c) {
    if (c->high > RANGE_MAX || c->low > RANGE_MAX) return ORC_PRECISION;
    int state = ORC_SUCCESS;
    uint32_t masked_high = 0;
    while (1) {
        masked_high = c->high & MASK;
        if (masked_high == (c->low & MASK)) {
            if (masked_high != 0x0) { c->high -= (RANGE_HALF + 1); c->low -= (RANGE_HALF + 1); }
            c->high <<= 0x1; c->high |= 0x1; c->low <<= 0x1;
            uint8_t value = (uint8_t)(masked_high >> (PRECISION - 1));
            state = bs_write_bit(&c->bitstream, value);
            if (state) return state;
            state = bac_clear_e3_queue(c, value);
            if (state) return state;
        } else if (c->low > RANGE_QUARTER && c->high <= RANGE_THREE_QUARTERS) {
            c->high -= (RANGE_QUARTER + 1); c->low -= (RANGE_QUARTER + 1);
            c->high <<= 0x1; c->high |= 0x1; c->low <<= 0x1;
            c->nb_e3 += 1;
        } else {
            break;
        }
    }
    return state;
}
/* BinaryArithmeticCoder.cpp:49-59 */
static int bac_encoding(bac_t* c, uint8_t input, double probability) {
    int state = bac_encode_bit(c, input, probability);
    if (state) return state;
    return bac_rescale_encoding(c);
}
/* BinaryArithmeticCoder.cpp:61-102 */
static int bac_stop_encoding(bac_t* c) {
    int state;
    c->nb_e3++;
    if (c->low < RANGE_QUARTER) {
        state = bs_write_bit(&c->bitstream, 0);
        if (state) return state;
        state = bac_clear_e3_queue(c, 0);
        if (state) return state;
    } else {
        state = bs_write_bit(&c->bitstream, 1);
        if (state) return state;
        state = bac_clear_e3_queue(c, 1);
        if (state) return state;
    }
    bac_reset(c);
    return state;
}
/* BinaryArithmeticCoder.cpp:104-122 : `storage` is sticky once the stream is exhausted */
static int bac_start_decoding(bac_t* c) {
    int state = ORC_SUCCESS;
    uint8_t storage = 0;
    for (uint32_t i = 0; i < PRECISION; i++) {
        if (!bs_entirely_read(&c->bitstream)) {
            state = bs_read_bit(&c->bitstream, &storage);
            if (state) return state;
        }
        c->code <<= 0x1;
        c->code |= storage;
    }
    return state;
}
/* BinaryArithmeticCoder.cpp:254-273 : `storage` untouched when code is outside [low, high] */
static int bac_decode_bit(bac_t* c, uint8_t* storage, double probability) {
    int state = bac_update_middle(c, probability);
    if (state) return state;
    if (c->code >= c->low && c->code <= c->middle) { c->high = c->middle; *storage = 0; }
    else if (c->code > c->middle && c->code <= c->high) { c->low = c->middle + 1; *storage = 1; }
    return state;
}
/* BinaryArithmeticCoder.cpp:275-320 */
static int bac_rescale_decoding(bac_t* c) {
    int state = ORC_SUCCESS;
    uint8_t storage = 0;
    while (1) {
        if (c->high <= RANGE_HALF) {
        } else if (c->low > RANGE_HALF) {
            c->high -= (RANGE_HALF + 1); c->low -= (RANGE_HALF + 1); c->code -= (RANGE_HALF + 1);
        } else if (c->high <= RANGE_THREE_QUARTERS && c->low > RANGE_QUARTER) {
            c->high -= (RANGE_QUARTER + 1); c->low -= (RANGE_QUARTER + 1); c->code -= (RANGE_QUARTER + 1);
        } else {
            break;
        }
        if (!bs_entirely_read(&c->bitstream)) {
            state = bs_read_bit(&c->bitstream, &storage);
            if (state) return state;
        }
        c->high = ((c->high << 0x1) & RANGE_MAX) | 0x1;
        c->low = ((c->low << 0x1) & RANGE_MAX) | 0x0;
        c->code = ((c->code << 0x1) & RANGE_MAX) | storage;
    }
    return state;
}
/* BinaryArithmeticCoder.cpp:124-134 */
static int bac_decoding(bac_t* c, uint8_t* storage, double probability) {
    int state = bac_decode_bit(c, storage, probability);
    if (state) return state;
    return bac_rescale_decoding(c);
}

/* ------------------------------------------------------------------ LosslessCoder ---- */

/* LosslessCoder.cpp:3-10 */
orc_coder_t* orc_new(uint32_t required_size_in_bits, uint8_t truncated_unary_length, const double* probabilities) {
    orc_coder_t* c = (orc_coder_t*)calloc(1, sizeof(orc_coder_t));
    if (!c) return NULL;
    if (!bac_init(&c->bac, required_size_in_bits) || !bs_init(&c->bypass, required_size_in_bits)) return NULL;
    c->truncated_unary_length = truncated_unary_length;
    for (unsigned i = 0; i < truncated_unary_length; i++) c->probabilities[i] = probabilities[i];
    return c;
}
void orc_free(orc_coder_t* c) {
    if (!c) return;
    free(c->bac.bitstream.data);
    free(c->bypass.data);
    free(c);
}
/* LosslessCoder.cpp:12-20 */
uint32_t orc_occupancy_in_bits_bac(const orc_coder_t* c) { return bs_occupancy(&c->bac.bitstream); }
uint32_t orc_occupancy_in_bits_bypass(const orc_coder_t* c) { return bs_occupancy(&c->bypass); }
/* test hooks: raw stream contents (the reference keeps these private; oracle/ref_shim.cpp exposes the same view) */
uint32_t orc_written_bits_bac(const orc_coder_t* c) { return c->bac.bitstream.write_index; }
uint32_t orc_written_bits_bypass(const orc_coder_t* c) { return c->bypass.write_index; }
const uint8_t* orc_bytes_bac(const orc_coder_t* c) { return c->bac.bitstream.data; }
const uint8_t* orc_bytes_bypass(const orc_coder_t* c) { return c->bypass.data; }

/* LosslessCoder.cpp:22-37 */
int orc_write_sign(orc_coder_t* c, int16_t input) {
    int state = ORC_SUCCESS;
    if (input) {
        if (input < 0) state = bs_write_bit(&c->bypass, 0);
        else state = bs_write_bit(&c->bypass, 1);
    }
    return state;
}
/* LosslessCoder.cpp:39-56 */
int orc_read_sign(orc_coder_t* c, int16_t* output) {
    int state = ORC_SUCCESS;
    if (*output) {
        uint8_t storage = 0;
        state = bs_read_bit(&c->bypass, &storage);
        if (state) return state;
        if (!storage) *output = (int16_t)(*output * -1);
    }
    return state;
}
/* LosslessCoder.cpp:58-111 */
int orc_write_eg0(orc_coder_t* c, uint16_t input) {
    int state = ORC_SUCCESS;
    const uint32_t input_plus_1 = (uint32_t)input + 1;
    const uint8_t nb_bits_minus_1 = (uint8_t)(orc_count_nb_bits(input_plus_1) - 1);
    for (uint8_t i = 0; i < nb_bits_minus_1; i++) {
        state = bs_write_bit(&c->bypass, 1);
        if (state) return state;
    }
    state = bs_write_bit(&c->bypass, 0);
    if (state) return state;
    const uint16_t suffix = (uint16_t)(input_plus_1 - (1u << nb_bits_minus_1));
    uint8_t bit_isolation = 0;
    for (uint8_t i = 0; i < nb_bits_minus_1; i++) {
        bit_isolation = (uint8_t)(suffix >> (nb_bits_minus_1 - i - 1) & 0x1);
        state = bs_write_bit(&c->bypass, bit_isolation);
        if (state) return state;
    }
    return state;
}
/* LosslessCoder.cpp:113-165 */
int orc_read_eg0(orc_coder_t* c, uint16_t* output) {
    int state = ORC_SUCCESS;
    uint8_t storage = 0;
    uint8_t nb_bits_minus_1 = 0;
    while (1) {
        state = bs_read_bit(&c->bypass, &storage);
        if (state) return state;
        if (!storage) break;
        else nb_bits_minus_1++;
    }
    *output = 0;
    for (uint8_t i = 0; i < nb_bits_minus_1; i++) {
        *output = (uint16_t)(*output << 1);
        state = bs_read_bit(&c->bypass, &storage);
        if (state) return state;
        *output |= storage & 0x1;
    }
    *output = (uint16_t)(*output + ((1 << nb_bits_minus_1) - 1));
    return state;
}
/* LosslessCoder.cpp:167-191 ; `.at(i)` -> ORC_OUT_OF_RANGE when i >= L (only reachable with L == 0) */
int orc_write_truncated_unary(orc_coder_t* c, uint16_t input) {
    int state = ORC_SUCCESS;
    uint8_t i = 0;
    for (i = 0; i < input; i++) {
        if (i >= c->truncated_unary_length) return ORC_OUT_OF_RANGE;
        state = bac_encoding(&c->bac, 1, c->probabilities[i]);
        if (state) return state;
        if (i == c->truncated_unary_length - 1) return state;
    }
    if (i >= c->truncated_unary_length) return ORC_OUT_OF_RANGE;
    state = bac_encoding(&c->bac, 0, c->probabilities[i]);
    return state;
}
/* LosslessCoder.cpp:193-230 */
int orc_read_truncated_unary(orc_coder_t* c, uint16_t* output) {
    int state = ORC_SUCCESS;
    *output = 0;
    uint8_t i = 0;
    uint8_t storage = 0;
    while (1) {
        if (i >= c->truncated_unary_length) return ORC_OUT_OF_RANGE;
        state = bac_decoding(&c->bac, &storage, c->probabilities[i]);
        if (state) return state;
        if (!storage) break;
        else (*output)++;
        if (i == c->truncated_unary_length - 1) break;
        else i++;
    }
    return state;
}
/* LosslessCoder.cpp:232-252 */
int orc_write_signed_ueg0(orc_coder_t* c, int16_t input) {
    int state = ORC_SUCCESS;
    const uint16_t absolute_input = (uint16_t)abs((int)input);
    state = orc_write_truncated_unary(c, absolute_input);
    if (state) return state;
    if (absolute_input >= c->truncated_unary_length) {
        uint16_t difference = (uint16_t)(absolute_input - c->truncated_unary_length);
        state = orc_write_eg0(c, difference);
        if (state) return state;
    }
    return orc_write_sign(c, input);
}
/* LosslessCoder.cpp:254-276 */
int orc_read_signed_ueg0(orc_coder_t* c, int16_t* output) {
    int state = ORC_SUCCESS;
    uint16_t read_absolute_value = 0;
    state = orc_read_truncated_unary(c, &read_absolute_value);
    if (state) return state;
    if (read_absolute_value == c->truncated_unary_length) {
        uint16_t difference = 0;
        state = orc_read_eg0(c, &difference);
        if (state) return state;
        read_absolute_value = (uint16_t)(read_absolute_value + difference);
    }
    *output = (int16_t)read_absolute_value;
    return orc_read_sign(c, output);
}
/* LosslessCoder.cpp:278-286 */
int orc_stop_bac_encoding(orc_coder_t* c) { return bac_stop_encoding(&c->bac); }
int orc_start_bac_decoding(orc_coder_t* c) { return bac_start_decoding(&c->bac); }
/* raw BAC access used by tests.cpp:69-132 (BinaryArithmeticCoder::encoding / ::decoding) */
int orc_bac_encoding(orc_coder_t* c, uint8_t input, double probability) { return bac_encoding(&c->bac, input, probability); }
int orc_bac_decoding(orc_coder_t* c, uint8_t* storage, double probability) { return bac_decoding(&c->bac, storage, probability); }

/* compression.cpp:3-65.
 * Returns 0 on success. On failure returns the error_code (1..5) and sets *stage:
 *   0 = NULL pointer (std::invalid_argument, compression.cpp:9-12; return value -1)
 *   1 = "during the encoding."                         (compression.cpp:32-35)
 *   2 = "when stopping the binary arithmetic encoding." (:38-41)
 *   3 = "when starting the binary arithmetic decoding." (:52-55)
 *   4 = "during the decoding."                          (:58-62)
 * Optionally copies out the two byte streams as they stand after stop_encoding (test hook).
 */
int orc_compress_lossless(uint32_t size, const int16_t* array_input, int16_t* array_output,
                          uint8_t truncated_unary_length, const double* probabilities,
                          uint32_t* nb_bits, int* stage,
                          uint8_t* bac_bytes_out, uint32_t* bac_bits_out,
                          uint8_t* bypass_bytes_out, uint32_t* bypass_bits_out) {
    *stage = 0;
    if (!array_input || !array_output || !probabilities) return -1;
    uint32_t tul = (uint32_t)truncated_unary_length;
    uint32_t required_size_in_bits = size * (32u > tul ? 32u : tul);
    orc_coder_t* c = orc_new(required_size_in_bits, truncated_unary_length, probabilities);
    if (!c) return -2;
    int state = ORC_SUCCESS;
    for (uint32_t i = 0; i < size; i++) {
        state = orc_write_signed_ueg0(c, array_input[i]);
        if (state) { *stage = 1; orc_free(c); return state; }
    }
    state = orc_stop_bac_encoding(c);
    if (state) { *stage = 2; orc_free(c); return state; }
    *nb_bits = orc_occupancy_in_bits_bac(c) + orc_occupancy_in_bits_bypass(c);
    if (bac_bits_out) *bac_bits_out = orc_written_bits_bac(c);
    if (bypass_bits_out) *bypass_bits_out = orc_written_bits_bypass(c);
    if (bac_bytes_out) memcpy(bac_bytes_out, orc_bytes_bac(c), (orc_written_bits_bac(c) + 7) >> 3);
    if (bypass_bytes_out) memcpy(bypass_bytes_out, orc_bytes_bypass(c), (orc_written_bits_bypass(c) + 7) >> 3);
    state = orc_start_bac_decoding(c);
    if (state) { *stage = 3; orc_free(c); return state; }
    for (uint32_t i = 0; i < size; i++) {
        state = orc_read_signed_ueg0(c, &array_output[i]);
        if (state) { *stage = 4; orc_free(c); return state; }
    }
    orc_free(c);
    return ORC_SUCCESS;
}
