/*
 * eae_coder.h -- C ABI of the host-side lossless coder (libeae_coder.so).
 *
 * Drop-in boundary for the reference's only native FFI on the hot path:
 *
 *   kodak_tensorflow/lossless/interface_cython.pyx:6-11
 *       cdef extern from "c++/source/compression.h":
 *           numpy.uint32_t compress_lossless(const uint32_t&, const int16_t* const, int16_t* const,
 *                                            const uint8_t&, const double* const) except +
 *   kodak_tensorflow/lossless/c++/source/compression.h:41-45   (C++ linkage, throws)
 *
 * Everything here is extern "C", plain pointers and sizes, no exceptions across the ABI: functions return an
 * eae_error_code; where the reference throws std::runtime_error("Error of type N <stage>") the stage is reported
 * through an out-parameter so a binding can rebuild the exact message.
 *
 * Bit-exactness contract (tests/test_coder_host.py): for every input, the bit counts, the two byte streams
 * (LSB-first packing, Bitstream.cpp:30-79) and the decoded symbols are identical to the reference's.
 * Threading: every function is re-entrant; the *_maps entry points fan independent maps out over an internal
 * thread pool (the reference is single-threaded; maps are independent streams, compression.py:67-81).
 */
#ifndef EAE_CODER_H
#define EAE_CODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* utils.h:12-19 (same numeric values) + two ABI-level codes for what the reference raises as C++ exceptions. */
typedef enum {
    EAE_SUCCESS = 0,
    EAE_CAPACITY_ERROR = 1,    /* bitstream full                       (Bitstream.cpp:32-35) */
    EAE_RESOURCE_ERROR = 2,    /* bypass stream under-run              (Bitstream.cpp:63-66) */
    EAE_PRECISION_ERROR = 3,   /* low/high left the 16-bit range       (BinaryArithmeticCoder.cpp:184-187) */
    EAE_PROBABILITY_ERROR = 4, /* p is NaN or outside ]0,1[            (BinaryArithmeticCoder.cpp:146-153) */
    EAE_OUT_OF_RANGE = 5,      /* std::out_of_range: m_probabilities.at(i) with L == 0 (LosslessCoder.cpp:173,189,208) */
    EAE_ROUNDTRIP_MISMATCH = 6,/* decode(encode(x)) != x in EAE_MODE_ROUNDTRIP_VERIFY: the AssertionError of
                                  lossless/compression.py:146-153 (cannot happen unless the coder is broken) */
    EAE_NULL_POINTER = -1,     /* std::invalid_argument                (compression.cpp:9-12) */
    EAE_BAD_ALLOC = -2
} eae_error_code;

/* Where compress_lossless failed -> the tail of the reference's runtime_error message (compression.cpp:32-62). */
typedef enum {
    EAE_STAGE_NONE = 0,
    EAE_STAGE_ENCODING = 1,       /* "during the encoding." */
    EAE_STAGE_STOP_ENCODING = 2,  /* "when stopping the binary arithmetic encoding." */
    EAE_STAGE_START_DECODING = 3, /* "when starting the binary arithmetic decoding." */
    EAE_STAGE_DECODING = 4        /* "during the decoding." */
} eae_stage;

const char* eae_coder_version(void);

/* ---- (1) reference-equivalent entry point ------------------------------------------------------------------------
 * Replaces compress_lossless (compression.cpp:3-65): encodes all symbols, flushes, counts bits, decodes all symbols.
 * *nb_bits = BAC bits + bypass bits. Internal stream capacity is size*max(32,L) bits each, as in the reference
 * (compression.cpp:24), so capacity errors occur for exactly the same inputs. */
int eae_coder_compress_lossless(uint32_t size, const int16_t* array_input, int16_t* array_output,
                                uint8_t truncated_unary_length, const double* probabilities,
                                uint32_t* nb_bits, int* stage);

/* ---- (2) split encode / decode: the streams the reference never returns (compression.cpp:27-64) ------------------
 * eae_coder_stream_capacity_bytes: bytes the caller must provide per stream = ceil(size*max(32,L)/8).
 * encode: writes the BAC stream (after stop_encoding) and the bypass stream; *_bits = number of valid bits.
 * decode: inverse; needs both streams and their bit lengths; reads exactly ceil(bits/8) bytes of each stream. */
uint32_t eae_coder_stream_capacity_bytes(uint32_t size, uint8_t truncated_unary_length);
int eae_coder_encode(uint32_t size, const int16_t* array_input, uint8_t truncated_unary_length,
                     const double* probabilities,
                     uint8_t* bac_bytes, uint32_t* bac_bits, uint8_t* bypass_bytes, uint32_t* bypass_bits, int* stage);
int eae_coder_decode(uint32_t size, int16_t* array_output, uint8_t truncated_unary_length,
                     const double* probabilities,
                     const uint8_t* bac_bytes, uint32_t bac_bits, const uint8_t* bypass_bytes, uint32_t bypass_bits,
                     int* stage);

/* ---- (3) batched, threaded: all maps of a batch of images after ONE device->host copy -----------------------------
 * symbols: n_maps contiguous maps of map_size int16 each (channel-major / planar: map m = symbols + m*map_size; this
 *          is `ref_int16[:, :, i].flatten()` of compression.py:77 for every (image, i), laid out back to back).
 * prob_row[m]: row of `probabilities` ([n_rows][L], row-major float64 -- binary_probabilities[i, :] of
 *          compression.py:79) used for map m, or -1 to skip the map (the exception map, compression.py:68-75, whose
 *          cost is computed by the caller from its histogram); skipped maps get nb_bits[m] = 0, status[m] = 0 and,
 *          when `reconstruction` is given, a verbatim copy.
 * mode:    EAE_MODE_ROUNDTRIP = encode + decode + write `reconstruction` (what compress_lossless does per map);
 *          EAE_MODE_ENCODE_ONLY = encode and count bits only (`reconstruction` may be NULL);
 *          EAE_MODE_ROUNDTRIP_VERIFY = encode + decode + compare with the input inside the worker threads
 *          (`reconstruction` may be NULL; status EAE_ROUNDTRIP_MISMATCH on a difference).
 * nb_bits[m], status[m], stage[m]: per-map results. Return value: 0, or the first non-zero status encountered.
 * n_threads <= 0 -> every CPU the process may use (affinity mask capped by the cgroup CPU quota; capped at n_maps). */
enum { EAE_MODE_ROUNDTRIP = 0, EAE_MODE_ENCODE_ONLY = 1, EAE_MODE_ROUNDTRIP_VERIFY = 2 };
int eae_coder_compress_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, int16_t* reconstruction,
                            uint8_t truncated_unary_length, const double* probabilities, const int32_t* prob_row,
                            uint32_t* nb_bits, int32_t* status, int32_t* stage, int mode, int n_threads);

/* Same, but keeps the streams. Map m owns the region [m*stream_stride_bytes, (m+1)*stream_stride_bytes) of `streams`:
 * its BAC bytes start at the beginning of the region, its bypass bytes at +stream_stride_bytes/2; bac_bits[m] and
 * bypass_bits[m] give the valid lengths. stream_stride_bytes/2 must be >= eae_coder_stream_capacity_bytes(map_size, L)
 * + 16 (whole-word stores), else EAE_CAPACITY_ERROR is returned before any work. Skipped maps (prob_row < 0) get 0 bits.
 * decode_maps is the inverse over the same layout. */
int eae_coder_encode_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols,
                          uint8_t truncated_unary_length, const double* probabilities, const int32_t* prob_row,
                          uint8_t* streams, uint64_t stream_stride_bytes,
                          uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status, int32_t* stage, int n_threads);
int eae_coder_decode_maps(uint32_t n_maps, uint32_t map_size, int16_t* symbols_out,
                          uint8_t truncated_unary_length, const double* probabilities, const int32_t* prob_row,
                          const uint8_t* streams, uint64_t stream_stride_bytes,
                          const uint32_t* bac_bits, const uint32_t* bypass_bits, int32_t* status, int32_t* stage,
                          int n_threads);

/* ---- (4) the LosslessCoder object, method by method (LosslessCoder.h:12-169) --------------------------------------
 * Needed so the reference's own unit cases (tests.cpp:134-352: sign, EG0, truncated unary, signed UEG0, raw BAC) can
 * be replayed against this library. */
typedef struct eae_lossless_coder eae_lossless_coder;
eae_lossless_coder* eae_lossless_coder_new(uint32_t required_size_in_bits, uint8_t truncated_unary_length,
                                           const double* probabilities);
void eae_lossless_coder_free(eae_lossless_coder* c);
uint32_t eae_lossless_coder_occupancy_in_bits_bac(const eae_lossless_coder* c);
uint32_t eae_lossless_coder_occupancy_in_bits_bypass(const eae_lossless_coder* c);
uint32_t eae_lossless_coder_written_bits_bac(const eae_lossless_coder* c);
uint32_t eae_lossless_coder_written_bits_bypass(const eae_lossless_coder* c);
/* copies ceil(written_bits/8) bytes; returns that count */
uint32_t eae_lossless_coder_copy_bac(const eae_lossless_coder* c, uint8_t* dst, uint32_t dst_cap);
uint32_t eae_lossless_coder_copy_bypass(const eae_lossless_coder* c, uint8_t* dst, uint32_t dst_cap);
int eae_lossless_coder_write_sign(eae_lossless_coder* c, int16_t input);
int eae_lossless_coder_read_sign(eae_lossless_coder* c, int16_t* output);
int eae_lossless_coder_write_eg0(eae_lossless_coder* c, uint16_t input);
int eae_lossless_coder_read_eg0(eae_lossless_coder* c, uint16_t* output);
int eae_lossless_coder_write_truncated_unary(eae_lossless_coder* c, uint16_t input);
int eae_lossless_coder_read_truncated_unary(eae_lossless_coder* c, uint16_t* output);
int eae_lossless_coder_write_signed_ueg0(eae_lossless_coder* c, int16_t input);
int eae_lossless_coder_read_signed_ueg0(eae_lossless_coder* c, int16_t* output);
int eae_lossless_coder_stop_bac_encoding(eae_lossless_coder* c);
int eae_lossless_coder_start_bac_decoding(eae_lossless_coder* c);
int eae_lossless_coder_bac_encoding(eae_lossless_coder* c, uint8_t input, double probability);
int eae_lossless_coder_bac_decoding(eae_lossless_coder* c, uint8_t* storage, double probability);

/* utils.cpp:13-28 (count_nb_bits) -- exported for the exhaustive check against the double-log2 original. */
uint8_t eae_coder_count_nb_bits(uint32_t input);

/* ---- (5) host-side statistics that feed the coder (lossless/stats.py:136-195) -------------------------------------
 * Per map: from int16 symbols (already centred-quantised and divided by the bin width), accumulate the truncated-
 * unary decision counts: for |s| < L: ones[0:|s|] += 1, zeros[|s|] += 1; else ones[:] += 1.
 * zeros/ones: [n_maps][L] int64, ACCUMULATED into (caller zeroes). */
int eae_coder_count_binary_decisions(uint32_t n_maps, uint32_t map_size, const int16_t* symbols,
                                     uint8_t truncated_unary_length, int64_t* zeros, int64_t* ones, int n_threads);

/* ---- (6) row sums in numpy's order (the approximate rate, tools/tools.py:523-537, 977-989) ------------------------------
 * `discrete_entropy` ends in `-numpy.sum(frequency*numpy.log2(frequency))`, once per feature map: 128 calls per image on the
 * reference's side, each over the few dozen non-empty bins of one map. numpy adds a contiguous float64 array pairwise (blocks
 * of <= 128 elements through eight running sums, longer arrays halved at multiples of eight), and the result's last bits
 * depend on that order. This is the same order for `rows` independent runs of one array: row i = values[bounds[i] ..
 * bounds[i + 1]), sums[i] = what `numpy.sum` returns for that slice. kodak/tools/tools.py checks it against `numpy.sum` itself
 * when it is first used and keeps calling `numpy.sum` if the installed numpy adds in another order. EAE_NULL_POINTER; EAE_OUT_OF_RANGE
 * for descending bounds. */
int eae_coder_pairwise_row_sums(const double* values, const int64_t* bounds, int64_t rows, double* sums);

/* ---- checkpoint ingestion helper ------------------------------------------------------------------------------------
 * CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of `size` bytes, continuing from `crc` (0 to start). The
 * reference restores its models with tf.train.Saver (eae/graph/EntropyAutoencoder.py:454-458); TensorFlow's bundle
 * and table formats checksum every block and tensor with this CRC (kodak/eae/graph/tf_checkpoint.py is the caller). */
uint32_t eae_crc32c(const void* data, size_t size, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* EAE_CODER_H */
