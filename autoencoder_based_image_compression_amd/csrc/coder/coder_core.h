// coder_core.h -- ONE implementation of the lossless coder for both machines: the host library (eae_coder.cpp, g++) and
// the gfx950 kernel that codes one map per lane (hip/coder_device.hip, hipcc). Same source, IEEE double on both sides
// (-ffp-contract=off), so the two produce identical bits by construction; tests compare them anyway.
//
// What it replaces: kodak_tensorflow/lossless/c++/source/{Bitstream,BinaryArithmeticCoder,LosslessCoder,compression}.cpp
// (UEG0 binarisation -> 16-bit binary arithmetic coder with E1/E2/E3 rescaling + a bypass stream).
// How it differs (same bits, different machine):
//  * bit I/O is word-level: bits are gathered in a 64-bit little-endian accumulator and flushed 8 bytes at a time; the
//    reference writes one bit per call through std::vector::at (Bitstream.cpp:30-59). LSB-first packing is preserved
//    (stream bit i lives at byte i>>3, bit i&7);
//  * E1/E2 renormalisation is closed-form in the encoder AND the decoder: the number of shifts is the count of leading
//    equal bits of low/high (one clz); the reference loops bit by bit (BinaryArithmeticCoder.cpp:182-252, 275-320).
// What it keeps exactly: the interval arithmetic (double multiply + floor, BinaryArithmeticCoder.cpp:154), the
// non-standard RANGE_THREE_QUARTERS = 3*0x3FFF = 49149, the flush rule, the sticky-bit decoder priming, stream
// capacities and therefore every error condition and code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EAE_HD __host__ __device__ __forceinline__
#else
#define EAE_HD inline
#endif

namespace eae_core {

// error codes: keep in sync with include/eae_coder.h (utils.h:12-19 + ABI extras)
enum { OK = 0, CAPACITY = 1, RESOURCE = 2, PRECISION = 3, PROBABILITY = 4, OUT_OF_RANGE = 5, MISMATCH = 6 };
enum { STAGE_NONE = 0, STAGE_ENCODING = 1, STAGE_STOP = 2, STAGE_START = 3, STAGE_DECODING = 4 };

constexpr uint32_t kRangeMax = 0xFFFFu;            // BinaryArithmeticCoder.cpp:16
constexpr uint32_t kRangeHalf = 0x7FFFu;           // :22
constexpr uint32_t kRangeQuarter = 0x3FFFu;        // :28
constexpr uint32_t kRangeThreeQuarters = 49149u;   // :29  (3 * 0x3FFF, not 0xBFFF)

EAE_HD uint32_t round_up_to_byte(uint32_t bits) {  // utils.cpp:3-11 with divisor 8
    const uint32_t r = bits % 8u;
    return r ? bits + (8u - r) : bits;
}
EAE_HD uint32_t count_nb_bits(uint32_t x) {  // utils.cpp:13-28: floor(log2(x)) + 1, 1 for 0 -- integer form
    return x ? (uint32_t)(32 - __builtin_clz(x)) : 1u;
}
EAE_HD uint32_t rev16(uint32_t v) {  // reverse the low 16 bits
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v << 16);   // v_bfrev_b32
#endif
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
    return v;
}
EAE_HD uint32_t required_bits(uint32_t size, uint32_t L) {  // compression.cpp:24 (uint32 arithmetic, as there)
    return size * (L > 32u ? L : 32u);
}
EAE_HD void store64(uint8_t* p, uint64_t v) {   // p is 8-byte aligned by construction (word index * 8 from an aligned base)
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint64_t*>(p) = v;
#else
    __builtin_memcpy(p, &v, 8);
#endif
}

// 32 stream bits starting at byte `pos` (a multiple of 4), never touching bytes at or beyond `limit`.
EAE_HD uint32_t load32(const uint8_t* data, uint32_t pos, uint32_t limit) {
    if (pos + 4u <= limit) {
#if defined(__HIP_DEVICE_COMPILE__)
        return *reinterpret_cast<const uint32_t*>(data + pos);   // device streams are 8-byte aligned regions
#else
        uint32_t w;
        __builtin_memcpy(&w, data + pos, 4);
        return w;
#endif
    }
    uint32_t w = 0;
    for (uint32_t i = 0; i < 4u; i++)
        if (pos + i < limit) w |= (uint32_t)data[pos + i] << (8u * i);
    return w;
}

// ---------------------------------------------------------------------------------------------------------------
// Bitstream (Bitstream.h:11-64): same observable behaviour, word-level storage. `data` must be 8-byte aligned and
// hold size_bits/8 bytes plus 8 bytes of slack (whole-word stores).
// ---------------------------------------------------------------------------------------------------------------
struct Bitstream {
    uint8_t* data;
    uint32_t size_bits;    // size_in_bits(): byte-rounded capacity
    uint32_t write_index;
    uint32_t read_index;
    uint32_t limit_bytes;  // bytes that may be read from `data`
    uint64_t acc;          // bits [write_index & ~63, write_index) not yet stored
    uint64_t rwin;         // read window: bit 0 = stream bit read_index; rcount valid bits
    uint32_t rcount;       // (read_index + rcount) % 32 == 0 whenever rcount > 0

    EAE_HD void init_writer(uint8_t* buf, uint32_t required_size_in_bits) {
        size_bits = round_up_to_byte(required_size_in_bits);
        data = buf;
        write_index = read_index = 0;
        acc = 0;
        rwin = 0;
        rcount = 0;
        limit_bytes = (size_bits >> 3) + 8;
    }
    EAE_HD void init_reader(const uint8_t* buf, uint32_t nb_bits) {
        data = const_cast<uint8_t*>(buf);
        size_bits = round_up_to_byte(nb_bits);
        write_index = nb_bits;
        read_index = 0;
        acc = 0;
        rwin = 0;
        rcount = 0;
        limit_bytes = size_bits >> 3;          // exact: never read past the caller's stream
    }
    EAE_HD uint32_t occupancy() const { return write_index - read_index; }   // Bitstream.cpp:20-23
    EAE_HD bool entirely_read() const { return write_index == read_index; }  // Bitstream.cpp:25-28

    // Appends the n (<= 32) low bits of `bits`; bit 0 is the first bit in time. Equivalent to n write_bit calls
    // (Bitstream.cpp:30-59); fails with capacity_error iff one of them would.
    EAE_HD int put(uint32_t bits, uint32_t n) {
        if (write_index + n > size_bits) return CAPACITY;
        const uint32_t sh = write_index & 63u;
        acc |= (uint64_t)bits << sh;
        if (sh + n >= 64u) {
            store64(data + ((write_index >> 6) << 3), acc);
            acc = sh ? ((uint64_t)bits >> (64u - sh)) : 0;
        }
        write_index += n;
        return OK;
    }
    // n copies of `bit` (the pending E3 queue, BinaryArithmeticCoder.cpp:322-337); n is unbounded.
    EAE_HD int put_run(uint32_t bit, uint32_t n) {
        const uint32_t word = bit ? 0xFFFFFFFFu : 0u;
        while (n >= 32u) {
            int s = put(word, 32);
            if (s) return s;
            n -= 32u;
        }
        return n ? put(word & ((1u << n) - 1u), n) : (int)OK;
    }
    // Makes every written bit visible in `data` (partial last word included).
    EAE_HD void flush() {
        if (write_index & 63u) store64(data + ((write_index >> 6) << 3), acc);
    }
    // Forgets the read window (needed only when bits are written after reads have started: the object API of the ABI).
    EAE_HD void sync_reader() { rcount = 0; rwin = 0; }
    // Tops the read window up to more than 32 valid bits (precondition: rcount <= 32), 32 stream bits per load.
    EAE_HD void refill() {
        if (rcount == 0) {
            const uint32_t sh = read_index & 31u;
            rwin = (uint64_t)(load32(data, (read_index >> 5) << 2, limit_bytes) >> sh);
            rcount = 32u - sh;
        }
        if (rcount <= 32u) {
            rwin |= (uint64_t)load32(data, (read_index + rcount) >> 3, limit_bytes) << rcount;
            rcount += 32u;
        }
    }
    // Up to 16 stream bits starting at read_index; the first bit in time is the MOST significant bit of the result
    // (what `code = (code << 1) | bit` builds). Caller guarantees n <= occupancy() and flush() after the last put.
    EAE_HD uint32_t take_msb_first(uint32_t n) {
        if (rcount < n) refill();
        const uint32_t field = (uint32_t)rwin & ((1u << n) - 1u);   // bit j = j-th bit in time
        rwin >>= n;
        rcount -= n;
        read_index += n;
        return rev16(field) >> (16u - n);
    }
    // One bit, caller guarantees occupancy() > 0.
    EAE_HD uint32_t take_bit() {
        if (rcount == 0) refill();
        const uint32_t bit = (uint32_t)rwin & 1u;
        rwin >>= 1;
        rcount--;
        read_index++;
        return bit;
    }
    // Bitstream.cpp:61-79 ; caller guarantees flush() happened after the last put.
    EAE_HD int read_bit(uint8_t& storage) {
        if (read_index >= write_index) return RESOURCE;
        storage = (uint8_t)((storage & 0xFE) | take_bit());
        return OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// BinaryArithmeticCoder (BinaryArithmeticCoder.h:9-117)
// ---------------------------------------------------------------------------------------------------------------
struct Bac {
    Bitstream bs;
    uint32_t low, middle, high, nb_e3, code;

    EAE_HD void init() { low = 0; middle = kRangeHalf; high = kRangeMax; nb_e3 = 0; code = 0; }   // :35-42
    EAE_HD void reset() { low = 0; middle = kRangeHalf; high = kRangeMax; nb_e3 = 0; }  // :136-142 (code is kept)

    EAE_HD int update_middle(double p) {  // :144-156 ; !(p > 0 && p < 1) also catches NaN
        if (!(p > 0. && p < 1.)) return PROBABILITY;
        // floor of a non-negative double == truncation
        middle = low + (uint32_t)(p * (double)(high - low));
        return OK;
    }

    // encoding() = encode_bit + rescale_encoding (:49-59, :158-252)
    EAE_HD int encode(uint32_t bit, double p) {
        int s = update_middle(p);
        if (s) return s;
        if (bit & 1u) low = middle + 1u; else high = middle;
        if (high > kRangeMax || low > kRangeMax) return PRECISION;
        // E1/E2: as long as the top bits agree, shift them out. n = number of leading equal bits (0..16).
        const uint32_t diff = (low ^ high) & 0xFFFFu;
        const uint32_t n = diff ? (uint32_t)__builtin_clz(diff) - 16u : 16u;
        if (n) {
            const uint32_t out = rev16(high);  // bit k of `out` = k-th emitted bit (MSB of high first)
            if (nb_e3 == 0) {
                s = bs.put(out & ((1u << n) - 1u), n);
                if (s) return s;
            } else {
                const uint32_t first = out & 1u;
                s = bs.put(first, 1);
                if (s) return s;
                s = bs.put_run(first ^ 1u, nb_e3);  // clear_e3_queue (:322-337)
                if (s) return s;
                nb_e3 = 0;
                if (n > 1u) {
                    s = bs.put((out >> 1) & ((1u << (n - 1u)) - 1u), n - 1u);
                    if (s) return s;
                }
            }
            low = (low << n) & 0xFFFFu;
            high = ((high << n) & 0xFFFFu) | ((1u << n) - 1u);
        }
        // E3 (:238-245). Once the top bits differ they keep differing, so no E1/E2 can follow.
        while (low > kRangeQuarter && high <= kRangeThreeQuarters) {
            high = ((high - (kRangeQuarter + 1u)) << 1) | 1u;
            low = (low - (kRangeQuarter + 1u)) << 1;
            nb_e3++;
        }
        return OK;
    }

    EAE_HD int stop_encoding() {  // :61-102
        nb_e3++;
        const uint32_t b = (low < kRangeQuarter) ? 0u : 1u;
        int s = bs.put(b, 1);
        if (s) return s;
        s = bs.put_run(b ^ 1u, nb_e3);
        if (s) return s;
        nb_e3 = 0;
        reset();
        bs.flush();
        return OK;
    }

    EAE_HD int start_decoding() {  // :104-122 ; `storage` keeps its last value once the stream is exhausted
        uint8_t storage = 0;
        for (uint32_t i = 0; i < 16u; i++) {
            if (!bs.entirely_read()) {
                int s = bs.read_bit(storage);
                if (s) return s;
            }
            code = (code << 1) | storage;
        }
        return OK;
    }

    // decoding() = decode_bit + rescale_decoding (:124-134, :254-320). The renormalisation is closed-form like the
    // encoder's: E1/E2 fire exactly while the top bits of low and high agree (n = leading equal bits, one clz), then
    // only E3 can fire. E2's `code -= 0x8000` is absorbed by the 16-bit mask after the shift. Stream exhaustion keeps
    // the reference's semantics: `storage` starts at 0 in every rescale call and, once no bit is left, repeats the
    // last bit read IN THIS CALL (:275-277, :303-310).
    EAE_HD int decode(uint8_t& storage, double p) {
        int s = update_middle(p);
        if (s) return s;
        if (code >= low && code <= middle) { high = middle; storage = 0; }
        else if (code > middle && code <= high) { low = middle + 1u; storage = 1; }
        uint32_t sticky = 0;
        const uint32_t diff = (low ^ high) & 0xFFFFu;
        const uint32_t n = diff ? (uint32_t)__builtin_clz(diff) - 16u : 16u;
        if (n) {
            const uint32_t avail = bs.write_index - bs.read_index;
            const uint32_t k = n < avail ? n : avail;
            uint32_t bits = k ? bs.take_msb_first(k) : 0u;
            if (k) sticky = bits & 1u;
            if (k < n) bits = (bits << (n - k)) | (sticky ? ((1u << (n - k)) - 1u) : 0u);
            low = (low << n) & kRangeMax;
            high = ((high << n) & kRangeMax) | ((1u << n) - 1u);
            code = ((code << n) & kRangeMax) | bits;
        }
        while (high <= kRangeThreeQuarters && low > kRangeQuarter && high > kRangeHalf && low <= kRangeHalf) {
            high -= kRangeQuarter + 1u; low -= kRangeQuarter + 1u; code -= kRangeQuarter + 1u;
            if (bs.read_index < bs.write_index) sticky = bs.take_bit();
            high = ((high << 1) & kRangeMax) | 1u;
            low = (low << 1) & kRangeMax;
            code = ((code << 1) & kRangeMax) | sticky;
        }
        return OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// LosslessCoder (LosslessCoder.h:12-169). `probabilities` points at L doubles owned by the caller, `prob_stride` apart
// (1 on the host; the kernel interleaves the rows of the lanes of a block in LDS).
// ---------------------------------------------------------------------------------------------------------------
struct LosslessCoder {
    Bac bac;
    Bitstream bypass;
    uint32_t L;
    uint32_t prob_stride;
    const double* probabilities;
    EAE_HD double probability(uint32_t i) const { return probabilities[i * prob_stride]; }

    EAE_HD int write_sign(int16_t input) {  // LosslessCoder.cpp:22-37
        return input ? bypass.put(input < 0 ? 0u : 1u, 1) : (int)OK;
    }
    EAE_HD int read_sign(int16_t& output) {  // :39-56
        if (output) {
            uint8_t storage = 0;
            int s = bypass.read_bit(storage);
            if (s) return s;
            if (!storage) output = (int16_t)(-(int)output);
        }
        return OK;
    }
    EAE_HD int write_eg0(uint16_t input) {  // :58-111 : n ones, a zero, then the n low bits of input+1, MSB first
        const uint32_t v = (uint32_t)input + 1u;
        const uint32_t n = count_nb_bits(v) - 1u;  // 0..16
        int s = bypass.put((1u << n) - 1u, n + 1u);   // n ones then a zero, the zero last in time
        if (s) return s;
        if (n) {
            const uint32_t suffix = v - (1u << n);
            s = bypass.put(rev16(suffix) >> (16u - n), n);   // MSB of the suffix first in time
        }
        return s;
    }
    EAE_HD int read_eg0(uint16_t& output) {  // :113-165
        uint8_t storage = 0;
        uint8_t n = 0;
        for (;;) {
            int s = bypass.read_bit(storage);
            if (s) return s;
            if (!storage) break;
            n++;
        }
        output = 0;
        for (uint8_t i = 0; i < n; i++) {
            output = (uint16_t)(output << 1);
            int s = bypass.read_bit(storage);
            if (s) return s;
            output |= storage & 1u;
        }
        output = (uint16_t)(output + ((1 << n) - 1));
        return OK;
    }
    EAE_HD int write_truncated_unary(uint16_t input) {  // :167-191
        if (L == 0) return OUT_OF_RANGE;  // m_probabilities.at(0) throws whatever the input
        const uint32_t ones = input < L ? input : L;
        for (uint32_t i = 0; i < ones; i++) {
            int s = bac.encode(1u, probability(i));
            if (s) return s;
        }
        return input < L ? bac.encode(0u, probability(input)) : (int)OK;
    }
    EAE_HD int read_truncated_unary(uint16_t& output) {  // :193-230
        output = 0;
        if (L == 0) return OUT_OF_RANGE;
        uint32_t i = 0;
        uint8_t storage = 0;
        for (;;) {
            int s = bac.decode(storage, probability(i));
            if (s) return s;
            if (!storage) break;
            output++;
            if (i == L - 1u) break;
            i++;
        }
        return OK;
    }
    EAE_HD int write_signed_ueg0(int16_t input) {  // :232-252
        const int v = (int)input;
        const uint16_t a = (uint16_t)(v < 0 ? -v : v);
        int s = write_truncated_unary(a);
        if (s) return s;
        if (a >= L) {
            s = write_eg0((uint16_t)(a - L));
            if (s) return s;
        }
        return write_sign(input);
    }
    EAE_HD int read_signed_ueg0(int16_t& output) {  // :254-276
        uint16_t a = 0;
        int s = read_truncated_unary(a);
        if (s) return s;
        if (a == L) {
            uint16_t d = 0;
            s = read_eg0(d);
            if (s) return s;
            a = (uint16_t)(a + d);
        }
        output = (int16_t)a;
        return read_sign(output);
    }

    // compress_lossless, first half (compression.cpp:27-42): every symbol, then the flush.
    EAE_HD int encode_map(uint32_t size, const int16_t* in, int* stage) {
        for (uint32_t i = 0; i < size; i++) {
            int s = write_signed_ueg0(in[i]);
            if (s) { *stage = STAGE_ENCODING; return s; }
        }
        int s = bac.stop_encoding();
        if (s) { *stage = STAGE_STOP; return s; }
        bypass.flush();
        return OK;
    }
    // second half (compression.cpp:50-63): prime the decoder and read every symbol back.
    EAE_HD int decode_map(uint32_t size, int16_t* out, int* stage) {
        int s = bac.start_decoding();
        if (s) { *stage = STAGE_START; return s; }
        for (uint32_t i = 0; i < size; i++) {
            s = read_signed_ueg0(out[i]);
            if (s) { *stage = STAGE_DECODING; return s; }
        }
        return OK;
    }
    // same, comparing with the input instead of storing (the assert_equal of lossless/compression.py:146-153)
    EAE_HD int verify_map(uint32_t size, const int16_t* in, int* stage) {
        int s = bac.start_decoding();
        if (s) { *stage = STAGE_START; return s; }
        int mismatch = 0;
        for (uint32_t i = 0; i < size; i++) {
            int16_t v;
            s = read_signed_ueg0(v);
            if (s) { *stage = STAGE_DECODING; return s; }
            mismatch |= (v != in[i]);
        }
        return mismatch ? (int)MISMATCH : (int)OK;
    }
};

}  // namespace eae_core
