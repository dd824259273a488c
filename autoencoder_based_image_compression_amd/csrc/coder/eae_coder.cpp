// eae_coder.cpp -- host-side lossless coder behind the C ABI of include/eae_coder.h.
//
// What it replaces: kodak_tensorflow/lossless/c++/source/{Bitstream,BinaryArithmeticCoder,LosslessCoder,compression}.cpp
// (UEG0 binarisation -> 16-bit binary arithmetic coder with E1/E2/E3 rescaling + a bypass stream), reached in the
// reference through lossless/interface_cython.pyx:54-58, 127 times per image per rate point.
//
// How it differs (same bits, different machine):
//  * bit I/O is word-level: bits are gathered in a 64-bit little-endian accumulator and flushed 8 bytes at a time; the
//    reference writes one bit per call through std::vector::at (Bitstream.cpp:30-59). LSB-first packing is preserved
//    (stream bit i lives at byte i>>3, bit i&7), so byte streams are identical on little-endian hosts.
//  * E1/E2 renormalisation is closed-form: the number of shifts is the count of leading equal bits of low/high
//    (one clz), emitted as one multi-bit put; the reference loops bit by bit (BinaryArithmeticCoder.cpp:182-252).
//  * independent maps are coded concurrently by a persistent thread pool (the reference is single-threaded).
// What it keeps exactly: the interval arithmetic (double multiply + floor, BinaryArithmeticCoder.cpp:154), the
// non-standard RANGE_THREE_QUARTERS = 3*0x3FFF = 49149, the flush rule, the sticky-bit decoder priming, stream
// capacities and therefore every error condition and code.
#include "eae_coder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kRangeMax = 0xFFFFu;            // BinaryArithmeticCoder.cpp:16
constexpr uint32_t kRangeHalf = 0x7FFFu;           // :22
constexpr uint32_t kRangeQuarter = 0x3FFFu;        // :28
constexpr uint32_t kRangeThreeQuarters = 49149u;   // :29  (3 * 0x3FFF, not 0xBFFF)
constexpr uint32_t kMask = 0x8000u;                // :33

inline uint32_t round_up_to_byte(uint32_t bits) {  // utils.cpp:3-11 with divisor 8
    const uint32_t r = bits % 8u;
    return r ? bits + (8u - r) : bits;
}

inline uint8_t count_nb_bits(uint32_t x) {  // utils.cpp:13-28: floor(log2(x)) + 1, 1 for 0 -- integer form
    return x ? (uint8_t)(32 - __builtin_clz(x)) : (uint8_t)1;
}

inline uint32_t rev16(uint32_t v) {  // reverse the low 16 bits
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Bitstream (Bitstream.h:11-64): same observable behaviour, word-level storage.
// ---------------------------------------------------------------------------------------------------------------
struct Bitstream {
    uint8_t* data = nullptr;   // capacity_bytes + 16 bytes of slack so that whole-word flushes never overrun
    bool owns = false;
    uint32_t size_bits = 0;    // size_in_bits(): byte-rounded capacity
    uint32_t write_index = 0;
    uint32_t read_index = 0;
    uint64_t acc = 0;          // bits [write_index & ~63, write_index) not yet stored
    uint32_t limit_bytes = 0;  // bytes that may be read from `data`

    bool init_owned(uint32_t required_size_in_bits) {
        size_bits = round_up_to_byte(required_size_in_bits);
        const size_t bytes = (size_t)(size_bits >> 3) + 16;
        data = (uint8_t*)std::malloc(bytes);
        owns = true;
        write_index = read_index = 0;
        acc = 0;
        limit_bytes = (uint32_t)bytes;
        return data != nullptr;
    }
    void init_external(uint8_t* buf, uint32_t required_size_in_bits) {  // buf must hold capacity_bytes (+8 slack)
        size_bits = round_up_to_byte(required_size_in_bits);
        data = buf;
        owns = false;
        write_index = read_index = 0;
        acc = 0;
        limit_bytes = (size_bits >> 3) + 8;    // callers of init_external provide >= 8 bytes of slack
    }
    void init_reader(const uint8_t* buf, uint32_t nb_bits) {
        data = const_cast<uint8_t*>(buf);
        owns = false;
        size_bits = round_up_to_byte(nb_bits);
        write_index = nb_bits;
        read_index = 0;
        acc = 0;
        limit_bytes = size_bits >> 3;          // exact: never read past the caller's stream
    }
    ~Bitstream() { if (owns) std::free(data); }

    uint32_t occupancy() const { return write_index - read_index; }   // Bitstream.cpp:20-23
    bool entirely_read() const { return write_index == read_index; }  // Bitstream.cpp:25-28

    // Appends the n (<= 32) low bits of `bits`; bit 0 is the first bit in time. Equivalent to n write_bit calls
    // (Bitstream.cpp:30-59); fails with capacity_error iff one of them would.
    inline int put(uint32_t bits, uint32_t n) {
        if (write_index + n > size_bits) return EAE_CAPACITY_ERROR;
        const uint32_t sh = write_index & 63u;
        acc |= (uint64_t)bits << sh;
        if (sh + n >= 64u) {
            std::memcpy(data + ((write_index >> 6) << 3), &acc, 8);
            acc = sh ? ((uint64_t)bits >> (64u - sh)) : 0;
        }
        write_index += n;
        return EAE_SUCCESS;
    }
    // n copies of `bit` (the pending E3 queue, BinaryArithmeticCoder.cpp:322-337); n is unbounded.
    inline int put_run(uint32_t bit, uint32_t n) {
        const uint32_t word = bit ? 0xFFFFFFFFu : 0u;
        while (n >= 32u) {
            int s = put(word, 32);
            if (s) return s;
            n -= 32u;
        }
        return n ? put(word & ((1u << n) - 1u), n) : EAE_SUCCESS;
    }
    // Makes every written bit visible in `data` (partial last word included).
    inline void flush() {
        if (write_index & 63u) std::memcpy(data + ((write_index >> 6) << 3), &acc, 8);
    }
    // Up to 16 stream bits starting at read_index, first bit in time = MOST significant bit of the result
    // (what `code = (code << 1) | bit` builds). Caller guarantees n <= occupancy(). `data` has >= 4 readable bytes of
    // slack after the last stream byte only for internally owned buffers, so external streams are read bytewise near
    // their end (total_bytes).
    inline uint32_t take_msb_first(uint32_t n) {
        const uint32_t byte = read_index >> 3;
        uint32_t word;
        if (byte + 4u <= limit_bytes) {
            std::memcpy(&word, data + byte, 4);
        } else {
            word = 0;
            for (uint32_t i = 0; byte + i < limit_bytes && i < 4u; i++) word |= (uint32_t)data[byte + i] << (8u * i);
        }
        const uint32_t field = (word >> (read_index & 7u)) & ((1u << n) - 1u);   // bit j = j-th bit in time
        read_index += n;
        return rev16(field) >> (16u - n);
    }
    // Bitstream.cpp:61-79 ; caller guarantees flush() happened after the last put.
    inline int read_bit(uint8_t& storage) {
        if (read_index >= write_index) return EAE_RESOURCE_ERROR;
        storage = (uint8_t)((storage & 0xFE) | ((data[read_index >> 3] >> (read_index & 7u)) & 1u));
        read_index++;
        return EAE_SUCCESS;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// BinaryArithmeticCoder (BinaryArithmeticCoder.h:9-117)
// ---------------------------------------------------------------------------------------------------------------
struct Bac {
    Bitstream bs;
    uint32_t low = 0, middle = kRangeHalf, high = kRangeMax, nb_e3 = 0, code = 0;

    void reset() { low = 0; middle = kRangeHalf; high = kRangeMax; nb_e3 = 0; }  // :136-142 (code is kept)

    inline int update_middle(double p) {  // :144-156 ; !(p > 0 && p < 1) also catches NaN
        if (!(p > 0. && p < 1.)) return EAE_PROBABILITY_ERROR;
        middle = low + (uint32_t)std::floor(p * (double)(high - low));
        return EAE_SUCCESS;
    }

    // encoding() = encode_bit + rescale_encoding (:49-59, :158-252)
    inline int encode(uint32_t bit, double p) {
        int s = update_middle(p);
        if (s) return s;
        if (bit & 1u) low = middle + 1u; else high = middle;
        if (high > kRangeMax || low > kRangeMax) return EAE_PRECISION_ERROR;
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
        return EAE_SUCCESS;
    }

    int stop_encoding() {  // :61-102
        nb_e3++;
        const uint32_t b = (low < kRangeQuarter) ? 0u : 1u;
        int s = bs.put(b, 1);
        if (s) return s;
        s = bs.put_run(b ^ 1u, nb_e3);
        if (s) return s;
        nb_e3 = 0;
        reset();
        bs.flush();
        return EAE_SUCCESS;
    }

    int start_decoding() {  // :104-122 ; `storage` keeps its last value once the stream is exhausted
        uint8_t storage = 0;
        for (uint32_t i = 0; i < 16u; i++) {
            if (!bs.entirely_read()) {
                int s = bs.read_bit(storage);
                if (s) return s;
            }
            code = (code << 1) | storage;
        }
        return EAE_SUCCESS;
    }

    // decoding() = decode_bit + rescale_decoding (:124-134, :254-320). The renormalisation is closed-form like the
    // encoder's: E1/E2 fire exactly while the top bits of low and high agree (n = leading equal bits, one clz), then
    // only E3 can fire. E2's `code -= 0x8000` is absorbed by the 16-bit mask after the shift. Stream exhaustion keeps
    // the reference's semantics: `storage` starts at 0 in every rescale call and, once no bit is left, repeats the
    // last bit read IN THIS CALL (:275-277, :303-310).
    inline int decode(uint8_t& storage, double p) {
        int s = update_middle(p);
        if (s) return s;
        if (code >= low && code <= middle) { high = middle; storage = 0; }
        else if (code > middle && code <= high) { low = middle + 1u; storage = 1; }
        uint32_t sticky = 0;
        const uint32_t diff = (low ^ high) & 0xFFFFu;
        // rescale_decoding has no precision check; values stay within 16 bits for every valid state. For a corrupted
        // state (low > 0xFFFF cannot happen: low <= high <= 0xFFFF by construction) the clz form is still exact.
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
            if (bs.read_index < bs.write_index) {
                sticky = (uint32_t)((bs.data[bs.read_index >> 3] >> (bs.read_index & 7u)) & 1u);
                bs.read_index++;
            }
            high = ((high << 1) & kRangeMax) | 1u;
            low = (low << 1) & kRangeMax;
            code = ((code << 1) & kRangeMax) | sticky;
        }
        return EAE_SUCCESS;
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// LosslessCoder (LosslessCoder.h:12-169)
// ---------------------------------------------------------------------------------------------------------------
struct eae_lossless_coder {
    Bac bac;
    Bitstream bypass;
    uint32_t L = 0;
    double probabilities[256];

    void set_probabilities(uint8_t l, const double* p) {
        L = l;
        for (uint32_t i = 0; i < L; i++) probabilities[i] = p[i];
    }

    inline int write_sign(int16_t input) {  // LosslessCoder.cpp:22-37
        return input ? bypass.put(input < 0 ? 0u : 1u, 1) : EAE_SUCCESS;
    }
    inline int read_sign(int16_t& output) {  // :39-56
        if (output) {
            uint8_t storage = 0;
            int s = bypass.read_bit(storage);
            if (s) return s;
            if (!storage) output = (int16_t)(-(int)output);
        }
        return EAE_SUCCESS;
    }
    inline int write_eg0(uint16_t input) {  // :58-111 : n ones, a zero, then the n low bits of input+1, MSB first
        const uint32_t v = (uint32_t)input + 1u;
        const uint32_t n = (uint32_t)count_nb_bits(v) - 1u;  // 0..16
        // prefix: n ones then a zero -> n+1 bits, the zero last in time (= highest position)
        int s = bypass.put((1u << n) - 1u, n + 1u);
        if (s) return s;
        if (n) {
            const uint32_t suffix = v - (1u << n);
            // MSB of the suffix first in time -> bit-reverse the n-bit field
            s = bypass.put(rev16(suffix) >> (16u - n), n);
        }
        return s;
    }
    inline int read_eg0(uint16_t& output) {  // :113-165
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
        return EAE_SUCCESS;
    }
    inline int write_truncated_unary(uint16_t input) {  // :167-191
        if (L == 0) return EAE_OUT_OF_RANGE;  // m_probabilities.at(0) throws whatever the input
        const uint32_t ones = input < L ? input : L;
        for (uint32_t i = 0; i < ones; i++) {
            int s = bac.encode(1u, probabilities[i]);
            if (s) return s;
        }
        return input < L ? bac.encode(0u, probabilities[input]) : EAE_SUCCESS;
    }
    inline int read_truncated_unary(uint16_t& output) {  // :193-230
        output = 0;
        if (L == 0) return EAE_OUT_OF_RANGE;
        uint32_t i = 0;
        uint8_t storage = 0;
        for (;;) {
            int s = bac.decode(storage, probabilities[i]);
            if (s) return s;
            if (!storage) break;
            output++;
            if (i == L - 1u) break;
            i++;
        }
        return EAE_SUCCESS;
    }
    inline int write_signed_ueg0(int16_t input) {  // :232-252
        const uint16_t a = (uint16_t)std::abs((int)input);
        int s = write_truncated_unary(a);
        if (s) return s;
        if (a >= L) {
            s = write_eg0((uint16_t)(a - L));
            if (s) return s;
        }
        return write_sign(input);
    }
    inline int read_signed_ueg0(int16_t& output) {  // :254-276
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
};

namespace {

inline uint32_t required_bits(uint32_t size, uint8_t L) {  // compression.cpp:24 (uint32 arithmetic, as there)
    return size * std::max<uint32_t>(32u, (uint32_t)L);
}

// Encodes one map into two caller-visible streams held by `c` (already initialised). Returns status, sets stage.
inline int encode_map(eae_lossless_coder& c, uint32_t size, const int16_t* in, int* stage) {
    for (uint32_t i = 0; i < size; i++) {
        int s = c.write_signed_ueg0(in[i]);
        if (s) { *stage = EAE_STAGE_ENCODING; return s; }
    }
    int s = c.bac.stop_encoding();
    if (s) { *stage = EAE_STAGE_STOP_ENCODING; return s; }
    c.bypass.flush();
    return EAE_SUCCESS;
}
inline int decode_map(eae_lossless_coder& c, uint32_t size, int16_t* out, int* stage) {
    int s = c.bac.start_decoding();
    if (s) { *stage = EAE_STAGE_START_DECODING; return s; }
    for (uint32_t i = 0; i < size; i++) {
        s = c.read_signed_ueg0(out[i]);
        if (s) { *stage = EAE_STAGE_DECODING; return s; }
    }
    return EAE_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent thread pool: parallel_for over independent maps with dynamic (atomic counter) scheduling.
// ---------------------------------------------------------------------------------------------------------------
class Pool {
public:
    static Pool& instance() { static Pool p; return p; }

    void parallel_for(uint32_t n_items, int n_threads, const std::function<void(uint32_t, int)>& fn) {
        if (n_items == 0) return;
        int hw = (int)std::thread::hardware_concurrency();
        if (hw <= 0) hw = 1;
        if (n_threads <= 0 || n_threads > hw) n_threads = hw;
        if ((uint32_t)n_threads > n_items) n_threads = (int)n_items;
        if (n_threads <= 1) {
            for (uint32_t i = 0; i < n_items; i++) fn(i, 0);
            return;
        }
        std::lock_guard<std::mutex> call_lock(call_mutex_);  // one parallel region at a time
        ensure_workers(n_threads - 1);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            n_items_ = n_items;
            next_.store(0, std::memory_order_relaxed);
            wanted_ = n_threads - 1;
            running_ = wanted_;
            generation_++;
        }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return running_ == 0; });
        job_ = nullptr;
    }

private:
    Pool() = default;
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void ensure_workers(int n) {
        while ((int)workers_.size() < n) {
            const int id = (int)workers_.size() + 1;
            workers_.emplace_back([this, id] { loop(id); });
        }
    }
    void work(int tid) {
        for (;;) {
            const uint32_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_items_) break;
            (*job_)(i, tid);
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (generation_ != seen && id <= wanted_); });
                if (stop_) return;
                seen = generation_;
            }
            work(id);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--running_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::mutex call_mutex_, m_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(uint32_t, int)>* job_ = nullptr;
    std::atomic<uint32_t> next_{0};
    uint32_t n_items_ = 0;
    int wanted_ = 0, running_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// Per-thread scratch streams for the batched entry points, reused across maps (no malloc per map).
struct Scratch {
    std::vector<uint8_t> bac, bypass;
    std::vector<int16_t> decoded;
    void reserve(size_t bytes) {
        if (bac.size() < bytes) { bac.resize(bytes); bypass.resize(bytes); }
    }
};

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* eae_coder_version(void) { return "eae_coder 1.0 (UEG0 + 16-bit BAC, bit-exact with the reference coder)"; }

uint8_t eae_coder_count_nb_bits(uint32_t input) { return count_nb_bits(input); }

uint32_t eae_coder_stream_capacity_bytes(uint32_t size, uint8_t L) {
    return round_up_to_byte(required_bits(size, L)) >> 3;
}

int eae_coder_compress_lossless(uint32_t size, const int16_t* in, int16_t* out, uint8_t L, const double* probs,
                                uint32_t* nb_bits, int* stage) {
    int dummy = 0;
    if (!stage) stage = &dummy;
    *stage = EAE_STAGE_NONE;
    if (!in || !out || !probs) return EAE_NULL_POINTER;
    eae_lossless_coder c;
    const uint32_t req = required_bits(size, L);
    if (!c.bac.bs.init_owned(req) || !c.bypass.init_owned(req)) return EAE_BAD_ALLOC;
    c.set_probabilities(L, probs);
    int s = encode_map(c, size, in, stage);
    if (s) return s;
    if (nb_bits) *nb_bits = c.bac.bs.occupancy() + c.bypass.occupancy();  // compression.cpp:49
    return decode_map(c, size, out, stage);
}

int eae_coder_encode(uint32_t size, const int16_t* in, uint8_t L, const double* probs,
                     uint8_t* bac_bytes, uint32_t* bac_bits, uint8_t* bypass_bytes, uint32_t* bypass_bits, int* stage) {
    int dummy = 0;
    if (!stage) stage = &dummy;
    *stage = EAE_STAGE_NONE;
    if (!in || !probs || !bac_bytes || !bypass_bytes || !bac_bits || !bypass_bits) return EAE_NULL_POINTER;
    eae_lossless_coder c;
    const uint32_t req = required_bits(size, L);
    if (!c.bac.bs.init_owned(req) || !c.bypass.init_owned(req)) return EAE_BAD_ALLOC;
    c.set_probabilities(L, probs);
    int s = encode_map(c, size, in, stage);
    if (s) return s;
    *bac_bits = c.bac.bs.write_index;
    *bypass_bits = c.bypass.write_index;
    std::memcpy(bac_bytes, c.bac.bs.data, (c.bac.bs.write_index + 7u) >> 3);
    std::memcpy(bypass_bytes, c.bypass.data, (c.bypass.write_index + 7u) >> 3);
    return EAE_SUCCESS;
}

int eae_coder_decode(uint32_t size, int16_t* out, uint8_t L, const double* probs,
                     const uint8_t* bac_bytes, uint32_t bac_bits, const uint8_t* bypass_bytes, uint32_t bypass_bits,
                     int* stage) {
    int dummy = 0;
    if (!stage) stage = &dummy;
    *stage = EAE_STAGE_NONE;
    if (!out || !probs || !bac_bytes || !bypass_bytes) return EAE_NULL_POINTER;
    eae_lossless_coder c;
    c.bac.bs.init_reader(bac_bytes, bac_bits);
    c.bypass.init_reader(bypass_bytes, bypass_bits);
    c.set_probabilities(L, probs);
    return decode_map(c, size, out, stage);
}

int eae_coder_compress_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, int16_t* reconstruction,
                            uint8_t L, const double* probs, const int32_t* prob_row,
                            uint32_t* nb_bits, int32_t* status, int32_t* stage, int mode, int n_threads) {
    if (!symbols || !probs || !nb_bits || !status) return EAE_NULL_POINTER;
    if (mode == EAE_MODE_ROUNDTRIP && !reconstruction) return EAE_NULL_POINTER;
    const bool verify = mode == EAE_MODE_ROUNDTRIP_VERIFY;
    const uint32_t req = required_bits(map_size, L);
    const size_t cap_bytes = (size_t)(round_up_to_byte(req) >> 3) + 16;
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    const int nt = (n_threads <= 0 || n_threads > hw) ? hw : n_threads;
    std::vector<Scratch> scratch((size_t)nt);
    std::atomic<int> first_error{0};
    auto body = [&](uint32_t m, int tid) {
        const int32_t row = prob_row ? prob_row[m] : (int32_t)m;
        const int16_t* in = symbols + (size_t)m * map_size;
        int st = EAE_STAGE_NONE;
        int s = EAE_SUCCESS;
        if (row < 0) {  // exception map: passed through, costed by the caller (compression.py:68-75)
            nb_bits[m] = 0;
            if (reconstruction) std::memcpy(reconstruction + (size_t)m * map_size, in, (size_t)map_size * 2);
        } else {
            Scratch& sc = scratch[(size_t)tid];
            sc.reserve(cap_bytes);
            eae_lossless_coder c;
            c.bac.bs.init_external(sc.bac.data(), req);
            c.bypass.init_external(sc.bypass.data(), req);
            c.set_probabilities(L, probs + (size_t)row * L);
            s = encode_map(c, map_size, in, &st);
            if (!s) {
                nb_bits[m] = c.bac.bs.occupancy() + c.bypass.occupancy();
                if (mode == EAE_MODE_ROUNDTRIP) {
                    s = decode_map(c, map_size, reconstruction + (size_t)m * map_size, &st);
                } else if (verify) {      // decode into scratch and compare: compression.py:146-153 without the copy
                    if (sc.decoded.size() < map_size) sc.decoded.resize(map_size);
                    s = decode_map(c, map_size, sc.decoded.data(), &st);
                    if (!s && std::memcmp(sc.decoded.data(), in, (size_t)map_size * 2) != 0) s = EAE_ROUNDTRIP_MISMATCH;
                }
            }
        }
        status[m] = s;
        if (stage) stage[m] = st;
        if (s) { int z = 0; first_error.compare_exchange_strong(z, s); }
    };
    Pool::instance().parallel_for(n_maps, nt, body);
    return first_error.load();
}

int eae_coder_encode_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L, const double* probs,
                          const int32_t* prob_row, uint8_t* streams, uint64_t stride,
                          uint32_t* bac_bits, uint32_t* bypass_bits, int32_t* status, int32_t* stage, int n_threads) {
    if (!symbols || !probs || !streams || !bac_bits || !bypass_bits || !status) return EAE_NULL_POINTER;
    const uint32_t req = required_bits(map_size, L);
    const uint64_t half = stride / 2;
    if (half < (uint64_t)(round_up_to_byte(req) >> 3) + 16) return EAE_CAPACITY_ERROR;
    std::atomic<int> first_error{0};
    auto body = [&](uint32_t m, int) {
        const int32_t row = prob_row ? prob_row[m] : (int32_t)m;
        int st = EAE_STAGE_NONE, s = EAE_SUCCESS;
        bac_bits[m] = bypass_bits[m] = 0;
        if (row >= 0) {
            eae_lossless_coder c;
            c.bac.bs.init_external(streams + (uint64_t)m * stride, req);
            c.bypass.init_external(streams + (uint64_t)m * stride + half, req);
            c.set_probabilities(L, probs + (size_t)row * L);
            s = encode_map(c, map_size, symbols + (size_t)m * map_size, &st);
            if (!s) { bac_bits[m] = c.bac.bs.write_index; bypass_bits[m] = c.bypass.write_index; }
        }
        status[m] = s;
        if (stage) stage[m] = st;
        if (s) { int z = 0; first_error.compare_exchange_strong(z, s); }
    };
    Pool::instance().parallel_for(n_maps, n_threads, body);
    return first_error.load();
}

int eae_coder_decode_maps(uint32_t n_maps, uint32_t map_size, int16_t* out, uint8_t L, const double* probs,
                          const int32_t* prob_row, const uint8_t* streams, uint64_t stride,
                          const uint32_t* bac_bits, const uint32_t* bypass_bits, int32_t* status, int32_t* stage,
                          int n_threads) {
    if (!out || !probs || !streams || !bac_bits || !bypass_bits || !status) return EAE_NULL_POINTER;
    const uint64_t half = stride / 2;
    std::atomic<int> first_error{0};
    auto body = [&](uint32_t m, int) {
        const int32_t row = prob_row ? prob_row[m] : (int32_t)m;
        int st = EAE_STAGE_NONE, s = EAE_SUCCESS;
        if (row >= 0) {
            eae_lossless_coder c;
            c.bac.bs.init_reader(streams + (uint64_t)m * stride, bac_bits[m]);
            c.bypass.init_reader(streams + (uint64_t)m * stride + half, bypass_bits[m]);
            c.set_probabilities(L, probs + (size_t)row * L);
            s = decode_map(c, map_size, out + (size_t)m * map_size, &st);
        }
        status[m] = s;
        if (stage) stage[m] = st;
        if (s) { int z = 0; first_error.compare_exchange_strong(z, s); }
    };
    Pool::instance().parallel_for(n_maps, n_threads, body);
    return first_error.load();
}

// ---- LosslessCoder object -------------------------------------------------------------------------------------------
eae_lossless_coder* eae_lossless_coder_new(uint32_t required_size_in_bits, uint8_t L, const double* probs) {
    eae_lossless_coder* c = new (std::nothrow) eae_lossless_coder();
    if (!c) return nullptr;
    if (!c->bac.bs.init_owned(required_size_in_bits) || !c->bypass.init_owned(required_size_in_bits)) {
        delete c;
        return nullptr;
    }
    if (probs) c->set_probabilities(L, probs);
    return c;
}
void eae_lossless_coder_free(eae_lossless_coder* c) { delete c; }
uint32_t eae_lossless_coder_occupancy_in_bits_bac(const eae_lossless_coder* c) { return c->bac.bs.occupancy(); }
uint32_t eae_lossless_coder_occupancy_in_bits_bypass(const eae_lossless_coder* c) { return c->bypass.occupancy(); }
uint32_t eae_lossless_coder_written_bits_bac(const eae_lossless_coder* c) { return c->bac.bs.write_index; }
uint32_t eae_lossless_coder_written_bits_bypass(const eae_lossless_coder* c) { return c->bypass.write_index; }
static uint32_t copy_stream(const Bitstream& b, uint8_t* dst, uint32_t cap) {
    const_cast<Bitstream&>(b).flush();
    const uint32_t n = (b.write_index + 7u) >> 3;
    if (dst && cap >= n) std::memcpy(dst, b.data, n);
    return n;
}
uint32_t eae_lossless_coder_copy_bac(const eae_lossless_coder* c, uint8_t* dst, uint32_t cap) { return copy_stream(c->bac.bs, dst, cap); }
uint32_t eae_lossless_coder_copy_bypass(const eae_lossless_coder* c, uint8_t* dst, uint32_t cap) { return copy_stream(c->bypass, dst, cap); }
int eae_lossless_coder_write_sign(eae_lossless_coder* c, int16_t v) { return c->write_sign(v); }
int eae_lossless_coder_read_sign(eae_lossless_coder* c, int16_t* v) { c->bypass.flush(); return c->read_sign(*v); }
int eae_lossless_coder_write_eg0(eae_lossless_coder* c, uint16_t v) { return c->write_eg0(v); }
int eae_lossless_coder_read_eg0(eae_lossless_coder* c, uint16_t* v) { c->bypass.flush(); return c->read_eg0(*v); }
int eae_lossless_coder_write_truncated_unary(eae_lossless_coder* c, uint16_t v) { return c->write_truncated_unary(v); }
int eae_lossless_coder_read_truncated_unary(eae_lossless_coder* c, uint16_t* v) { return c->read_truncated_unary(*v); }
int eae_lossless_coder_write_signed_ueg0(eae_lossless_coder* c, int16_t v) { return c->write_signed_ueg0(v); }
int eae_lossless_coder_read_signed_ueg0(eae_lossless_coder* c, int16_t* v) { c->bypass.flush(); return c->read_signed_ueg0(*v); }
int eae_lossless_coder_stop_bac_encoding(eae_lossless_coder* c) { return c->bac.stop_encoding(); }
int eae_lossless_coder_start_bac_decoding(eae_lossless_coder* c) { c->bac.bs.flush(); return c->bac.start_decoding(); }
int eae_lossless_coder_bac_encoding(eae_lossless_coder* c, uint8_t bit, double p) { return c->bac.encode(bit, p); }
int eae_lossless_coder_bac_decoding(eae_lossless_coder* c, uint8_t* storage, double p) { return c->bac.decode(*storage, p); }

// ---- statistics (lossless/stats.py:136-195) ------------------------------------------------------------------------
int eae_coder_count_binary_decisions(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, uint8_t L,
                                     int64_t* zeros, int64_t* ones, int n_threads) {
    if (!symbols || !zeros || !ones) return EAE_NULL_POINTER;
    auto body = [&](uint32_t m, int) {
        // histogram of min(|s|, L), then the prefix sums of stats.py:188-194
        std::vector<int64_t> hist((size_t)L + 1, 0);
        const int16_t* s = symbols + (size_t)m * map_size;
        for (uint32_t i = 0; i < map_size; i++) {
            uint32_t a = (uint32_t)std::abs((int)s[i]);
            hist[a < L ? a : L]++;
        }
        int64_t* z = zeros + (size_t)m * L;
        int64_t* o = ones + (size_t)m * L;
        int64_t above = hist[L];  // symbols with |s| >= L : ones[:] += n
        for (int j = (int)L - 1; j >= 0; j--) {
            z[j] += hist[(size_t)j];
            o[j] += above;       // every |s| > j contributes a 1 at position j
            above += hist[(size_t)j];
        }
    };
    Pool::instance().parallel_for(n_maps, n_threads, body);
    return EAE_SUCCESS;
}

}  // extern "C"
