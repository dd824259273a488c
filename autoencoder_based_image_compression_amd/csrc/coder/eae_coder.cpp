// eae_coder.cpp -- host-side lossless coder behind the C ABI of include/eae_coder.h.
//
// Replaces kodak_tensorflow/lossless/c++/source/*.cpp as reached through lossless/interface_cython.pyx:54-58 (127 calls
// per image per rate point in the reference). The arithmetic lives in coder_core.h, shared source with the gfx950
// kernel that codes one map per lane (hip/coder_device.hip); this file adds buffer ownership, the persistent thread
// pool over independent maps (the reference is single-threaded) and the extern "C" surface.
#include "eae_coder.h"

#include <nmmintrin.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#if defined(__linux__)
#include <sched.h>
#endif

#include "coder_core.h"

using eae_core::Bitstream;
using eae_core::round_up_to_byte;
using eae_core::required_bits;

// The LosslessCoder object of the C ABI: the shared core plus host-owned stream buffers and a private copy of the
// probabilities (LosslessCoder.cpp:3-10 copies them into m_probabilities).
struct eae_lossless_coder {
    eae_core::LosslessCoder core;
    std::vector<uint8_t> bac_buf, bypass_buf;
    double probabilities[256];

    bool init_owned(uint32_t required_size_in_bits, uint8_t L, const double* p) {
        const size_t bytes = (size_t)(round_up_to_byte(required_size_in_bits) >> 3) + 16;
        try {
            bac_buf.assign(bytes, 0);
            bypass_buf.assign(bytes, 0);
        } catch (const std::bad_alloc&) {
            return false;
        }
        core.bac.init();
        core.bac.bs.init_writer(bac_buf.data(), required_size_in_bits);
        core.bypass.init_writer(bypass_buf.data(), required_size_in_bits);
        core.L = L;
        core.prob_stride = 1;
        for (uint32_t i = 0; i < L && p; i++) probabilities[i] = p[i];
        core.probabilities = probabilities;
        return true;
    }
};

namespace {

// A coder over caller-provided stream memory (scratch per thread, or the caller's stream region).
inline void init_external(eae_core::LosslessCoder& c, uint8_t* bac, uint8_t* bypass, uint32_t req, uint8_t L, const double* p) {
    c.bac.init();
    c.bac.bs.init_writer(bac, req);
    c.bypass.init_writer(bypass, req);
    c.L = L;
    c.prob_stride = 1;
    c.probabilities = p;
}

// CPUs this process may really use: the affinity mask capped by the cgroup-v2 CPU quota (a container may see 256 hardware
// threads behind a 16-CPU quota; a pool sized for 256 only thrashes). Falls back to hardware_concurrency().
static int usable_cpus() {
    static const int cached = [] {
        int n = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
                const long q = std::atol(quota) / period;
                if (q >= 1 && q < n) n = (int)q;
            }
            std::fclose(f);
        }
#endif
        return n > 0 ? n : 1;
    }();
    return cached;
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent thread pool: parallel_for over independent maps with dynamic (atomic counter) scheduling.
// ---------------------------------------------------------------------------------------------------------------
class Pool {
public:
    static Pool& instance() { static Pool p; return p; }

    void parallel_for(uint32_t n_items, int n_threads, const std::function<void(uint32_t, int)>& fn) {
        if (n_items == 0) return;
        const int hw = usable_cpus();
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
    std::vector<uint64_t> bac, bypass;      // uint64: 8-byte aligned for the word stores
    void reserve(size_t bytes) {
        const size_t words = (bytes + 7) / 8;
        if (bac.size() < words) { bac.resize(words); bypass.resize(words); }
    }
};

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* eae_coder_version(void) { return "eae_coder 1.1 (UEG0 + 16-bit BAC, bit-exact with the reference coder; core shared with the gfx950 kernel)"; }

uint8_t eae_coder_count_nb_bits(uint32_t input) { return (uint8_t)eae_core::count_nb_bits(input); }

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
    if (!c.init_owned(required_bits(size, L), L, probs)) return EAE_BAD_ALLOC;
    int s = c.core.encode_map(size, in, stage);
    if (s) return s;
    if (nb_bits) *nb_bits = c.core.bac.bs.occupancy() + c.core.bypass.occupancy();  // compression.cpp:49
    return c.core.decode_map(size, out, stage);
}

int eae_coder_encode(uint32_t size, const int16_t* in, uint8_t L, const double* probs,
                     uint8_t* bac_bytes, uint32_t* bac_bits, uint8_t* bypass_bytes, uint32_t* bypass_bits, int* stage) {
    int dummy = 0;
    if (!stage) stage = &dummy;
    *stage = EAE_STAGE_NONE;
    if (!in || !probs || !bac_bytes || !bypass_bytes || !bac_bits || !bypass_bits) return EAE_NULL_POINTER;
    eae_lossless_coder c;
    if (!c.init_owned(required_bits(size, L), L, probs)) return EAE_BAD_ALLOC;
    int s = c.core.encode_map(size, in, stage);
    if (s) return s;
    *bac_bits = c.core.bac.bs.write_index;
    *bypass_bits = c.core.bypass.write_index;
    std::memcpy(bac_bytes, c.bac_buf.data(), (*bac_bits + 7u) >> 3);
    std::memcpy(bypass_bytes, c.bypass_buf.data(), (*bypass_bits + 7u) >> 3);
    return EAE_SUCCESS;
}

int eae_coder_decode(uint32_t size, int16_t* out, uint8_t L, const double* probs,
                     const uint8_t* bac_bytes, uint32_t bac_bits, const uint8_t* bypass_bytes, uint32_t bypass_bits,
                     int* stage) {
    int dummy = 0;
    if (!stage) stage = &dummy;
    *stage = EAE_STAGE_NONE;
    if (!out || !probs || !bac_bytes || !bypass_bytes) return EAE_NULL_POINTER;
    eae_core::LosslessCoder c;
    c.bac.init();
    c.bac.bs.init_reader(bac_bytes, bac_bits);
    c.bypass.init_reader(bypass_bytes, bypass_bits);
    c.L = L;
    c.prob_stride = 1;
    c.probabilities = probs;
    return c.decode_map(size, out, stage);
}

int eae_coder_compress_maps(uint32_t n_maps, uint32_t map_size, const int16_t* symbols, int16_t* reconstruction,
                            uint8_t L, const double* probs, const int32_t* prob_row,
                            uint32_t* nb_bits, int32_t* status, int32_t* stage, int mode, int n_threads) {
    if (!symbols || !probs || !nb_bits || !status) return EAE_NULL_POINTER;
    if (mode == EAE_MODE_ROUNDTRIP && !reconstruction) return EAE_NULL_POINTER;
    const bool verify = mode == EAE_MODE_ROUNDTRIP_VERIFY;
    const uint32_t req = required_bits(map_size, L);
    const size_t cap_bytes = (size_t)(round_up_to_byte(req) >> 3) + 16;
    const int hw = usable_cpus();
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
            eae_core::LosslessCoder c;
            init_external(c, (uint8_t*)sc.bac.data(), (uint8_t*)sc.bypass.data(), req, L, probs + (size_t)row * L);
            s = c.encode_map(map_size, in, &st);
            if (!s) {
                nb_bits[m] = c.bac.bs.occupancy() + c.bypass.occupancy();
                if (mode == EAE_MODE_ROUNDTRIP) s = c.decode_map(map_size, reconstruction + (size_t)m * map_size, &st);
                else if (verify) s = c.verify_map(map_size, in, &st);   // compression.py:146-153 without the copy
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
    if (half < (uint64_t)(round_up_to_byte(req) >> 3) + 16 || (stride & 15u) || ((uintptr_t)streams & 7u)) return EAE_CAPACITY_ERROR;
    std::atomic<int> first_error{0};
    auto body = [&](uint32_t m, int) {
        const int32_t row = prob_row ? prob_row[m] : (int32_t)m;
        int st = EAE_STAGE_NONE, s = EAE_SUCCESS;
        bac_bits[m] = bypass_bits[m] = 0;
        if (row >= 0) {
            eae_core::LosslessCoder c;
            init_external(c, streams + (uint64_t)m * stride, streams + (uint64_t)m * stride + half, req, L, probs + (size_t)row * L);
            s = c.encode_map(map_size, symbols + (size_t)m * map_size, &st);
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
            eae_core::LosslessCoder c;
            c.bac.init();
            c.bac.bs.init_reader(streams + (uint64_t)m * stride, bac_bits[m]);
            c.bypass.init_reader(streams + (uint64_t)m * stride + half, bypass_bits[m]);
            c.L = L;
            c.prob_stride = 1;
            c.probabilities = probs + (size_t)row * L;
            s = c.decode_map(map_size, out + (size_t)m * map_size, &st);
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
    if (!c->init_owned(required_size_in_bits, L, probs)) {
        delete c;
        return nullptr;
    }
    return c;
}
void eae_lossless_coder_free(eae_lossless_coder* c) { delete c; }
uint32_t eae_lossless_coder_occupancy_in_bits_bac(const eae_lossless_coder* c) { return c->core.bac.bs.occupancy(); }
uint32_t eae_lossless_coder_occupancy_in_bits_bypass(const eae_lossless_coder* c) { return c->core.bypass.occupancy(); }
uint32_t eae_lossless_coder_written_bits_bac(const eae_lossless_coder* c) { return c->core.bac.bs.write_index; }
uint32_t eae_lossless_coder_written_bits_bypass(const eae_lossless_coder* c) { return c->core.bypass.write_index; }
static uint32_t copy_stream(const Bitstream& b, uint8_t* dst, uint32_t cap) {
    const_cast<Bitstream&>(b).flush();
    const uint32_t n = (b.write_index + 7u) >> 3;
    if (dst && cap >= n) std::memcpy(dst, b.data, n);
    return n;
}
uint32_t eae_lossless_coder_copy_bac(const eae_lossless_coder* c, uint8_t* dst, uint32_t cap) { return copy_stream(c->core.bac.bs, dst, cap); }
uint32_t eae_lossless_coder_copy_bypass(const eae_lossless_coder* c, uint8_t* dst, uint32_t cap) { return copy_stream(c->core.bypass, dst, cap); }
int eae_lossless_coder_write_sign(eae_lossless_coder* c, int16_t v) { return c->core.write_sign(v); }
int eae_lossless_coder_read_sign(eae_lossless_coder* c, int16_t* v) { c->core.bypass.flush(); c->core.bypass.sync_reader(); return c->core.read_sign(*v); }
int eae_lossless_coder_write_eg0(eae_lossless_coder* c, uint16_t v) { return c->core.write_eg0(v); }
int eae_lossless_coder_read_eg0(eae_lossless_coder* c, uint16_t* v) { c->core.bypass.flush(); c->core.bypass.sync_reader(); return c->core.read_eg0(*v); }
int eae_lossless_coder_write_truncated_unary(eae_lossless_coder* c, uint16_t v) { return c->core.write_truncated_unary(v); }
int eae_lossless_coder_read_truncated_unary(eae_lossless_coder* c, uint16_t* v) { return c->core.read_truncated_unary(*v); }
int eae_lossless_coder_write_signed_ueg0(eae_lossless_coder* c, int16_t v) { return c->core.write_signed_ueg0(v); }
int eae_lossless_coder_read_signed_ueg0(eae_lossless_coder* c, int16_t* v) { c->core.bypass.flush(); c->core.bypass.sync_reader(); return c->core.read_signed_ueg0(*v); }
int eae_lossless_coder_stop_bac_encoding(eae_lossless_coder* c) { return c->core.bac.stop_encoding(); }
int eae_lossless_coder_start_bac_decoding(eae_lossless_coder* c) { c->core.bac.bs.flush(); c->core.bac.bs.sync_reader(); return c->core.bac.start_decoding(); }
int eae_lossless_coder_bac_encoding(eae_lossless_coder* c, uint8_t bit, double p) { return c->core.bac.encode(bit, p); }
int eae_lossless_coder_bac_decoding(eae_lossless_coder* c, uint8_t* storage, double p) { return c->core.bac.decode(*storage, p); }

// ---- statistics (lossless/stats.py:136-195) ------------------------------------------------------------------------
// numpy's pairwise summation of a contiguous float64 run (numpy/_core/src/umath/loops_utils.h.src, unchanged since 1.9 but for the
// -0.0 start of the short form): n < 8 one running sum; n <= 128 eight running sums over the multiples of eight, combined as a
// balanced tree, then the rest; longer runs halved at a multiple of eight. Compiled without contraction or reassociation (Makefile).
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

int eae_coder_pairwise_row_sums(const double* values, const int64_t* bounds, int64_t rows, double* sums) {
    if (!bounds || !sums || (!values && rows > 0 && bounds[rows] > bounds[0])) return EAE_NULL_POINTER;
    for (int64_t i = 0; i < rows; ++i) {
        const int64_t n = bounds[i + 1] - bounds[i];
        if (n < 0) return EAE_OUT_OF_RANGE;
        // numpy.sum starts from the identity of the addition, +0.0, and adds the array's pairwise sum to it
        sums[i] = 0.0 + pairwise_sum(values + bounds[i], n);
    }
    return EAE_SUCCESS;
}

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

// ---- CRC-32C for checkpoint ingestion (the SSE4.2 crc32 instruction implements exactly this polynomial) ---------------
uint32_t eae_crc32c(const void* data, size_t size, uint32_t crc) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint64_t state = (uint32_t)~crc;
    while (size && ((uintptr_t)p & 7)) { state = _mm_crc32_u8((uint32_t)state, *p++); --size; }
    for (; size >= 8; size -= 8, p += 8) {
        uint64_t word;
        std::memcpy(&word, p, 8);
        state = _mm_crc32_u64(state, word);
    }
    while (size--) state = _mm_crc32_u8((uint32_t)state, *p++);
    return ~(uint32_t)state;
}

}  // extern "C"
