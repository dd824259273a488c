// misc.hip -- library identity and device probing.
#include "common.h"
#include "conv_gemm.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

// ---- kernel-form overrides: the environment is read here, once per library load (and on a test's explicit request) ----
// g_eae_launch_options is a plain global: the two eae_hip_debug_* setters below are TEST HOOKS, compiled only into the test build
// (-DEAE_TEST_HOOKS: lib/libeae_hip_test.so, csrc/Makefile), to be called from the one thread that launches, with nothing in
// flight (tests/conftest.py: launch_options); the product library does not hold them, so its launches only ever read.
static EaeLaunchOptions read_launch_options() {
    EaeLaunchOptions o{};
    const char* e = std::getenv("EAE_HIP_GEMM");
    o.gemm = e && (e[0] == 's' || e[0] == 'u' || e[0] == 'w' || e[0] == 'l') ? e[0] : 0;
    if (e && e[0] && !o.gemm) std::fprintf(stderr, "libeae_hip: EAE_HIP_GEMM=%s is none of s / u / w / l: ignored (form chosen by shape)\n", e);
    e = std::getenv("EAE_HIP_SPLIT_WAVES");
    o.split_waves = e ? std::atoi(e) : 3;
    if (o.split_waves < 1 || o.split_waves > 3) o.split_waves = 3;
    e = std::getenv("EAE_HIP_FORCE_TILE");
    o.force_tile = e ? std::atoi(e) : 0;
    e = std::getenv("EAE_HIP_FORCE_NT");
    o.force_nt = e ? std::atoi(e) : 0;
    e = std::getenv("EAE_HIP_LATENT");
    o.latent = e && (e[0] == 'w' || e[0] == 'l' || e[0] == 'q') ? e[0] : 'q';
    if (std::getenv("EAE_HIP_LATENT_LDS") != nullptr) o.latent = 'l';      // round 1's name for the LDS form
    e = std::getenv("EAE_HIP_PACK");
    o.pack = e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1;
    e = std::getenv("EAE_HIP_SPLIT_WPB");
    o.split_wpb = e && (std::atoi(e) == 1 || std::atoi(e) == 4) ? std::atoi(e) : 0;
    o.split_mute = 0;                                                      // never from the environment
    e = std::getenv("EAE_HIP_ASSUME_PARTITIONED");
    o.assume_partitioned = e && e[0] == '1' ? 1 : 0;
    return o;
}
EaeLaunchOptions g_eae_launch_options = read_launch_options();

#ifdef EAE_TEST_HOOKS
extern "C" int eae_hip_debug_reload_launch_options(void) {
    const int mute = g_eae_launch_options.split_mute;
    g_eae_launch_options = read_launch_options();
    g_eae_launch_options.split_mute = mute;
    return EAE_HIP_OK;
}
extern "C" int eae_hip_debug_set_split_mute(int on) {
    g_eae_launch_options.split_mute = on ? 1 : 0;
    return EAE_HIP_OK;
}
#endif

extern "C" int eae_hip_partition_info(int* compute_units, int* xcds, int* whole_device) {
    const int cus = eae_compute_units();
    if (cus <= 0) return (int)hipErrorNoDevice;
    if (compute_units) *compute_units = cus;
    if (xcds) *xcds = eae_is_gfx950() ? (cus + 31) / 32 : 0;
    if (whole_device) *whole_device = eae_is_whole_mi355x() ? 1 : 0;
    static bool told = false;
    if (!told && eae_is_gfx950() && !eae_is_whole_mi355x()) {
        told = true;
        std::fprintf(stderr, "libeae_hip: this logical device shows %d compute units, not the 256 of a whole MI355X (compute partition "
                             "DPX / QPX / CPX?): the conv launches keep whole tiles (same results, the tile order is tuned for SPX)\n", cus);
    }
    return EAE_HIP_OK;
}

extern "C" const char* eae_hip_version(void) { return "eae_hip 1.0 (gfx950, f32 MFMA, bit-exact vs oracle/transforms_oracle.c)"; }

extern "C" int eae_hip_device_info(char* name, int name_cap, int* compute_units, int* clock_mhz, int64_t* hbm_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (name && name_cap > 0) {
        std::strncpy(name, p.gcnArchName, (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return EAE_HIP_OK;
}

// Results for the host, without hipMemcpyAsync (which was measured to block the calling thread for the whole queue depth
// of its stream on this platform when the stream is busy): a kernel writes the bytes into pinned, device-mapped host
// memory; the host reads them after an event on the same stream.
__global__ void publish_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t words) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
// A step counter for the host: bumps a device word and leaves the new value in pinned host memory, behind everything the stream
// has done so far (a kernel boundary orders it after the publish_kernel in front of it). A host thread that knows how many times
// the step was submitted waits for that value with plain loads -- no event to record, query or wait on.
__global__ void sequence_kernel(uint32_t* __restrict__ counter, volatile uint32_t* __restrict__ host_word) {
    const uint32_t v = *counter + 1u;
    *counter = v;
    *host_word = v;
    __threadfence_system();
}
extern "C" int eae_hip_publish_sequence(void* counter_device, void* word_host_mapped, void* stream) {
    if (!counter_device || !word_host_mapped) return -1;
    hipLaunchKernelGGL(sequence_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t*)counter_device, (volatile uint32_t*)word_host_mapped);
    return (int)hipGetLastError();
}

// The end of one side of a step in ONE launch (a launch costs the submitting thread ~10 us, and a step of one image is two dozen of
// them): [the conv workspace's error word collected, as conv_workspace_collect_kernel does] -> the result block copied to pinned
// memory -> the accumulators among it (words from `clear_from` on) zeroed for the slot's next step -> the step counter bumped and
// left in pinned memory behind the copy. Several blocks copy; the last one through (a ticket word, left at zero) bumps the counter.
__global__ __launch_bounds__(256) void publish_step_kernel(uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t words, uint64_t clear_from,
                                                           unsigned int* ws, uint32_t* error_word, uint32_t* tickets, uint32_t* counter,
                                                           volatile uint32_t* host_word) {
    if (ws != nullptr) {      // one block (the entry point sees to it): the error word lies in `src`
        const unsigned int e = ws[eae_conv_gemm::SPLIT_ERROR_WORD];
        if (e != 0u) {
            __syncthreads();
            for (int i = threadIdx.x; i < eae_conv_gemm::SPLIT_WORDS; i += blockDim.x) ws[i] = 0u;
            if (threadIdx.x == 0) atomicAdd(error_word, e);
            __threadfence();
        }
        __syncthreads();
    }
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
        dst[i] = __builtin_nontemporal_load(src + i);
        if (i >= clear_from) src[i] = 0u;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = gridDim.x == 1u || atomicAdd(tickets, 1u) == gridDim.x - 1u;
        if (last) {
            if (gridDim.x != 1u) *tickets = 0u;
            __threadfence_system();      // the other blocks' copies (fenced before their tickets) before the counter
            const uint32_t v = *counter + 1u;
            *counter = v;
            *host_word = v;
            __threadfence_system();
        }
    }
}
extern "C" int eae_hip_publish_step(void* src_device, void* dst_host_mapped, uint64_t bytes, uint64_t clear_from_byte, void* conv_workspace,
                                    uint32_t* error_word, void* tickets_device, void* counter_device, void* word_host_mapped, void* stream) {
    if (!src_device || !dst_host_mapped || !tickets_device || !counter_device || !word_host_mapped || (bytes & 3u) || (clear_from_byte & 3u)) return -1;
    if ((conv_workspace == nullptr) != (error_word == nullptr)) return -1;
    const uint64_t words = bytes / 4;
    if (conv_workspace && words > 65536u) return -1;      // the collected word is copied by the block that collected it
    unsigned blocks = (unsigned)((words + 255) / 256 > 64 ? 64 : (words + 255) / 256);
    if (blocks == 0u || conv_workspace) blocks = 1u;
    hipLaunchKernelGGL(publish_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)src_device, (uint32_t*)dst_host_mapped, words,
                       clear_from_byte / 4, (unsigned int*)conv_workspace, error_word, (uint32_t*)tickets_device, (uint32_t*)counter_device,
                       (volatile uint32_t*)word_host_mapped);
    return (int)hipGetLastError();
}

extern "C" int eae_hip_publish_to_host(const void* src_device, void* dst_host_mapped, uint64_t bytes, void* stream) {
    if (!src_device || !dst_host_mapped || (bytes & 3u)) return -1;
    if (bytes == 0) return 0;
    const uint64_t words = bytes / 4;
    const unsigned blocks = (unsigned)((words + 255) / 256 > 64 ? 64 : (words + 255) / 256);
    hipLaunchKernelGGL(publish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src_device,
                       (uint32_t*)dst_host_mapped, words);
    return (int)hipGetLastError();
}
