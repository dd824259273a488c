// r06: which physical CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on, on an 8-XCD MI355X in SPX mode?
// Launches enough one-wave blocks to fill the chip on a stream with the first `n` mask bits set (and on one with the complement) and
// prints the set of (XCC, SE, SH, CU) the blocks reported. Build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include <vector>
__global__ void where_kernel(unsigned* out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);      // HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);     // XCC_ID
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static int run(const char* tag, const std::vector<uint32_t>& mask, unsigned* dev, int blocks) {
    hipStream_t s;
    if (mask.empty()) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    CK(hipMemsetAsync(dev, 0xFF, (size_t)blocks * 8, s));
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, s, dev, 200000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> host(2 * blocks);
    CK(hipMemcpy(host.data(), dev, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < blocks; b++) {
        const unsigned hw = host[2 * b], xcc = host[2 * b + 1] & 15u;
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    size_t total = 0;
    std::printf("%s:", tag);
    for (auto& kv : per_xcc) { total += kv.second.size(); std::printf(" xcc%u:%zu", kv.first, kv.second.size()); }
    std::printf("  = %zu distinct CUs\n", total);
    if (total <= 40) for (auto& kv : per_xcc) { std::printf("   xcc%u:", kv.first); for (unsigned c : kv.second) std::printf(" se%u.sh%u.cu%u", c >> 8, (c >> 4) & 1, c & 15); std::printf("\n"); }
    CK(hipStreamDestroy(s));
    return 0;
}
int main(int argc, char** argv) {
    const int blocks = 8192;
    unsigned* dev;
    CK(hipMalloc(&dev, (size_t)blocks * 8));
    if (run("no mask", {}, dev, blocks)) return 1;
    for (int n : {8, 16, 32, 64}) {
        std::vector<uint32_t> low(8, 0u), high(8, 0xFFFFFFFFu);
        for (int i = 0; i < n; i++) { low[i / 32] |= 1u << (i % 32); high[i / 32] &= ~(1u << (i % 32)); }
        char tag[64];
        std::snprintf(tag, sizeof tag, "first %d bits", n);
        if (run(tag, low, dev, blocks)) return 1;
        std::snprintf(tag, sizeof tag, "all but the first %d bits", n);
        if (run(tag, high, dev, blocks)) return 1;
    }
    return 0;
}
