// Which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X? Each block records
// (XCC_ID, SE, CU) of the CU it ran on; the host prints the distinct set per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <cstdint>

__global__ void where(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that blocks spread over every CU the queue may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed %d\n", name, (int)e); return; }
    const int n = 8192;
    uint32_t* d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus;
    int per_xcc[8] = {0};
    for (uint32_t v : h) {
        const uint32_t xcc = v >> 16, cu = (v >> 8) & 0xF, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    }
    for (uint32_t c : cus) per_xcc[c >> 12]++;
    printf("%s: %zu distinct CUs; per XCC:", name, cus.size());
    for (int i = 0; i < 8; i++) printf(" %d", per_xcc[i]);
    printf("\n   ");
    int k = 0;
    for (uint32_t c : cus) { if (k++ < 40) printf(" x%u.se%u.cu%u", c >> 12, (c >> 8) & 7, c & 0xF); }
    printf("\n");
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("multiProcessorCount %d\n", prop.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xFFFFFFFFu);
    run("all 256", all);
    std::vector<uint32_t> m(8, 0);
    m[0] = 0xFF; run("bits 0-7", m);
    m.assign(8, 0); for (int i = 0; i < 8; i++) m[i] = 1; run("bits 0,32,..,224", m);
    m.assign(8, 0); m[0] = 0xFFFFFFFFu; run("bits 0-31", m);
    m.assign(8, 0xFFFFFFFFu); m[0] = 0xFFFFFF00u; run("all but bits 0-7", m);
    m.assign(8, 0); m[7] = 0xFF000000u; run("bits 248-255", m);
    return 0;
}
