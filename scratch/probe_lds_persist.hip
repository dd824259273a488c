// Does a long-lived wave keep its LDS while MFMA-heavy workgroups come and go on the same CUs?  (round 3: the decoder core's
// ring seemed to lose its content under load.)  48 blocks of 64 threads, dynamic LDS as the decoder core asks for; every lane
// writes a pattern into its column, idles for ~2 ms in steps, re-reads the column after every step and counts what changed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ unsigned int lds[];

__global__ __launch_bounds__(64) void hold_kernel(unsigned int* bad, int words, int rounds) {
    const unsigned lane = threadIdx.x;
    for (int w = 0; w < words; ++w) lds[w * 64 + lane] = 0x9E3779B9u * (blockIdx.x * 64 + lane + 1) + w;
    unsigned int wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
        for (int w = 0; w < words; ++w) wrong += lds[w * 64 + lane] != 0x9E3779B9u * (blockIdx.x * 64 + lane + 1) + w;
    }
    bad[blockIdx.x * 64 + lane] = wrong;
}

// an LDS- and MFMA-heavy neighbour: fills its 38 KB of LDS with junk over and over
__global__ __launch_bounds__(256, 3) void noisy_kernel(float* out, int iters) {
    __shared__ float junk[9472];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc = {0};
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 9472; i += 256) junk[i] = (float)(it + i);
        __syncthreads();
        const float a = junk[(threadIdx.x * 7 + it) % 9472];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, (float)k, acc, 0, 0, 0);
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0];
}

int main() {
    unsigned int* bad;
    float* out;
    hipMalloc(&bad, 48 * 64 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    hipStream_t a, b;
    hipStreamCreate(&a);
    hipStreamCreate(&b);
    for (int load = 0; load < 2; ++load) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(bad, 0xFF, 48 * 64 * 4);
            hipDeviceSynchronize();
            if (load) for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(noisy_kernel, dim3(4096), dim3(256), 0, a, out, 200);
            hipLaunchKernelGGL(hold_kernel, dim3(48), dim3(64), 13824, b, bad, 54, 40);
            hipDeviceSynchronize();
            std::vector<unsigned int> h(48 * 64);
            hipMemcpy(h.data(), bad, h.size() * 4, hipMemcpyDeviceToHost);
            unsigned long total = 0; int lanes = 0;
            for (auto v : h) { total += v; lanes += v != 0; }
            printf("load %d rep %d: %lu changed words seen by %d lanes (%s)\n", load, rep, total, lanes, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
