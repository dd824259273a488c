// Are two kernels launched back to back into ONE stream really ordered on this box?  A: one block, idles ~3 ms, then sets a flag.
// B: many blocks, reads the flag at once.  With stream order B must see 1, every time.  Tried on a plain stream, a non-blocking one
// and a high-priority non-blocking one (what torch hands out), with and without a busy neighbour stream, GPU_MAX_HW_QUEUES as given.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void slow_setter(unsigned int* flag) {
    for (int i = 0; i < 3000; ++i) __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) *flag = 1u;
}
__global__ void reader(const unsigned int* flag, unsigned int* seen) {
    if (threadIdx.x == 0) seen[blockIdx.x] = *flag;
}
__global__ __launch_bounds__(256) void busy(float* out, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
    unsigned int *flag, *seen;
    float* out;
    hipMalloc(&flag, 4); hipMalloc(&seen, 3072 * 4); hipMalloc(&out, 4096 * 256 * 4);
    hipStream_t plain, nonblocking, prio, other;
    hipStreamCreate(&plain);
    hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&prio, hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
    const char* names[3] = {"plain", "non-blocking", "non-blocking high priority"};
    hipStream_t streams[3] = {plain, nonblocking, prio};
    for (int load = 0; load < 2; ++load)
        for (int k = 0; k < 3; ++k) {
            int early = 0;
            for (int rep = 0; rep < 20; ++rep) {
                hipMemsetAsync(flag, 0, 4, streams[k]);
                hipMemsetAsync(seen, 0xFF, 3072 * 4, streams[k]);
                if (load) hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, other, out, 20000);
                hipLaunchKernelGGL(slow_setter, dim3(48), dim3(64), 0, streams[k], flag);
                hipLaunchKernelGGL(reader, dim3(3072), dim3(64), 0, streams[k], flag, seen);
                hipDeviceSynchronize();
                unsigned int h[3072];
                hipMemcpy(h, seen, sizeof(h), hipMemcpyDeviceToHost);
                for (int i = 0; i < 3072; ++i) early += h[i] != 1u;
            }
            printf("GPU_MAX_HW_QUEUES=%s load=%d stream=%s: reader blocks that ran before the setter finished: %d of %d\n",
                   getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset", load, names[k], early, 20 * 3072);
        }
    return 0;
}
