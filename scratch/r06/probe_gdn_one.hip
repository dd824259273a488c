#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void one(float x, float a, float* o) {
    const float y0r = __builtin_amdgcn_rsqf(a);
    float g = a * y0r, h = 0.5f * y0r;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, a);
    const float s = __builtin_fmaf(d, h, g);
    float y = h + h;
    o[0] = s; o[1] = y; o[2] = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, y, 1.f);
    const float y1 = __builtin_fmaf(e, y, y);
    const float y2 = __builtin_fmaf(__builtin_fmaf(e, e, e), y, y);
    o[3] = e; o[4] = y1; o[5] = y2;
    float yr = o[2]; yr = __builtin_fmaf(__builtin_fmaf(-s, yr, 1.f), yr, yr); o[6] = yr;
    for (int v = 0; v < 3; v++) {
        const float yy = v == 0 ? y1 : (v == 1 ? y2 : yr);
        float q = x * yy;
        const float r1 = __builtin_fmaf(-s, q, x);
        const float q1 = __builtin_fmaf(r1, yy, q);
        const float r2 = __builtin_fmaf(-s, q1, x);
        const float q2 = __builtin_fmaf(r2, yy, q1);
        o[7 + 5 * v] = q; o[8 + 5 * v] = r1; o[9 + 5 * v] = q1; o[10 + 5 * v] = r2; o[11 + 5 * v] = q2;
    }
    o[22] = x / sqrtf(a); o[23] = sqrtf(a);
}
int main() {
    float* d; float h[24];
    hipMalloc(&d, sizeof h);
    const unsigned int ab = 0x407fffffu; float a; memcpy(&a, &ab, 4);
    hipLaunchKernelGGL(one, dim3(1), dim3(1), 0, 0, 1.0f, a, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"s","y0","rcp(s)","e","y1","y2","yr(after newton)","q","r1","q1","r2","q2","q","r1","q1","r2","q2","q","r1","q1","r2","q2","want q","want s"};
    for (int i = 0; i < 24; i++) { unsigned int b; memcpy(&b, &h[i], 4); printf("%-18s %.10g  0x%08x\n", names[i], h[i], b); }
    return 0;
}
