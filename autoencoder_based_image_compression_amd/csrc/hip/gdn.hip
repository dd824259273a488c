// gdn.hip -- tfuls.gdn / tfuls.inverse_gdn as a standalone op (tfutils.py:363-397, 480-509): rows of 128 channels,
// out = x (/ or *) sqrt(beta + x^2 . gamma). The 128x128 contraction runs on f32 MFMA; used on the path for
// inverse_gdn #4 on the dequantised latents (components.py:53-58).
#include "common.h"

namespace {
constexpr int TM = 128;

__global__ __launch_bounds__(256, 2) void gdn_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int inverse,
                                                     float* __restrict__ out, long rows) {
    __shared__ __attribute__((aligned(16))) float Xs[TM * EAE_XS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const long row0 = (long)blockIdx.x * TM;
    // coalesced load of the tile: 128 rows x 128 floats, float4 per thread per pass
    for (int i = tid; i < TM * (EAE_C / 4); i += 256) {
        const int r = i >> 5, q = i & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < rows) v = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * EAE_C + 4 * q);
        float* dst = Xs + r * EAE_XS_STRIDE + 4 * q;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    const int col0 = lane & 31;
    f32x16 d[4];
    gdn_denominator<4>(Xs, wm, lane, gamma, 0, d);   // gamma is packed (eae_hip_pack_gamma)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bt = beta[col0 + 32 * t];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            if (row0 + m < rows)
                out[(size_t)(row0 + m) * EAE_C + col0 + 32 * t] =
                    gdn_apply(Xs[m * EAE_XS_STRIDE + col0 + 32 * t], d[t][r], bt, inverse != 0);
        }
    }
}
}  // namespace

extern "C" int eae_hip_gdn(const float* x, const float* gamma, const float* beta, int inverse, float* out,
                           int64_t rows, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0) return EAE_HIP_BAD_ARGUMENT;
    const long grid = (rows + TM - 1) / TM;
    hipLaunchKernelGGL(gdn_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, inverse, out,
                       (long)rows);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
