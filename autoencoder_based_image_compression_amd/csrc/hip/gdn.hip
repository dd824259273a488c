// gdn.hip -- tfuls.gdn / tfuls.inverse_gdn as a standalone op (tfutils.py:363-397, 480-509): rows of 128 channels,
// out = x (/ or *) sqrt(beta + x^2 . gamma). The 128x128 contraction runs on f32 MFMA; used on the path for
// inverse_gdn #4 on the dequantised latents (components.py:53-58).
#include "common.h"

namespace {
// ROWS = 128: every wave 32 rows x all 128 channels. ROWS = 32: the four waves share 32 rows and take a 32-channel tile each -- four
// times as many blocks of a quarter of the work for inputs of fewer than 128 rows per CU (the normalisation pass behind the small-layer
// conv forms of ONE Kodak image: 6,144 rows = 48 blocks of 128 rows, 37 us of mostly latency; 192 blocks of 32). The same per-element
// chain either way (gdn_denominator: k ascending).
template <int ROWS>
__global__ __launch_bounds__(256, 2) void gdn_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int inverse,
                                                     float* __restrict__ out, long rows) {
    constexpr int WM = ROWS / 32, NT = WM;       // waves along rows; 32-channel tiles per wave (4 waves cover ROWS x 128)
    __shared__ __attribute__((aligned(16))) float Xs[ROWS * EAE_XS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, t0 = (wave / WM) * NT;
    const long row0 = (long)blockIdx.x * ROWS;
    // coalesced load of the tile: ROWS rows x 128 floats, float4 per thread per pass
    for (int i = tid; i < ROWS * (EAE_C / 4); i += 256) {
        const int r = i >> 5, q = i & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < rows) v = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * EAE_C + 4 * q);
        float* dst = Xs + r * EAE_XS_STRIDE + 4 * q;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    const int col0 = lane & 31;
    f32x16 d[NT];
    gdn_denominator<NT>(Xs, wm, lane, gamma, t0, d);   // gamma is packed (eae_hip_pack_gamma)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bt = beta[col0 + 32 * (t0 + t)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 32 + acc_row32(r, lane);
            if (row0 + m < rows)
                out[(size_t)(row0 + m) * EAE_C + col0 + 32 * (t0 + t)] =
                    gdn_apply(Xs[m * EAE_XS_STRIDE + col0 + 32 * (t0 + t)], d[t][r], bt, inverse != 0);
        }
    }
}
}  // namespace

extern "C" int eae_hip_gdn(const float* x, const float* gamma, const float* beta, int inverse, float* out,
                           int64_t rows, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (rows < 128L * eae_compute_units()) {      // fewer than a 128-row block per CU
        hipLaunchKernelGGL(gdn_kernel<32>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, inverse, out, (long)rows);
    } else {
        hipLaunchKernelGGL(gdn_kernel<128>, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, inverse, out, (long)rows);
    }
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

// ---- proof harness of common.h's sqrt_mid / div_mid (tests/test_gpu_kernels.py; not part of the path: test build only) ----------
#ifdef EAE_TEST_HOOKS
namespace {
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {       // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// mode 0: every float whose bits lie in [first, last] through sqrt_mid and sqrtf. mode 1: `count` pseudo-random pairs (x, s) of the
// range div_mid is guarded for -- every exponent of the range, random mantissas, both signs of x, plus the extreme mantissas at
// every exponent pair -- through div_mid and `/`. out[0] += mismatches (bit patterns differ and not both NaN), out[1] = an example.
__global__ __launch_bounds__(256) void mid_forms_check_kernel(int mode, unsigned long long first, unsigned long long count,
                                                             unsigned long long seed, unsigned long long* out) {
    unsigned long long bad = 0, example = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        if (mode == 0) {
            const float a = __uint_as_float((unsigned int)(first + i));
            const unsigned int got = __float_as_uint(sqrt_mid(a)), want = __float_as_uint(sqrtf(a));
            const bool both_nan = (got & 0x7FFFFFFFu) > 0x7F800000u && (want & 0x7FFFFFFFu) > 0x7F800000u;
            if (got != want && !both_nan) { bad++; example = first + i; }
        } else {
            const unsigned long long h = mix64(seed + i), h2 = mix64(h);
            // exponents: x in [2^-60, 2^60], s in [2^-20, 2^40] (biased 67..187 and 107..167); the top of each range only with mantissa 0
            unsigned int ex = 67u + (unsigned int)(h % 121u), es = 107u + (unsigned int)((h >> 8) % 61u);
            unsigned int mx = (unsigned int)(h >> 20) & 0x7FFFFFu, ms = (unsigned int)(h2 >> 20) & 0x7FFFFFu;
            const unsigned int kind = (unsigned int)(h2 & 15u);        // a share of the pairs with extreme mantissas
            if (kind == 0) mx = 0; else if (kind == 1) mx = 0x7FFFFFu; else if (kind == 2) ms = 0; else if (kind == 3) ms = 0x7FFFFFu;
            else if (kind == 4) { mx = 0x7FFFFFu; ms = 0x7FFFFFu; } else if (kind == 5) { mx = 0; ms = 0x7FFFFFu; } else if (kind == 6) { mx = 0x7FFFFFu; ms = 0; }
            else if (kind == 7) { mx = ms; }
            if (ex == 187u) mx = 0;
            if (es == 167u) ms = 0;
            const float x = __uint_as_float(((unsigned int)(h2 >> 63) << 31) | (ex << 23) | mx), s = __uint_as_float((es << 23) | ms);
            const unsigned int got = __float_as_uint(div_mid(x, s)), want = __float_as_uint(x / s);
            if (got != want) { bad++; example = ((unsigned long long)__float_as_uint(x) << 32) | __float_as_uint(s); }
        }
    }
    if (bad) { atomicAdd(&out[0], bad); out[1] = example; }
}
}  // namespace

extern "C" int eae_hip_debug_check_mid_forms(int mode, uint64_t first, uint64_t count, uint64_t seed, uint64_t* out2_device, void* stream) {
    if ((mode != 0 && mode != 1) || !out2_device || count == 0) return EAE_HIP_BAD_ARGUMENT;
    if (mode == 0 && first + count > (1ull << 32)) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(mid_forms_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, mode, (unsigned long long)first,
                       (unsigned long long)count, (unsigned long long)seed, reinterpret_cast<unsigned long long*>(out2_device));
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
#endif
