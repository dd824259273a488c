// svhn.hip -- BASELINE.json configs[0]: the reference's didactic SVHN entropy autoencoder (svhn/eae/EntropyAutoencoder.py),
// a FLOAT64 fully connected 3072 -> 300 -> 200 -> 300 -> 3072 network with LeakyReLU(0.1):
//   encoder  svhn/eae/EntropyAutoencoder.py:218-247     y = LReLU(x W1 + b1) W2 + b2
//   decoder  :249-278                                    x^ = LReLU(q W3 + b3) W4 + b4
// plus the element-wise steps of svhn/eae/utils.py:54-74: preprocess_svhn (svhn/svhn/svhn.py:210), tls.quantization
// (svhn/tools/tools.py:1095), de-normalisation + tls.cast_float_to_uint8 (:166) and the squared error of tls.mean_psnr
// (:857-859). 3.9 MFLOP per image: this path is tiny, what matters is float64 and a reproducible order:
// every dot product is ONE float64 FMA chain with k ascending from +0, bias added afterwards -- the order
// oracle/svhn_oracle.c runs, so GPU == oracle exactly (numpy.dot's own BLAS order is unspecified; tests bound the
// distance to it).
#include "common.h"

namespace {
constexpr int ROWS = 8;     // rows of x per block
constexpr int KCHUNK = 256; // k values staged in LDS per pass

// out[n][m] = act( (sum_k x[n][k] * w[k][m]) + b[m] ),  act = identity or LeakyReLU(0.1) (tools.py:692-694: 0.1 * v for v < 0)
__global__ __launch_bounds__(256) void dense_f64_kernel(const double* __restrict__ x, const double* __restrict__ w,
                                                        const double* __restrict__ b, double* __restrict__ out,
                                                        int n, int k, int m, int leaky) {
    __shared__ double xs[ROWS][KCHUNK];
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int row0 = blockIdx.y * ROWS;
    double acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.0;
    for (int k0 = 0; k0 < k; k0 += KCHUNK) {
        __syncthreads();
        for (int i = threadIdx.x; i < ROWS * KCHUNK; i += 256) {
            const int r = i / KCHUNK, kk = i % KCHUNK;
            xs[r][kk] = (row0 + r < n && k0 + kk < k) ? x[(size_t)(row0 + r) * k + k0 + kk] : 0.0;
        }
        __syncthreads();
        if (col < m) {
            const int kend = (k - k0) < KCHUNK ? (k - k0) : KCHUNK;
            for (int kk = 0; kk < kend; ++kk) {
                const double wv = w[(size_t)(k0 + kk) * m + col];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) acc[r] = fma(xs[r][kk], wv, acc[r]);
            }
        }
    }
    if (col < m) {
        const double bv = b[col];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (row0 + r < n) {
                double v = acc[r] + bv;
                if (leaky && v < 0.0) v = 0.1 * v;
                out[(size_t)(row0 + r) * m + col] = v;
            }
        }
    }
}

// (uint8 - mean[j]) / std   (svhn.py:210)
__global__ void preprocess_kernel(const uint8_t* __restrict__ u8, const double* __restrict__ mean, double std_training,
                                  double* __restrict__ out, long total, int d) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[i] = ((double)u8[i] - mean[i % d]) / std_training;
}

// q = bw * round_half_even(y / bw) (tools.py:1095); symbol = round(q / bw) for the histogram behind discrete_entropy;
// checks[0] += |symbol| >= 2^31 (outside this build's domain), checks[1] += |quantization(y) - y| >= 1.5e-10.
__global__ void quantize_f64_kernel(const double* __restrict__ y, double bw, double* __restrict__ q,
                                    int* __restrict__ symbols, unsigned int* checks, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const double qq = bw * rint(y[i] / bw);
        if (q) q[i] = qq;
        const double s = rint(qq / bw);
        if (!(fabs(s) < 2147483648.0)) atomicAdd(&checks[0], 1u);
        // tools.py:214-217 when y is passed as already quantised: |quantization(y) - y| < 1.5e-10
        if (!(fabs(qq - y[i]) < 1.5e-10)) atomicAdd(&checks[1], 1u);
        if (symbols) symbols[i] = (int)s;
    }
}

// histogram of int32 symbols over [lo, lo + nb): global atomics (tiny inputs); overflow counts the rest
__global__ void hist_i32_kernel(const int* __restrict__ symbols, long total, int lo, int nb, unsigned int* __restrict__ hist,
                                unsigned int* overflow) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = (long)symbols[i] - lo;
        if (v >= 0 && v < nb) atomicAdd(&hist[v], 1u); else atomicAdd(overflow, 1u);
    }
}

__global__ void minmax_i32_kernel(const int* __restrict__ symbols, long total, int* minmax) {
    int lo = 2147483647, hi = -2147483647 - 1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int v = symbols[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&minmax[0], lo); atomicMax(&minmax[1], hi); }
}

// rec * std + mean[j] -> clip [0, 255] -> round half even -> uint8 (utils.py:71-73, tools.py:166); squared error per image
__global__ __launch_bounds__(256) void postprocess_kernel(const double* __restrict__ rec, double std_training,
                                                          const double* __restrict__ mean, uint8_t* __restrict__ out,
                                                          const uint8_t* __restrict__ ref, unsigned long long* sse, int d) {
    const int img = blockIdx.x;
    unsigned long long s = 0;
    for (int j = threadIdx.x; j < d; j += 256) {
        const size_t i = (size_t)img * d + j;
        const double v = rec[i] * std_training + mean[j];
        const double c = fmin(fmax(v, 0.0), 255.0);
        const unsigned int u = (unsigned int)rint(c);
        out[i] = (uint8_t)u;
        if (ref) {
            const int e = (int)ref[i] - (int)u;
            s += (unsigned long long)(e * e);
        }
    }
    if (ref && sse) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        __shared__ unsigned long long red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) sse[img] = red[0] + red[1] + red[2] + red[3];
    }
}

unsigned grid_for(long total) {
    const long blocks = (total + 255) / 256;
    return (unsigned)(blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks));
}
}  // namespace

extern "C" int eae_hip_svhn_dense_f64(const double* x, const double* w, const double* b, double* out, int n, int k, int m,
                                      int leaky_relu, void* stream) {
    if (!x || !w || !b || !out || n <= 0 || k <= 0 || m <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(dense_f64_kernel, dim3((m + 255) / 256, (n + ROWS - 1) / ROWS), dim3(256), 0, (hipStream_t)stream, x, w, b,
                       out, n, k, m, leaky_relu);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_svhn_preprocess(const uint8_t* images, const double* mean, double std_training, double* out, int n,
                                       int d, void* stream) {
    if (!images || !mean || !out || n <= 0 || d <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long)n * d)), dim3(256), 0, (hipStream_t)stream, images, mean,
                       std_training, out, (long)n * d, d);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_svhn_quantize_f64(const double* y, double bin_width, double* q, int32_t* symbols, uint32_t* checks,
                                         int64_t count, void* stream) {
    if (!y || !checks || count <= 0 || !(bin_width > 0.0)) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(quantize_f64_kernel, dim3(grid_for((long)count)), dim3(256), 0, (hipStream_t)stream, y, bin_width, q,
                       symbols, checks, (long)count);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_svhn_symbol_range(const int32_t* symbols, int64_t count, int32_t* minmax, void* stream) {
    if (!symbols || !minmax || count <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(minmax_i32_kernel, dim3(grid_for((long)count)), dim3(256), 0, (hipStream_t)stream, symbols, (long)count,
                       minmax);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_svhn_symbol_histogram(const int32_t* symbols, int64_t count, int32_t lowest, int32_t nb_bins,
                                             uint32_t* hist, uint32_t* overflow, void* stream) {
    if (!symbols || !hist || !overflow || count <= 0 || nb_bins <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(hist_i32_kernel, dim3(grid_for((long)count)), dim3(256), 0, (hipStream_t)stream, symbols, (long)count,
                       lowest, nb_bins, hist, overflow);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_svhn_postprocess(const double* reconstruction, double std_training, const double* mean,
                                        uint8_t* out_u8, const uint8_t* ref_u8, uint64_t* sse, int n, int d, void* stream) {
    if (!reconstruction || !mean || !out_u8 || n <= 0 || d <= 0) return EAE_HIP_BAD_ARGUMENT;
    if ((ref_u8 != nullptr) != (sse != nullptr)) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(postprocess_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, reconstruction, std_training, mean,
                       out_u8, ref_u8, reinterpret_cast<unsigned long long*>(sse), d);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
