// quantize.hip -- the uniform quantiser and everything the coder / rate estimate need from it, in one pass over the
// latents (HBM-bound: 4 B read + up to 10 B written per latent value):
//   centring                     reconstructing_eae_kodak.py:170-178
//   tls.quantize_per_map         tools/tools.py:927-929        bw * round_half_even(x / bw), float32
//   tls.cast_float_to_int16      tools/tools.py:126-133 on cq / bw (lossless/compression.py:142)
//   de-centring                  reconstructing_eae_kodak.py:192
//   tls.count_nb_deads           tools/tools.py:318-320        (per-map "any non-zero" flags)
// plus the per-map symbol histograms behind tls.count_symbols / discrete_entropy / rate_3d (tools.py:376-388,
// 523-537, 977-989) and lossless/stats.py:181-195, as a second kernel over the planar symbols.
#include "common.h"

namespace {
constexpr int PIX = 64;   // pixels per block

__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ y, const float* __restrict__ map_mean,
                                                       const float* __restrict__ bin_widths, float* __restrict__ cq_out,
                                                       float* __restrict__ shifted_out, int16_t* __restrict__ symbols,
                                                       unsigned int* __restrict__ nonzero, unsigned int* checks,
                                                       int hw, int chunks) {
    __shared__ int16_t tile[EAE_C][PIX + 2];
    const int tid = threadIdx.x;
    const int c = tid & 127, half = tid >> 7;
    const int img = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const float m = map_mean ? map_mean[c] : 0.f;
    const float bw = bin_widths[c];
    bool any_nonzero = false;
    unsigned int bad = 0, not_quantized = 0, altered = 0;
    const int p0 = chunk * PIX + half * (PIX / 2);
#pragma unroll 4
    for (int i = 0; i < PIX / 2; ++i) {
        const int pix = p0 + i;
        if (pix < hw) {
            const size_t idx = ((size_t)img * hw + pix) * EAE_C + c;
            const float yin = y[idx];
            const float centered = yin - m;
            const float r = round_half_even(centered / bw);
            const float cq = bw * r;
            // compression.py:142: symbols come from cq / bw again (not from r), rounded half to even
            const float rs = round_half_even(cq / bw);
            if (!(fabsf(rs) < 32768.f)) bad++;              // tools.py:130-132 (AssertionError in the reference)
            // tools.py:372-375 on an input that claims to be quantised already: |bw*round(x/bw) - x| < 1.5e-10
            if (!(fabs((double)cq - (double)centered) < 1.5e-10)) not_quantized++;
            // compression.py:149-153: int16 symbol * bw must give the input back exactly
            if (!((float)(int16_t)(int)rs * bw == centered)) altered++;
            if (cq_out) cq_out[idx] = cq;
            if (shifted_out) shifted_out[idx] = cq + m;
            if (cq != 0.f) any_nonzero = true;              // NaN counts as non-zero, like sum(abs(.)) == 0 failing
            tile[c][half * (PIX / 2) + i] = (int16_t)(int)rs;
        }
    }
    if (any_nonzero && nonzero) nonzero[img * EAE_C + c] = 1u;   // benign race: every writer stores 1
    if (checks) {
        if (bad) atomicAdd(&checks[0], bad);
        if (not_quantized) atomicAdd(&checks[1], not_quantized);
        if (altered) atomicAdd(&checks[2], altered);
    }
    if (symbols) {
        __syncthreads();
        // planar write: a wave covers 64 consecutive pixels of one map (128 B)
        for (int i = tid; i < EAE_C * PIX; i += 256) {
            const int ch = i >> 6, px = i & 63;
            const int pix = chunk * PIX + px;
            if (pix < hw) symbols[((size_t)img * EAE_C + ch) * hw + pix] = tile[ch][px];
        }
    }
}

// Any channel count (the tools.* helpers accept arbitrary arrays, e.g. one map with a scalar bin width): one element per
// thread, same arithmetic as above.
__global__ void quantize_generic_kernel(const float* __restrict__ y, const float* __restrict__ map_mean,
                                        const float* __restrict__ bin_widths, float* __restrict__ cq_out,
                                        float* __restrict__ shifted_out, int16_t* __restrict__ symbols,
                                        unsigned int* __restrict__ nonzero, unsigned int* checks, long total, int hw,
                                        int c_count) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c_count);
        const long pixel = idx / c_count;
        const long img = pixel / hw;
        const int pix = (int)(pixel % hw);
        const float m = map_mean ? map_mean[c] : 0.f;
        const float bw = bin_widths[c];
        const float centered = y[idx] - m;
        const float r = round_half_even(centered / bw);
        const float cq = bw * r;
        const float rs = round_half_even(cq / bw);
        if (checks) {
            if (!(fabsf(rs) < 32768.f)) atomicAdd(&checks[0], 1u);
            if (!(fabs((double)cq - (double)centered) < 1.5e-10)) atomicAdd(&checks[1], 1u);
            if (!((float)(int16_t)(int)rs * bw == centered)) atomicAdd(&checks[2], 1u);
        }
        if (cq_out) cq_out[idx] = cq;
        if (shifted_out) shifted_out[idx] = cq + m;
        if (nonzero && cq != 0.f) nonzero[img * c_count + c] = 1u;
        if (symbols) symbols[((size_t)img * c_count + c) * hw + pix] = (int16_t)(int)rs;
    }
}

// tls.count_nb_deads (tools.py:318-320) on an arbitrary stack [N][hw][C]: flag (n, c) when some element is non-zero.
__global__ void nonzero_flags_kernel(const float* __restrict__ x, unsigned int* __restrict__ nonzero, long total, int hw,
                                     int c_count) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        if (x[idx] != 0.f) nonzero[(idx / c_count / hw) * c_count + idx % c_count] = 1u;
    }
}

// tls.cast_float_to_int16 (tools.py:126-133): int16(round_half_even(x)); range_error counts |round(x)| >= 32768.
__global__ void cast_int16_kernel(const float* __restrict__ x, int16_t* __restrict__ out, long count,
                                  unsigned int* range_error) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float r = round_half_even(x[i]);
        if (!(fabsf(r) < 32768.f)) atomicAdd(range_error, 1u);
        out[i] = (int16_t)(int)r;
    }
}

// Per-map means over all rows, EXACTLY lossless/stats.py:306 `numpy.mean(y_float32, axis=(0, 1, 2))`: numpy reduces the
// leading axes of a C-contiguous array row by row into a float32 accumulator per map (out[c] += y[row][c], rows ascending;
// no pairwise summation, that only applies along a contiguous reduction axis) and then divides by the float32 row count.
// One thread per map runs that chain; consecutive threads read consecutive floats, so every row is one coalesced line.
// (The mean feeds `y - mean` before the quantiser: one ulp of difference can flip a symbol relative to statistics the
// reference saved, so this is arithmetic to reproduce, not to improve.)
__global__ __launch_bounds__(64) void map_means_kernel(const float* __restrict__ y, float* __restrict__ means, long rows,
                                                       int c_count) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= c_count) return;
    float acc = 0.f;
    const float* p = y + c;
#pragma unroll 8
    for (long r = 0; r < rows; ++r) acc = acc + p[r * c_count];
    means[c] = acc / (float)rows;
}

// One block per map. LDS histogram when the bins fit, global atomics otherwise (caller zeroed hist/overflow).
constexpr int LDS_BINS = 8192;
__global__ __launch_bounds__(256) void hist_kernel(const int16_t* __restrict__ symbols, unsigned int* __restrict__ hist,
                                                   int radius, unsigned int* __restrict__ overflow, int map_size,
                                                   long first_map, long map_step) {
    __shared__ unsigned int bins[LDS_BINS];
    const int nb = 2 * radius + 1;
    const int16_t* s = symbols + (size_t)(first_map + (long)blockIdx.x * map_step) * map_size;
    unsigned int* h = hist + (size_t)blockIdx.x * nb;
    unsigned int over = 0;
    if (nb <= LDS_BINS) {
        for (int i = threadIdx.x; i < nb; i += 256) bins[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < map_size; i += 256) {
            const int v = (int)s[i] + radius;
            if (v >= 0 && v < nb) atomicAdd(&bins[v], 1u); else over++;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256) if (bins[i]) h[i] += bins[i];   // this block owns the row
    } else {
        for (int i = threadIdx.x; i < map_size; i += 256) {
            const int v = (int)s[i] + radius;
            if (v >= 0 && v < nb) atomicAdd(&h[v], 1u); else over++;
        }
    }
    if (over) atomicAdd(&overflow[blockIdx.x], over);
}

__global__ void cast_bt601_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long count) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        out[i] = (uint8_t)(unsigned int)round_half_even(fminf(fmaxf(x[i], 16.f), 235.f));
}

__global__ __launch_bounds__(256) void sse_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                  unsigned long long* sse, long pixels, int blocks_per_image) {
    const int img = blockIdx.x / blocks_per_image, part = blockIdx.x % blocks_per_image;
    const uint8_t* pa = a + (size_t)img * pixels;
    const uint8_t* pb = b + (size_t)img * pixels;
    unsigned long long s = 0;
    for (long i = (long)part * 256 + threadIdx.x; i < pixels; i += (long)blocks_per_image * 256) {
        const int d = (int)pa[i] - (int)pb[i];
        s += (unsigned long long)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sse[img], red[0] + red[1] + red[2] + red[3]);
}
}  // namespace

static unsigned grid_for(long total) {
    const long blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

extern "C" int eae_hip_quantize_maps(const float* y, const float* map_mean, const float* bin_widths, float* cq_out,
                                     float* shifted_out, int16_t* symbols_planar, uint32_t* nonzero_flags,
                                     uint32_t* checks, int n, int hw, int c, void* stream) {
    if (!y || !bin_widths || n <= 0 || hw <= 0 || c <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (c == EAE_C) {
        const int chunks = (hw + PIX - 1) / PIX;
        hipLaunchKernelGGL(quantize_kernel, dim3(n * chunks), dim3(256), 0, (hipStream_t)stream, y, map_mean, bin_widths,
                           cq_out, shifted_out, symbols_planar, nonzero_flags, checks, hw, chunks);
    } else {
        const long total = (long)n * hw * c;
        hipLaunchKernelGGL(quantize_generic_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, map_mean,
                           bin_widths, cq_out, shifted_out, symbols_planar, nonzero_flags, checks, total, hw, c);
    }
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_nonzero_flags(const float* x, uint32_t* nonzero_flags, int n, int hw, int c, void* stream) {
    if (!x || !nonzero_flags || n <= 0 || hw <= 0 || c <= 0) return EAE_HIP_BAD_ARGUMENT;
    const long total = (long)n * hw * c;
    hipLaunchKernelGGL(nonzero_flags_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, nonzero_flags, total,
                       hw, c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

// ---- statistics for lossless/stats.py:197-241 (find_index_map_exception) ---------------------------------------------
// Per-map minimum / maximum (numpy.amin / amax of stats.py:103-104) through order-preserving uint keys and integer atomics.
__device__ __forceinline__ unsigned int float_key(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__global__ void minmax_init_kernel(unsigned int* __restrict__ keys, int c_count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < c_count) { keys[c] = 0xFFFFFFFFu; keys[c_count + c] = 0u; }
}
__global__ __launch_bounds__(256) void map_minmax_kernel(const float* __restrict__ y, unsigned int* __restrict__ keys, long rows,
                                                         int c_count) {
    const int c = threadIdx.x % c_count;
    const int lanes_per_c = 256 / c_count;
    const int sub = threadIdx.x / c_count;
    if (sub >= lanes_per_c) return;
    unsigned int lo = 0xFFFFFFFFu, hi = 0u;
    for (long r = (long)blockIdx.x * lanes_per_c + sub; r < rows; r += (long)gridDim.x * lanes_per_c) {
        const unsigned int k = float_key(y[r * c_count + c]);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    atomicMin(&keys[c], lo);
    atomicMax(&keys[c_count + c], hi);
}
__global__ void minmax_final_kernel(const unsigned int* __restrict__ keys, float* __restrict__ out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = key_float(keys[i]);
}
// Per-map histogram of floor(y) (the unit-width intervals of compute_probabilities_intervals, stats.py:70-134):
// hist[c][floor(y) + radius] += 1; values outside [-radius, radius] are counted in overflow[c].
__global__ __launch_bounds__(256) void floor_hist_kernel(const float* __restrict__ y, unsigned int* __restrict__ hist, int radius,
                                                         unsigned int* __restrict__ overflow, long rows, int c_count) {
    const int c = threadIdx.x % c_count;
    const int lanes_per_c = 256 / c_count;
    const int sub = threadIdx.x / c_count;
    if (sub >= lanes_per_c) return;
    const int nb = 2 * radius + 1;
    unsigned int over = 0;
    for (long r = (long)blockIdx.x * lanes_per_c + sub; r < rows; r += (long)gridDim.x * lanes_per_c) {
        const float f = floorf(y[r * c_count + c]);
        if (f >= (float)-radius && f <= (float)radius) atomicAdd(&hist[(size_t)c * nb + ((int)f + radius)], 1u);
        else over++;                                                   // also NaN
    }
    if (over) atomicAdd(&overflow[c], over);
}

// tls.rgb_to_ycbcr (tools.py:1019-1083): ITU-R BT.601 in float64, terms added left to right as numpy evaluates them, clip to
// [0, 255], round half to even, uint8. The constants are the same IEEE doubles Python forms (65.481/255. etc.).
__global__ void rgb_to_ycbcr_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ ycbcr, uint8_t* __restrict__ luma,
                                    long count) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const double r = (double)rgb[3 * i], g = (double)rgb[3 * i + 1], b = (double)rgb[3 * i + 2];
        const double y = 16. + (65.481 / 255.) * r + (128.553 / 255.) * g + (24.966 / 255.) * b;
        const double cb = 128. - (37.797 / 255.) * r - (74.203 / 255.) * g + (112. / 255.) * b;
        const double cr = 128. + (112. / 255.) * r - (93.786 / 255.) * g - (18.214 / 255.) * b;
        const uint8_t y8 = (uint8_t)rint(fmin(fmax(y, 0.), 255.));
        if (luma) luma[i] = y8;
        if (ycbcr) {
            ycbcr[3 * i] = y8;
            ycbcr[3 * i + 1] = (uint8_t)rint(fmin(fmax(cb, 0.), 255.));
            ycbcr[3 * i + 2] = (uint8_t)rint(fmin(fmax(cr, 0.), 255.));
        }
    }
}

extern "C" int eae_hip_map_means(const float* y, float* means, int64_t rows, int c, void* stream) {
    if (!y || !means || rows <= 0 || c <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(map_means_kernel, dim3((unsigned)((c + 63) / 64)), dim3(64), 0, (hipStream_t)stream, y, means, (long)rows, c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_cast_int16(const float* x, int16_t* out, int64_t count, uint32_t* range_error, void* stream) {
    if (!x || !out || !range_error || count <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(cast_int16_kernel, dim3(grid_for((long)count)), dim3(256), 0, (hipStream_t)stream, x, out, (long)count,
                       range_error);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_symbol_histograms(const int16_t* symbols_planar, uint32_t* hist, int hist_radius,
                                         uint32_t* overflow, int n_maps, int map_size, void* stream) {
    if (!symbols_planar || !hist || !overflow || n_maps <= 0 || map_size <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (hist_radius < 0 || hist_radius > 32768) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(hist_kernel, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, symbols_planar, hist, hist_radius,
                       overflow, map_size, 0L, 1L);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_symbol_histograms_strided(const int16_t* symbols_planar, uint32_t* hist, int hist_radius,
                                                 uint32_t* overflow, int n_maps, int map_size, int64_t first_map,
                                                 int64_t map_step, void* stream) {
    if (!symbols_planar || !hist || !overflow || n_maps <= 0 || map_size <= 0 || first_map < 0 || map_step <= 0) return EAE_HIP_BAD_ARGUMENT;
    if (hist_radius < 0 || hist_radius > 32768) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(hist_kernel, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, symbols_planar, hist, hist_radius,
                       overflow, map_size, (long)first_map, (long)map_step);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_cast_bt601(const float* x, uint8_t* out, int64_t count, void* stream) {
    if (!x || !out || count <= 0) return EAE_HIP_BAD_ARGUMENT;
    const long blocks = (count + 255) / 256;
    hipLaunchKernelGGL(cast_bt601_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, x,
                       out, (long)count);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_sse_u8(const uint8_t* a, const uint8_t* b, uint64_t* sse, int n, int64_t pixels_per_image,
                              void* stream) {
    if (!a || !b || !sse || n <= 0 || pixels_per_image <= 0) return EAE_HIP_BAD_ARGUMENT;
    long bpi = (pixels_per_image + 4095) / 4096;
    if (bpi > 64) bpi = 64;
    hipLaunchKernelGGL(sse_kernel, dim3((unsigned)(n * bpi)), dim3(256), 0, (hipStream_t)stream, a, b,
                       reinterpret_cast<unsigned long long*>(sse), (long)pixels_per_image, (int)bpi);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_map_minmax(const float* y, float* minmax, uint32_t* scratch_keys, int64_t rows, int c, void* stream) {
    if (!y || !minmax || !scratch_keys || rows <= 0 || c <= 0 || c > 256) return EAE_HIP_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const long per_block = 256 / c;
    long blocks = (rows + per_block * 64 - 1) / (per_block * 64);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, scratch_keys, c);
    hipLaunchKernelGGL(map_minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, scratch_keys, (long)rows, c);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(2), dim3(256), 0, s, scratch_keys, minmax, 2 * c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_floor_histograms(const float* y, uint32_t* hist, int radius, uint32_t* overflow, int64_t rows, int c,
                                        void* stream) {
    if (!y || !hist || !overflow || rows <= 0 || c <= 0 || c > 256 || radius < 0 || radius > (1 << 20)) return EAE_HIP_BAD_ARGUMENT;
    const long per_block = 256 / c;
    long blocks = (rows + per_block * 64 - 1) / (per_block * 64);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(floor_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, hist, radius, overflow,
                       (long)rows, c);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}

extern "C" int eae_hip_rgb_to_ycbcr(const uint8_t* rgb, uint8_t* ycbcr, uint8_t* luma, int64_t count, void* stream) {
    if (!rgb || (!ycbcr && !luma) || count <= 0) return EAE_HIP_BAD_ARGUMENT;
    hipLaunchKernelGGL(rgb_to_ycbcr_kernel, dim3(grid_for((long)count)), dim3(256), 0, (hipStream_t)stream, rgb, ycbcr, luma,
                       (long)count);
    EAE_HIP_CHECK_LAUNCH();
    return EAE_HIP_OK;
}
