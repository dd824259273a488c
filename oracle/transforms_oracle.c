/*
 * oracle/transforms_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or called from the product).
 *
 * Plain-C restatement of the analysis / synthesis transforms the reference builds out of TensorFlow ops:
 *   tf.nn.conv2d(..., 'SAME')            kodak_tensorflow/eae/graph/components.py:119-122, 126-129, 133-136
 *   tf.nn.bias_add                       components.py:123, 130, 137-142, 68, 76
 *   tfuls.gdn / tfuls.inverse_gdn        kodak_tensorflow/tfutils/tfutils.py:393-397, 505-509
 *   tf.nn.conv2d_transpose(..., 'SAME')  components.py:63-67, 71-75, 79-83
 * TensorFlow itself is a third-party dependency that is NOT under /root/reference (README.md:12 pins "0.11.0 ...
 * 1.4.0", no lockfile) and is not installable here, so the op semantics follow TF's published definition
 * (SURVEY.md appendix A.2-A.4): cross-correlation, SAME padding pad_before = pad_total / 2, conv2d_transpose = the
 * gradient of conv2d w.r.t. its input.
 *
 * PARITY STATUS: the reference's own tests pin only GDN/IGDN with gamma = 0, beta = 4 (test_tfutils.py:398-423,
 * 493-518), the x16 shape law (test_eae.py:71-139) and "zero latents -> constant image" (test_eae.py:141-176) at this
 * boundary; tests/test_oracle_transforms.py checks those, and cross-checks this file against an independent
 * float64 / torch-CPU evaluation of the same definitions (conv_transpose against autograd of the forward conv).
 * Conv / transposed-conv OUTPUT VALUES of the TF graph are "parity unpinned" by the reference (no golden tensors,
 * no trained weights in the mount: .MISSING_LARGE_BLOBS).
 *
 * Arithmetic: float32 throughout (the TF graph is float32). Each dot product is one fused-multiply-add chain, started
 * from +0, bias added afterwards (tf.nn.bias_add is a separate op), in this fixed order:
 *     input channels in blocks of ORC_CHANNEL_BLOCK = 32 (outer), then kernel row, kernel column, channel within the block.
 * Blocking the reduction over input channels is what cache-friendly convolution code does (the 25 taps then re-use one
 * 128-byte slice of every input pixel); TF's own summation order is unspecified (Eigen contraction). This order is a
 * legitimate instance, and it is the one the gfx950 kernels reproduce bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_CHANNEL_BLOCK 32

/* TF 'SAME' padding before the first element (appendix A.2). */
static int same_pad_before(int in, int k, int s) {
    const int out = (in + s - 1) / s;
    int total = (out - 1) * s + k - in;
    if (total < 0) total = 0;
    return total / 2;
}

/* tf.nn.conv2d(x, w[k,k,cin,cout], strides=[1,s,s,1], 'SAME') (+ bias if non-NULL).
 * x: [n][h][w][cin] float32; out: [n][ceil(h/s)][ceil(w/s)][cout].
 * y[i][j][co] = b[co] + sum_{u,v,ci} x[i*s+u-pb][j*s+v-pb][ci] * w[u][v][ci][co]   (components.py:119-123) */
void orc_conv2d_same(const float* x, int n, int h, int w, int cin, const float* wt, int k, int s, int cout,
                     const float* bias, float* out) {
    const int ho = (h + s - 1) / s, wo = (w + s - 1) / s;
    const int pbh = same_pad_before(h, k, s), pbw = same_pad_before(w, k, s);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < ho; ++i) {
            float* acc = (float*)malloc(sizeof(float) * (size_t)cout);
            for (int j = 0; j < wo; ++j) {
                for (int co = 0; co < cout; ++co) acc[co] = 0.f;
                for (int c0 = 0; c0 < cin; c0 += ORC_CHANNEL_BLOCK) {
                    const int c1 = c0 + ORC_CHANNEL_BLOCK < cin ? c0 + ORC_CHANNEL_BLOCK : cin;
                    for (int u = 0; u < k; ++u) {
                        const int r = i * s + u - pbh;
                        for (int v = 0; v < k; ++v) {
                            const int c = j * s + v - pbw;
                            if (r < 0 || r >= h || c < 0 || c >= w) continue;  /* zero padding contributes nothing */
                            const float* xp = x + (((size_t)b * h + r) * w + c) * cin;
                            const float* wp = wt + ((size_t)(u * k + v) * cin) * cout;
                            for (int ci = c0; ci < c1; ++ci) {
                                const float xv = xp[ci];
                                const float* wrow = wp + (size_t)ci * cout;
                                for (int co = 0; co < cout; ++co) acc[co] = fmaf(xv, wrow[co], acc[co]);
                            }
                        }
                    }
                }
                float* o = out + (((size_t)b * ho + i) * wo + j) * cout;
                for (int co = 0; co < cout; ++co) o[co] = bias ? acc[co] + bias[co] : acc[co];
            }
            free(acc);
        }
}

/* tf.nn.conv2d_transpose(x, w[k,k,cout,cin], output_shape=[n, s*h, s*w, cout], strides=[1,s,s,1], 'SAME')
 * (+ bias if non-NULL): the gradient of the forward conv (whose input is [s*h][s*w][cout]) w.r.t. its input.
 * y[I][J][co] = sum_{u,v,ci : I = p*s+u-pb, J = q*s+v-pb} x[p][q][ci] * w[u][v][co][ci]   (appendix A.3)
 * with pb = SAME pad_before of the forward conv on the OUTPUT size. Chain order: channel block, u ascending, v
 * ascending, ci within the block. */
void orc_conv2d_transpose_same(const float* x, int n, int h, int w, int cin, const float* wt, int k, int s, int cout,
                               const float* bias, float* out) {
    const int ho = h * s, wo = w * s;
    const int pbh = same_pad_before(ho, k, s), pbw = same_pad_before(wo, k, s);
    /* layout-only: [u][v][cout][cin] -> [u][v][cin][cout] so that the inner loop runs over contiguous co */
    float* wp = (float*)malloc(sizeof(float) * (size_t)k * k * cin * cout);
    for (int t = 0; t < k * k; ++t)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) wp[((size_t)t * cin + ci) * cout + co] = wt[((size_t)t * cout + co) * cin + ci];
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int I = 0; I < ho; ++I) {
            float* acc = (float*)malloc(sizeof(float) * (size_t)cout);
            for (int J = 0; J < wo; ++J) {
                for (int co = 0; co < cout; ++co) acc[co] = 0.f;
                for (int c0 = 0; c0 < cin; c0 += ORC_CHANNEL_BLOCK) {
                    const int c1 = c0 + ORC_CHANNEL_BLOCK < cin ? c0 + ORC_CHANNEL_BLOCK : cin;
                    for (int u = 0; u < k; ++u) {
                        const int pn = I + pbh - u;
                        if (pn < 0 || pn % s != 0) continue;
                        const int p = pn / s;
                        if (p >= h) continue;
                        for (int v = 0; v < k; ++v) {
                            const int qn = J + pbw - v;
                            if (qn < 0 || qn % s != 0) continue;
                            const int q = qn / s;
                            if (q >= w) continue;
                            const float* xp = x + (((size_t)b * h + p) * w + q) * cin;
                            const float* wslab = wp + (size_t)(u * k + v) * cin * cout;
                            for (int ci = c0; ci < c1; ++ci) {
                                const float xv = xp[ci];
                                const float* wrow = wslab + (size_t)ci * cout;
                                for (int co = 0; co < cout; ++co) acc[co] = fmaf(xv, wrow[co], acc[co]);
                            }
                        }
                    }
                }
                float* o = out + (((size_t)b * ho + I) * wo + J) * cout;
                for (int co = 0; co < cout; ++co) o[co] = bias ? acc[co] + bias[co] : acc[co];
            }
            free(acc);
        }
    free(wp);
}

/* The same operator in the order of a GEMM + col2im implementation (what im2col-based convolution code does for
 * conv2d_transpose: one matrix product [sites x cin] . [cin x k*k*cout] into a column buffer, then the overlapping
 * patches added into the image site by site). Used for transpose_conv_3 (components.py:79-83: 9x9, stride 4, 128 -> 1),
 * where the fused single chain above forces 7 of every 16 matrix-unit products of the gfx950 kernel to be structural
 * zeros (round 6):
 *     part[p][q][u][v][co] = one fmaf chain over the cin channels, started from +0; the channels of each block of 16 in the order
 *                            0 4 8 12 1 5 9 13 2 6 10 14 3 7 11 15 -- the order in which a matrix unit whose four operand rows are
 *                            fed by 16-byte loads (four consecutive channels per row) takes them; any channels beyond the last
 *                            whole block of 16 ascending
 *     y[I][J][co]          = (((+0 + part(first site)) + part(next site)) + ...)   plain float additions, the
 *                            contributing sites (p, q) in raster order: p ascending, then q ascending
 *                            (u = I + pb - s p and v = J + pb - s q therefore descending), bias added last.
 * Same value as orc_conv2d_transpose_same up to float32 rounding (tests/test_oracle_transforms.py holds both against the
 * float64 definition). */
void orc_conv2d_transpose_same_col2im(const float* x, int n, int h, int w, int cin, const float* wt, int k, int s, int cout,
                                      const float* bias, float* out) {
    const int ho = h * s, wo = w * s;
    const int pbh = same_pad_before(ho, k, s), pbw = same_pad_before(wo, k, s);
    const int kc = k * k * cout;
    /* layout-only: [u][v][cout][cin] -> [cin][u][v][cout] so that the inner loop runs over contiguous (tap, co) */
    float* wp = (float*)malloc(sizeof(float) * (size_t)kc * cin);
    for (int t = 0; t < kc; ++t)
        for (int ci = 0; ci < cin; ++ci) wp[(size_t)ci * kc + t] = wt[(size_t)t * cin + ci];
    /* the column buffer: [b][p][q][u][v][co] */
    float* col = (float*)malloc(sizeof(float) * (size_t)n * h * w * kc);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int p = 0; p < h; ++p)
            for (int q = 0; q < w; ++q) {
                const float* xp = x + (((size_t)b * h + p) * w + q) * cin;
                float* cp = col + (((size_t)b * h + p) * w + q) * kc;
                for (int t = 0; t < kc; ++t) cp[t] = 0.f;
                for (int step = 0; step < cin; ++step) {
                    /* block of 16: step 4 e + r of the block is channel 4 r + e */
                    const int in_block = step & 15;
                    const int ci = (step & ~15) + 16 <= cin ? (step & ~15) + 4 * (in_block & 3) + (in_block >> 2) : step;
                    const float xv = xp[ci];
                    const float* wrow = wp + (size_t)ci * kc;
                    for (int t = 0; t < kc; ++t) cp[t] = fmaf(xv, wrow[t], cp[t]);
                }
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int I = 0; I < ho; ++I)
            for (int J = 0; J < wo; ++J) {
                /* the sites with 0 <= u = I + pb - s p < k: p from ceil((I + pb - k + 1) / s) to floor((I + pb) / s) */
                int p_lo = I + pbh - k + 1, q_lo = J + pbw - k + 1;
                p_lo = p_lo <= 0 ? 0 : (p_lo + s - 1) / s;
                q_lo = q_lo <= 0 ? 0 : (q_lo + s - 1) / s;
                const int p_hi = (I + pbh) / s < h - 1 ? (I + pbh) / s : h - 1;
                const int q_hi = (J + pbw) / s < w - 1 ? (J + pbw) / s : w - 1;
                for (int co = 0; co < cout; ++co) {
                    float acc = 0.f;
                    for (int p = p_lo; p <= p_hi; ++p) {
                        const int u = I + pbh - p * s;
                        for (int q = q_lo; q <= q_hi; ++q) {
                            const int v = J + pbw - q * s;
                            acc = acc + col[(((size_t)b * h + p) * w + q) * kc + (size_t)(u * k + v) * cout + co];
                        }
                    }
                    out[(((size_t)b * ho + I) * wo + J) * cout + co] = bias ? acc + bias[co] : acc;
                }
            }
    free(col);
    free(wp);
}

/* tfuls.gdn (tfutils.py:393-397) / tfuls.inverse_gdn (tfutils.py:505-509) on rows of c channels:
 *   d[c] = (sum_k x[k]^2 * gamma[k][c]) + beta[c];  gdn: x[c] / sqrt(d[c]);  igdn: x[c] * sqrt(d[c]).
 * `matmul(x**2, gamma)` then `+ beta`, then sqrt, then divide / multiply -- in that order, float32. */
void orc_gdn(const float* x, int64_t rows, int c, const float* gamma, const float* beta, int inverse, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * c;
        float* o = out + (size_t)r * c;
        float* d = (float*)malloc(sizeof(float) * (size_t)c);
        for (int j = 0; j < c; ++j) d[j] = 0.f;
        for (int kx = 0; kx < c; ++kx) {
            const float x2 = xr[kx] * xr[kx];
            const float* grow = gamma + (size_t)kx * c;
            for (int j = 0; j < c; ++j) d[j] = fmaf(x2, grow[j], d[j]);
        }
        for (int j = 0; j < c; ++j) {
            const float sq = sqrtf(d[j] + beta[j]);
            o[j] = inverse ? xr[j] * sq : xr[j] / sq;
        }
        free(d);
    }
}

/* uint8 -> float32 with no offset and no scale (eae/batching.py:95). */
void orc_u8_to_f32(const uint8_t* x, int64_t count, float* out) {
    for (int64_t i = 0; i < count; ++i) out[i] = (float)x[i];
}
