/*
 * oracle/svhn_oracle.c -- TEST INFRASTRUCTURE ONLY.
 * Plain-C restatement of one dense layer of the reference's numpy SVHN autoencoder
 * (svhn/eae/EntropyAutoencoder.py:239-246, 270-277: numpy.dot(x, W) + tile(b), optionally tls.leaky_relu,
 * svhn/tools/tools.py:692-694), float64. numpy.dot is BLAS (summation order unspecified); this restatement fixes the
 * order to one FMA chain, k ascending from +0, bias added afterwards -- the order the HIP kernel reproduces.
 * Pinned by tests/test_oracle_svhn.py against the reference's own numpy code (tests/golden/svhn_golden.npz) within
 * 1e-12 relative (float64 rounding over K <= 3072 terms).
 */
#include <math.h>
#include <stdint.h>

void orc_dense_f64(const double* x, const double* w, const double* b, double* out, int n, int k, int m, int leaky) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < m; ++c) {
            double acc = 0.0;
            for (int kk = 0; kk < k; ++kk) acc = fma(x[(long)r * k + kk], w[(long)kk * m + c], acc);
            double v = acc + b[c];
            if (leaky && v < 0.0) v = 0.1 * v;
            out[(long)r * m + c] = v;
        }
}
