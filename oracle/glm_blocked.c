/*
 * "B2" CPU baseline: the same population ll + gradient as
 * oracle/glm_oracle.c, restructured the way a tuned CPU implementation would be --
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY; nothing under theano_pyglm_amd/ links or calls it.
 *
 * Reference arithmetic (pyglm/glm.py:31-52, impulse.py:58, coord_descent.py:27-30) for ALL
 * post-synaptic neurons in one sweep over the materialised features fS (impulse.py:114-130):
 *   per time tile of TT rows:  X = F.Wmat  (TT x K times K x N)   -> rate epilogue -> r
 *                              G += F^T r                          (K x N)
 * so fS is streamed from DRAM once per evaluation instead of once per neuron (the reference's
 * per-neuron Theano calls read it N times), with OpenMP over time blocks and per-thread G
 * accumulators reduced in thread order.  Parity with glm_oracle.c is checked in tests/test_oracle.py.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TT 32

void oracle_ll_grad_blocked(const uint8_t* S, int64_t nT, int N, int B, const double* fS,
                            const double* fstim, int Dstim, const double* theta, const double* Weff,
                            int nlin, double dt, double* ll_out, double* grad_out, int threads)
{
    const int K = N * B, KT = K + Dstim, P = 1 + Dstim + K;
    double* Wmat = (double*)malloc(sizeof(double) * (size_t)KT * N);
    double* bias = (double*)malloc(sizeof(double) * N);
    for (int n = 0; n < N; ++n) {
        bias[n] = theta[(size_t)n * P];
        for (int k = 0; k < K; ++k)
            Wmat[(size_t)k * N + n] = theta[(size_t)n * P + 1 + Dstim + k] * Weff[(size_t)(k / B) * N + n];
        for (int j = 0; j < Dstim; ++j) Wmat[(size_t)(K + j) * N + n] = theta[(size_t)n * P + 1 + j];
    }
    if (threads < 1) threads = 1;
    const int64_t ntile = (nT + TT - 1) / TT;
    double* Gall = grad_out ? (double*)calloc((size_t)threads * KT * N, sizeof(double)) : NULL;
    double* llall = (double*)calloc((size_t)threads * 2 * N, sizeof(double));
#pragma omp parallel num_threads(threads)
    {
        const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
        double* G = Gall ? Gall + (size_t)tid * KT * N : NULL;
        double* ll = llall + (size_t)tid * 2 * N;
        double* gb = ll + N;
        double* X = (double*)malloc(sizeof(double) * TT * N);
        /* contiguous range of tiles per thread: the summation order depends only on `threads` */
        const int64_t t_lo = ntile * tid / nth, t_hi = ntile * (tid + 1) / nth;
        for (int64_t tile = t_lo; tile < t_hi; ++tile) {
            const int64_t t0 = tile * TT;
            const int rows = (int)((t0 + TT <= nT) ? TT : nT - t0);
            for (int t = 0; t < rows; ++t)
                for (int n = 0; n < N; ++n) X[t * N + n] = bias[n];
            /* forward: two time rows at a time share every Wmat row */
            for (int t = 0; t < rows; t += 2) {
                const int two = (t + 1 < rows);
                const double* f0 = fS + (size_t)(t0 + t) * K;
                const double* f1 = two ? f0 + K : f0;
                double* x0 = X + (size_t)t * N;
                double* x1 = two ? x0 + N : x0;
                for (int k = 0; k < K; ++k) {
                    const double a0 = f0[k], a1 = two ? f1[k] : 0.0;
                    const double* w = Wmat + (size_t)k * N;
                    if (two) {
#pragma omp simd
                        for (int n = 0; n < N; ++n) {
                            x0[n] += a0 * w[n];
                            x1[n] += a1 * w[n];
                        }
                    } else {
#pragma omp simd
                        for (int n = 0; n < N; ++n) x0[n] += a0 * w[n];
                    }
                }
                for (int j = 0; j < Dstim; ++j) {
                    const double* w = Wmat + (size_t)(K + j) * N;
                    const double a0 = fstim[(size_t)(t0 + t) * Dstim + j];
                    const double a1 = two ? fstim[(size_t)(t0 + t + 1) * Dstim + j] : 0.0;
                    for (int n = 0; n < N; ++n) {
                        x0[n] += a0 * w[n];
                        if (two) x1[n] += a1 * w[n];
                    }
                }
            }
            /* rate epilogue: X becomes r.  Vectorisable form (this file is built with -ffast-math so
             * that gcc uses glibc's libmvec exp / log1p / log): finite currents only. */
            for (int t = 0; t < rows; ++t) {
                const uint8_t* s = S + (size_t)(t0 + t) * N;
                double* x = X + (size_t)t * N;
                if (nlin == 1) {
#pragma omp simd
                    for (int n = 0; n < N; ++n) {
                        const double xv = x[n], sv = (double)s[n];
                        const double e = exp(-fabs(xv));
                        const double lam = fmax(xv, 0.0) + log1p(e);
                        const double inv = 1.0 / (1.0 + e);
                        const double sig = xv >= 0 ? inv : e * inv;
                        ll[n] += -dt * lam + log(lam) * sv;
                        const double r = (-dt + sv / lam) * sig;
                        x[n] = r;
                        gb[n] += r;
                    }
                } else {
#pragma omp simd
                    for (int n = 0; n < N; ++n) {
                        const double xv = x[n], sv = (double)s[n];
                        const double lam = exp(xv);
                        ll[n] += -dt * lam + xv * sv;
                        const double r = -dt * lam + sv;
                        x[n] = r;
                        gb[n] += r;
                    }
                }
            }
            if (!G) continue;
            /* backward: G[k][:] += sum_t F[t][k] r[t][:], four time rows per pass over G */
            for (int t = 0; t < rows; t += 4) {
                const int m = (rows - t < 4) ? rows - t : 4;
                const double* f = fS + (size_t)(t0 + t) * K;
                const double* r = X + (size_t)t * N;
                for (int k = 0; k < KT; ++k) {
                    double a[4] = {0, 0, 0, 0};
                    for (int q = 0; q < m; ++q)
                        a[q] = (k < K) ? f[(size_t)q * K + k] : fstim[(size_t)(t0 + t + q) * Dstim + (k - K)];
                    double* g = G + (size_t)k * N;
                    if (a[0] == 0.0 && a[1] == 0.0 && a[2] == 0.0 && a[3] == 0.0) continue;
                    const double* r0 = r;
                    const double* r1 = r + (m > 1 ? N : 0);
                    const double* r2 = r + (m > 2 ? 2 * N : 0);
                    const double* r3 = r + (m > 3 ? 3 * N : 0);
#pragma omp simd
                    for (int n = 0; n < N; ++n)
                        g[n] += a[0] * r0[n] + a[1] * r1[n] + a[2] * r2[n] + a[3] * r3[n];
                }
            }
        }
        free(X);
    }
    for (int n = 0; n < N; ++n) {
        double l = 0.0, g0 = 0.0;
        for (int t = 0; t < threads; ++t) {
            l += llall[(size_t)t * 2 * N + n];
            g0 += llall[(size_t)t * 2 * N + N + n];
        }
        ll_out[n] = l;
        if (grad_out) grad_out[(size_t)n * P] = g0;
    }
    if (grad_out) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int k = 0; k < KT; ++k) {
            for (int n = 0; n < N; ++n) {
                double g = 0.0;
                for (int t = 0; t < threads; ++t) g += Gall[((size_t)t * KT + k) * N + n];
                if (k < K)
                    grad_out[(size_t)n * P + 1 + Dstim + k] = g * Weff[(size_t)(k / B) * N + n];
                else
                    grad_out[(size_t)n * P + 1 + (k - K)] = g;
            }
        }
    }
    free(Wmat);
    free(bias);
    free(Gall);
    free(llall);
}

/* Glm.set_data (glm.py:99-110) re-copies S, fS (and fstim) into Theano shared variables before every
 * evaluation (coord_descent.py:52-57): the cost of that copy alone, one thread like numpy's memcpy. */
void oracle_set_data_copy(const double* fS, size_t n_fs, const double* Sf, size_t n_s, double* dst)
{
    memcpy(dst, fS, n_fs * sizeof(double));
    memcpy(dst + n_fs, Sf, n_s * sizeof(double));
}
