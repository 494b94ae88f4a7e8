// direct-form helpers: features, impulse currents, state read-back, per-pair inner ll
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// direct-form helpers (not MFMA): features, impulse currents, state, MCMC inner ll
// ---------------------------------------------------------------------------
// fS[t][n'][b]  (basis.py:201-236).  One block per 16-row tile.
__global__ void k_features(const int2* __restrict__ spk, const int* __restrict__ wlo,
                           const int* __restrict__ whi, const double* __restrict__ phi,
                           double* __restrict__ fS, long long nT, int N, int B, int R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < B * R; i += blockDim.x) phiS[i] = phi[i];
    __syncthreads();
    const int tile = blockIdx.x;
    const int t0 = tile * 16;
    const int K = N * B;
    for (int id = threadIdx.x; id < 16 * K; id += blockDim.x) {
        const int kb = id % K;
        const int t = id / K;
        const long long tg = (long long)t0 + t;
        if (tg >= nT) continue;
        const int np = kb / B, b = kb % B;
        const int lo = wlo[(size_t)tile * N + np];
        const int hi = whi[(size_t)tile * N + np];
        fS[tg * K + kb] = conv_one(spk, lo, hi, (int)tg, R, phiS + b * R);
    }
}

// I_impT[n'][t] = sum_b fS[t,n',b] w[n',b]   (impulse.py:58), transposed for coalescing.
// One block per 64-row tile (4 window tiles of 16).
__global__ void k_impulse_T(const int2* __restrict__ spk, const int* __restrict__ wlo,
                            const int* __restrict__ whi, const double* __restrict__ phi,
                            const double* __restrict__ w, double* __restrict__ IimpT,
                            long long nT, int nT16, int N, int B, int R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    double* wS = phiS + B * R;
    for (int i = threadIdx.x; i < B * R; i += blockDim.x) phiS[i] = phi[i];
    for (int i = threadIdx.x; i < N * B; i += blockDim.x) wS[i] = w[i];
    __syncthreads();
    const int t0 = blockIdx.x * 64;
    const int tl0 = blockIdx.x * 4;
    int tl1 = tl0 + 3;
    if (tl1 > nT16 - 1) tl1 = nT16 - 1;
    for (int id = threadIdx.x; id < 64 * N; id += blockDim.x) {
        const int t = id & 63;
        const int np = id >> 6;
        const long long tg = (long long)t0 + t;
        if (tg >= nT) continue;
        const int lo = wlo[(size_t)tl0 * N + np];
        const int hi = whi[(size_t)tl1 * N + np];
        double a = 0.0;
        for (int j = lo; j < hi; ++j) {
            const int2 e = spk[j];
            const int d = (int)tg - e.x - 1;
            if (d >= 0 && d < R) {
                double h = 0.0;
                for (int b = 0; b < B; ++b) h = fma(phiS[b * R + d], wS[np * B + b], h);
                a = fma((double)e.y, h, a);
            }
        }
        IimpT[(size_t)np * nT + tg] = a;
    }
}

// I_net[t] = sum_n' Weff_col[n'] I_impT[n'][t]  (glm.py:39);  I_stim[t] = fstim[t,:].wstim
__global__ void k_inet(const double* __restrict__ IimpT, const double* __restrict__ weff_col,
                       const double* __restrict__ fstim, const double* __restrict__ wstim,
                       double* __restrict__ Inet, double* __restrict__ Istim, long long nT, int N,
                       int Dstim)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int np = 0; np < N; ++np) a = fma(weff_col[np], IimpT[(size_t)np * nT + t], a);
        Inet[t] = a;
        double s = 0.0;
        for (int j = 0; j < Dstim; ++j) s = fma(fstim[t * Dstim + j], wstim[j], s);
        Istim[t] = s;
    }
}

__global__ void k_lam(const double* __restrict__ Inet, const double* __restrict__ Istim,
                      double bias, int nlin, double* __restrict__ lam, long long nT)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        const double x = bias + Istim[t] + Inet[t];
        double sig, ll;
        lam[t] = (nlin == 1) ? pgl_softplus_parts(x, sig, ll) : exp(x);
    }
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, double a, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        y[i] = fma(a, x[i], y[i]);
}

// MCMC inner ll (gibbs.py:910-937): for k < K:
//   x = bias + stim[t] + base[t] - aw_cur*col[t] + w[k]*col[t]
//   ll_k = sum_t -dt*lam_k(t)  +  sum_{spike bins of n_post} S*log(lam_k(t))
// The first sum streams all nT bins (k_ll_current: one exp per element and weight, log1p by
// its series in the |x| > 9.25 tail); the second only visits the post-synaptic neuron's own
// spike events (k_ll_current_spikes, ~2 % of the bins), so no log is evaluated for silent bins.
#define PGL_KMAX 16
__device__ __forceinline__ double pgl_lambda_only(const double x, const int nlin,
                                                  const double* __restrict__ C)
{
    if (nlin != 1) return pgl_exp(x, C);
    const double e = pgl_exp(-fabs(x), C);
    double l1p;
    if (__all(e < C[23])) {                        // |x| > 9.25: alternating series, error < e^6
        l1p = e * fma(-e, fma(-e, fma(-e, fma(-e, 0.2, 0.25), C[22]), 0.5), 1.0);
    } else if (__all(e < 0.1)) {                   // |x| > 2.3: log1p(e) = 2 atanh(s), s = e/(2+e) < 0.048
        const double rc = pgl_rcp(2.0 + e);
        const double s = e * rc;
        const double z = s * s;                    // z < 2.3e-3: z^7/15 < 2.3e-20
        const double q = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, 1.0 / 13.0, 1.0 / 11.0), 1.0 / 9.0), 1.0 / 7.0),
                                           0.2), C[22]), 1.0);
        l1p = e * ((rc + rc) * q);                 // e last: a denormal e (x ~ -745) must not be halved to zero on the way
    } else {
        const double u = 1.0 + e;
        l1p = pgl_log(u, C) + (e - (u - 1.0)) * pgl_rcp(u);
    }
    return fmax(x, 0.0) + l1p;
}

// candidate weights travel as a kernel argument (<= 16 doubles): no host-to-device copy per batch
struct PglWeights {
    double w[PGL_KMAX];
};

__global__ __launch_bounds__(256) void k_ll_current(const double* __restrict__ base,
                                                    const double* __restrict__ stim,
                                                    const double* __restrict__ colv, double bias,
                                                    double aw_cur, const PglWeights wv,
                                                    int K, int nlin, double dt, long long t_lo,
                                                    long long nT, double* __restrict__ part)
{
    __shared__ double red[4][PGL_KMAX];
    double wk[PGL_KMAX], acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        wk[k] = (k < K) ? wv.w[k] : 0.0;
        acc[k] = 0.0;
    }
    for (long long t = t_lo + blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        const double c = colv[t];
        const double x0 = bias + (stim ? stim[t] : 0.0) + base[t] - aw_cur * c;
#pragma unroll
        for (int k = 0; k < PGL_KMAX; ++k) {
            if (k < K) {
                const double lam = pgl_lambda_only(fma(wk[k], c, x0), nlin, PGL_C);
                // reference semantics: lam == 0 makes log(lam)*S NaN even for S = 0 (glm.py:52)
                acc[k] += (lam == 0.0) ? __builtin_nan("") : -dt * lam;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PGL_KMAX)
        part[(size_t)blockIdx.x * PGL_KMAX + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// spike-bin part: events [e_lo, e_hi) of the post-synaptic neuron, grid-strided; one partial
// row per block (fixed order)
__global__ __launch_bounds__(256) void k_ll_current_spikes(const int2* __restrict__ spk, int e_lo,
                                                           int e_hi, const double* __restrict__ base,
                                                           const double* __restrict__ stim,
                                                           const double* __restrict__ colv,
                                                           double bias, double aw_cur,
                                                           const PglWeights wv, int K,
                                                           int nlin, double* __restrict__ part)
{
    __shared__ double red[4][PGL_KMAX];
    double acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) acc[k] = 0.0;
    for (int i = e_lo + blockIdx.x * blockDim.x + threadIdx.x; i < e_hi; i += gridDim.x * blockDim.x) {
        const int2 e = spk[i];
        const long long t = e.x;
        const double c = colv[t];
        const double x0 = bias + (stim ? stim[t] : 0.0) + base[t] - aw_cur * c;
        const double s = (double)e.y;
        for (int k = 0; k < K; ++k) {
            const double x = fma(wv.w[k], c, x0);
            const double loglam = (nlin == 1) ? pgl_log(pgl_lambda_only(x, nlin, PGL_C), PGL_C) : x;
            acc[k] = fma(s, loglam, acc[k]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PGL_KMAX)
        part[(size_t)blockIdx.x * PGL_KMAX + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[k] = sum over the partial rows; one 64-lane block per k (fixed summation order)
__global__ __launch_bounds__(64) void k_reduce_parts(const double* __restrict__ part, int nblocks,
                                                     int K, double* __restrict__ out)
{
    const int k = blockIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += part[(size_t)b * PGL_KMAX + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[k] = s;
}
