// k_prep_w, k_finalize* and the row kernels of the 3-phase path
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// prep: Wmat in MFMA B-fragment order + bias vector
//   Wfrag[pt][ks][lane] = Wmat[k = 4ks + (lane>>4)][n = 16pt + (lane&15)]
// ---------------------------------------------------------------------------
__global__ void k_prep_w(const double* __restrict__ theta, const double* __restrict__ Weff,
                         double* __restrict__ Wfrag, double* __restrict__ bias, int N, int B,
                         int Dstim, int Kimp, int Ktot, int KS, int n_lo, int npost, int nPT,
                         int pair, int Nall, int np0, int DsAll, int ds0, const int* __restrict__ pidx)
{
    // N / Dstim / Kimp describe the launch's feature-column slice (see FusedParams)
    const int P = 1 + DsAll + Nall * B;
    const long long total = (long long)nPT * KS * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int ks = (int)((i >> 6) % KS);
        const int pt = (int)((i >> 6) / KS);
        const int k = 4 * ks + (lane >> 4);
        const int n = 16 * pt + (lane & 15);
        double v = 0.0;
        if (n < npost && k < Ktot) {
            if (k < Kimp) {
                const int npre = np0 + k / B;
                v = theta[(size_t)n * P + 1 + DsAll + np0 * B + k] * Weff[(size_t)npre * Nall + (pidx ? pidx[n] : n_lo + n)];
            } else {
                v = theta[(size_t)n * P + 1 + ds0 + (k - Kimp)];
            }
        }
        // pair layout (V2): [pt][ks/2][lane][ks&1] so that one 16-byte load feeds two k-steps
        const long long o = pair ? (((long long)pt * (KS / 2) + (ks >> 1)) * 64 + lane) * 2 + (ks & 1) : i;
        Wfrag[o] = v;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nPT * 16; n += gridDim.x * blockDim.x)
        bias[n] = (n < npost) ? theta[(size_t)n * P] : 0.0;
}

// ---------------------------------------------------------------------------
// finalize: deterministic reduction of the per-chunk partials, Weff chain rule,
// scatter into the (npost, P) gradient layout
// ---------------------------------------------------------------------------
// ll_n and d ll_n / d bias of neuron n: its nChunks * nsub partials are one contiguous run (pgl_store_ll); 256 threads
// stride over it with four independent sums each, fixed-order butterfly + fixed-order combination of the waves --
// deterministic for a given launch geometry.
__device__ __forceinline__ void pgl_reduce_ll(const double* __restrict__ llpart, const double* __restrict__ gbpart,
                                              double* __restrict__ ll_out, double* __restrict__ grad_out,
                                              const int n, const int P, const int nPT, const int nChunks,
                                              const int nsub, double (*red)[64])
{
    // always the first 256 threads of the block, whatever its size: the ll of an ll-only call (k_finalize_ll) and of
    // an ll+grad call (trailing blocks of k_finalize) are then the same sums in the same order, bit for bit
    const int nthr = 256, t = (int)threadIdx.x, lane = t & 63, w = t >> 6, nw = 4;
    const int total = nChunks * nsub;
    const double* __restrict__ lp = llpart + (size_t)n * total;
    const double* __restrict__ gp = gbpart + (size_t)n * total;
    double sl[4] = {0.0, 0.0, 0.0, 0.0}, sg[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = t; i0 < total && t < nthr; i0 += 4 * nthr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * nthr;
            if (i < total) {
                sl[j] += lp[i];
                sg[j] += gp[i];
            }
        }
    }
    double a = (sl[0] + sl[1]) + (sl[2] + sl[3]), b = (sg[0] + sg[1]) + (sg[2] + sg[3]);
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0 && w < nw) {
        red[w][0] = a;
        red[w][1] = b;
    }
    __syncthreads();
    if (t == 0) {
        double x = 0.0, y = 0.0;
        for (int j = 0; j < nw; ++j) {
            x += red[j][0];
            y += red[j][1];
        }
        ll_out[n] = x;
        if (grad_out != nullptr) grad_out[(size_t)n * P] = y;
    }
}

__global__ __launch_bounds__(1024) void k_finalize(const double* __restrict__ Gpart, const double* __restrict__ llpart,
                           const double* __restrict__ gbpart, const double* __restrict__ Weff,
                           double* __restrict__ ll_out, double* __restrict__ grad_out, int N, int B,
                           int Dstim, int Kimp, int Ktot, int KT, int n_lo, int npost, int nPT,
                           int nChunks, int Nall, int np0, int DsAll, int ds0, int nsub, int kt0,
                           int nkt, const int* __restrict__ pidx, const double* __restrict__ wtpart = nullptr,
                           int nwt = 0, int ldy = 0, int Bt = 0)
{
    // reduces the k-tiles [kt0, kt0 + nkt) of every post tile (the two halves of the two-pass
    // kernels are reduced by separate launches: the first one runs beside pass 2).
    // A block of blockDim.x / 64 waves (1 .. 16) owns one 64-element fragment of G: its chunk partials are one
    // contiguous run of nChunks x 512 bytes (pgl_gpart), the waves take consecutive pieces of it.
    const int P = 1 + DsAll + Nall * B;
    // (Measured round 3 and dropped: splitting the chunks of a fragment over Q blocks whose last arriver -- agent-scope
    //  fence + counter -- combines the partials, to put a 16-fragment reduction on all CUs: the release / acquire fences
    //  (__threadfence in every wave of 800 blocks) write back and invalidate the XCD's L2: +27 us at C1, +140 us at C2.  The same holds for any reduction
    //  "in the tail" of the fused kernels: a kernel boundary is the cheapest cross-XCD synchronisation there is.)
    const int nwf = (int)(blockDim.x >> 6);
    const long long nfrag = (long long)nPT * nkt * 256;
    const int gblocks = (int)((nfrag + 63) / 64);
    __shared__ double red[16][64];
    if ((int)blockIdx.x >= gblocks) {
        // trailing blocks: ll_n and d ll_n / d bias (one block per neuron), when the caller folded the
        // ll reduction into this launch (nsub > 0) -- it then runs beside the G reduction
        if (nsub <= 0) return;
        const int n = (int)blockIdx.x - gblocks;
        if (n >= npost) {
            // ... and behind them (fused stimulus backward): d ll / d w_t[n][bt] = sum over the nwt block partials of
            // k_sepf_finish_d, one block per temporal basis, wave w takes the partials w, w + nwf, ... (fixed order)
            const int bt = n - npost;
            if (wtpart == nullptr || bt >= Bt || grad_out == nullptr) return;
            const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            if (lane < ldy) {
                int i = w, u = 0;
                for (; i < nwt; i += nwf, ++u) a[u & 3] += wtpart[((size_t)i * 3 + bt) * ldy + lane];
            }
            red[w][lane] = (a[0] + a[1]) + (a[2] + a[3]);
            __syncthreads();
            if (w == 0 && lane < npost) {
                double v = 0.0;
                for (int j = 0; j < nwf; ++j) v += red[j][lane];
                grad_out[(size_t)lane * P + 1 + bt] = v;
            }
            return;
        }
        pgl_reduce_ll(llpart, gbpart, ll_out, grad_out, n, P, nPT, nChunks, nsub, red);
        return;
    }
    if (grad_out == nullptr) return;
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const long long gid = blockIdx.x * 64LL + lane;
    const int r = (int)((gid >> 6) & 3);
    const int kt = kt0 + (int)((gid >> 8) % nkt);
    const int pt = (int)((gid >> 8) / nkt);
    const int k = 16 * kt + (lane >> 4) + 4 * r;
    const int n = 16 * pt + (lane & 15);
    const bool live = gid < nfrag && n < npost && k < Ktot;
    double s = 0.0;
    if (live) {
        // wave w sums the chunks [c0, c1) with eight interleaved partial sums (eight loads in flight per
        // lane); the waves' sums are combined in a fixed order: deterministic for a given geometry
        const int per = (nChunks + nwf - 1) / nwf;
        const int c0 = w * per, c1 = (c0 + per < nChunks) ? c0 + per : nChunks;
        const double* gp = Gpart + (((size_t)pt * KT + kt) * 4 + r) * ((size_t)nChunks * 64) + lane;
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += gp[(size_t)(c + j) * 64];
        }
        for (int j = 0; c < c1; ++c, ++j) a[j] += gp[(size_t)c * 64];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        s = 0.0;
        for (int j = 0; j < nwf; ++j) s += red[j][lane];
        if (k < Kimp) {
            const int npre = np0 + k / B;
            grad_out[(size_t)n * P + 1 + DsAll + np0 * B + k] = s * Weff[(size_t)npre * Nall + (pidx ? pidx[n] : n_lo + n)];
        } else {
            grad_out[(size_t)n * P + 1 + ds0 + (k - Kimp)] = s;
        }
    }
}

// ll-only evaluations: one block per neuron
__global__ __launch_bounds__(256) void k_finalize_ll(const double* __restrict__ llpart,
                                                     const double* __restrict__ gbpart,
                                                     double* __restrict__ ll_out,
                                                     double* __restrict__ grad_out, int P, int npost,
                                                     int nPT, int nChunks, int nsub)
{
    __shared__ double red[16][64];
    const int n = blockIdx.x;
    if (n >= npost) return;
    pgl_reduce_ll(llpart, gbpart, ll_out, grad_out, n, P, nPT, nChunks, nsub, red);
}

// ---------------------------------------------------------------------------
// Sliced (general) path, phase 2: x = Xbuf + bias -> ll terms and residuals r (in place).
// Thread = one column n of `rows` consecutive bins; per-thread ll / sum(r) partials are reduced
// per neuron by k_rows_reduce (fixed order).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_epilogue(double* __restrict__ Xbuf, int xstride,
                                                       const double* __restrict__ bias,
                                                       const uint8_t* __restrict__ S, int Nall,
                                                       int n_lo, int npost, long long t_lo,
                                                       long long t_hi, int rows, int nlin, double dt,
                                                       double* __restrict__ llp,
                                                       double* __restrict__ gbp,
                                                       const int* __restrict__ pidx)
{
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= npost) return;
    const long long t0 = t_lo + (long long)blockIdx.x * rows;
    long long t1 = t0 + rows;
    if (t1 > t_hi) t1 = t_hi;
    const double b = bias[n];
    double ll = 0.0, gb = 0.0;
    for (long long t = t0; t < t1; ++t) {
        const double x = Xbuf[t * xstride + n] + b;
        const double s = (double)S[t * Nall + (pidx ? pidx[n] : n_lo + n)];
        double term, res;
        pgl_rate_terms(x, s, nlin, dt, term, res, PGL_C);
        ll += term;
        gb += res;
        Xbuf[t * xstride + n] = res;
    }
    llp[(size_t)blockIdx.x * npost + n] = ll;
    gbp[(size_t)blockIdx.x * npost + n] = gb;
}

// rows [t_lo,t_hi) outside the evaluated range must carry r = 0 for the backward launches
__global__ void k_rows_zero(double* __restrict__ Xbuf, int xstride, long long r0, long long r1)
{
    const long long total = (r1 - r0) * xstride;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        Xbuf[r0 * xstride + i] = 0.0;
}

__global__ __launch_bounds__(64) void k_rows_reduce(const double* __restrict__ llp,
                                                    const double* __restrict__ gbp, int nblk,
                                                    int npost, int P, double* __restrict__ ll_out,
                                                    double* __restrict__ grad_out)
{
    const int n = blockIdx.x;
    double a = 0.0, g = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        a += llp[(size_t)b * npost + n];
        g += gbp[(size_t)b * npost + n];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        g += __shfl_xor(g, o, 64);
    }
    if (threadIdx.x == 0) {
        ll_out[n] = a;
        if (grad_out != nullptr) grad_out[(size_t)n * P] = g;
    }
}
