// stimulus feature builds, separable stimulus (tap rate and frame rate), thin GEMMs, STA
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// Stimulus feature build on the device (bkgd.py:122-154, 303-340; basis.py:201-273):
//   1. zx[t,bx] = sum_d interp(stim)[t,d] * basis_x[d,bx]   (np.interp onto the dt grid, then the
//      spatial projection; basis_x == nullptr means identity, Bx == D)
//   2. f[t,bx,bt] = sum_{tau=1..Rt} zx[t-tau,bx] * basis_t[tau-1,bt]   (strictly causal)
// written as fstim[t][col], col = bt*Bx+bx (layout 0, SpatiotemporalStimulus) or bx*Bt+bt
// (layout 1, BasisStimulus: d*B+b).
// ---------------------------------------------------------------------------
__global__ void k_stim_project(const double* __restrict__ stim, long long Tstim, int D,
                               double dt_stim, double dt, const double* __restrict__ basis_x,
                               int Bx, double* __restrict__ zx, long long nT)
{
    const long long total = nT * Bx;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / Bx;
        const int bx = (int)(i - t * Bx);
        const double x = dt * (double)t;
        // np.interp: clamp outside [xp[0], xp[-1]], else slope form on the bracketing interval
        long long i0 = (long long)floor(x / dt_stim);
        if (i0 > Tstim - 2) i0 = Tstim - 2;
        if (i0 < 0) i0 = 0;
        // guard against x/dt_stim rounding across a knot
        while (i0 + 1 < Tstim - 1 && dt_stim * (double)(i0 + 1) <= x) ++i0;
        while (i0 > 0 && dt_stim * (double)i0 > x) --i0;
        const double x0 = dt_stim * (double)i0, x1 = dt_stim * (double)(i0 + 1);
        const bool beyond = (Tstim < 2) || (x >= dt_stim * (double)(Tstim - 1));
        double acc = 0.0;
        const int d_lo = (basis_x == nullptr) ? bx : 0;          // identity spatial basis: column bx only
        const int d_hi = (basis_x == nullptr) ? bx + 1 : D;
        for (int d = d_lo; d < d_hi; ++d) {
            double v;
            if (beyond) {
                v = stim[(Tstim - 1) * D + d];
            } else {
                const double f0 = stim[i0 * D + d], f1 = stim[(i0 + 1) * D + d];
                v = (f1 - f0) / (x1 - x0) * (x - x0) + f0;
            }
            acc = (basis_x == nullptr) ? v : fma(v, basis_x[(size_t)d * Bx + bx], acc);
        }
        zx[i] = acc;
    }
}

// one block = 256 consecutive bins of one spatial column bx; zx window and basis_t in LDS
__global__ __launch_bounds__(256) void k_stim_conv(const double* __restrict__ zx,
                                                   const double* __restrict__ basis_t, int Rt,
                                                   int Bt, int Bx, int layout,
                                                   double* __restrict__ fstim, long long nT)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* zs = reinterpret_cast<double*>(smem);           // [Rt + 256]
    double* bs = zs + Rt + 256;                             // [Rt][Bt]
    const int bx = blockIdx.y;
    const long long t0 = (long long)blockIdx.x * 256;
    for (int i = threadIdx.x; i < Rt + 256; i += 256) {
        const long long t = t0 - Rt + i;
        zs[i] = (t >= 0 && t < nT) ? zx[t * Bx + bx] : 0.0;
    }
    for (int i = threadIdx.x; i < Rt * Bt; i += 256) bs[i] = basis_t[i];
    __syncthreads();
    const long long t = t0 + threadIdx.x;
    if (t >= nT) return;
    const int Dst = Bx * Bt;
    for (int bt = 0; bt < Bt; ++bt) {
        double a = 0.0;
        // zs index of bin t - tau is threadIdx.x + Rt - tau
        for (int tau = 1; tau <= Rt; ++tau) a = fma(zs[threadIdx.x + Rt - tau], bs[(tau - 1) * Bt + bt], a);
        const int colo = layout == 0 ? bt * Bx + bx : bx * Bt + bt;
        fstim[t * Dst + colo] = a;
    }
}

// ---------------------------------------------------------------------------
// Separable (rank-1) stimulus path for wide stimuli (SpatiotemporalStimulus, bkgd.py:172-345):
//   I_stim[t,n] = sum_{bt,bx} fstim[t,bt,bx] w_t[n,bt] w_x[n,bx]          (bkgd.py:214-227)
// with fstim[t,bt,bx] = sum_tau zx[t-tau,bx] basis_t[tau-1,bt] and zx = interp(stim) . basis_x
// (bkgd.py:303-340, basis.py:238-273).  Interpolation, projection and filtering are linear, so
//   y_n       = interp( (stim . basis_x) . w_x[n] )        a GEMM at the STIMULUS frame rate (T_stim rows)
//   I_stim[:,n] = causal conv of y_n with h_n = basis_t . w_t[n]            (Rt taps)
// and the dense (nT, Bt*Bx) feature matrix (7.4 GB at D_stim = 1024, T = 300 s) is never formed.
// Gradients by the transposes:  rho_n[tau] = sum_t r[t,n] y_n[t-tau]  ->  d/dw_t = basis_t^T rho_n;
//   q_n[s] = sum_tau r[s+tau,n] h_n[tau-1],  Qf = interp^T q_n,  d/dw_x = (stim . basis_x)^T Qf.
// ---------------------------------------------------------------------------
// np.interp of a frame-rate series (clamped at both ends): value at bin t and, for the transpose, the
// bracketing frame and weight.  Same knot logic as k_stim_project.
__device__ __forceinline__ void pgl_interp_knot(const long long t, const double dt, const double dt_stim,
                                                const long long Tstim, long long& i0, double& a, bool& beyond)
{
    const double x = dt * (double)t;
    i0 = (long long)floor(x / dt_stim);
    if (i0 > Tstim - 2) i0 = Tstim - 2;
    if (i0 < 0) i0 = 0;
    while (i0 + 1 < Tstim - 1 && dt_stim * (double)(i0 + 1) <= x) ++i0;
    while (i0 > 0 && dt_stim * (double)i0 > x) --i0;
    beyond = (Tstim < 2) || (x >= dt_stim * (double)(Tstim - 1));
    const double x0 = dt_stim * (double)i0, x1 = dt_stim * (double)(i0 + 1);
    a = (x - x0) / (x1 - x0);
}
__device__ __forceinline__ double pgl_interp_frames(const double* __restrict__ yf, const long long t,
                                                    const double dt, const double dt_stim, const long long Tstim)
{
    if (t < 0) return 0.0;
    long long i0;
    double a;
    bool beyond;
    pgl_interp_knot(t, dt, dt_stim, Tstim, i0, a, beyond);
    if (beyond) return yf[Tstim - 1];
    const double f0 = yf[i0], f1 = yf[i0 + 1];
    return (f1 - f0) * a + f0;
}

// C[m][n] (ldc) = sum_k A[m][k] (lda) * B[n][k] (ldb): small f64 GEMM, 64 x 64 tiles, 256 threads x (4 x 4)
__global__ __launch_bounds__(256) void k_gemm_nt(const double* __restrict__ A, int lda,
                                                 const double* __restrict__ Bm, int ldb,
                                                 double* __restrict__ C, int ldc, int M, int Nn, int Kd)
{
    __shared__ double As[32][65], Bs[32][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < Kd; k0 += 32) {
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i >> 5, kk = i & 31;
            As[kk][r] = (m0 + r < M && k0 + kk < Kd) ? A[(size_t)(m0 + r) * lda + k0 + kk] : 0.0;
            Bs[kk][r] = (n0 + r < Nn && k0 + kk < Kd) ? Bm[(size_t)(n0 + r) * ldb + k0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                b[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + ty * 4 + i < M && n0 + tx * 4 + j < Nn)
                C[(size_t)(m0 + ty * 4 + i) * ldc + n0 + tx * 4 + j] = acc[i][j];
}

struct SepParams {
    const double* __restrict__ Yf;       // (npost, Tstim) frame-rate projections y_n
    const double* __restrict__ basis_t;  // (Rt, Bt)
    const double* __restrict__ theta;    // (npost, P) rows [bias, w_t(Bt), w_x(Bx), w_imp]
    int P, Bt, Rt, npost, xs;
    long long Tstim, nT, t_lo, t_hi;
    double dt, dt_stim;
    double* __restrict__ X;              // (nT, xs) currents (forward: += I_stim) / residuals r (backward)
};

#define PGL_SEP_TB 1024
// forward: X[t][j] += sum_{tau=1..Rt} y_j[t-tau] h_j[tau-1];  grid = (time blocks, npost), block = 256
__global__ __launch_bounds__(256) void k_sep_conv_fwd(const SepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ys = reinterpret_cast<double*>(smem);            // [Rt + TB]: bins tb0 - Rt .. tb0 + TB - 1
    double* hs = ys + p.Rt + PGL_SEP_TB;                     // [Rt]
    const int j = blockIdx.y;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * PGL_SEP_TB;
    const double* yf = p.Yf + (size_t)j * p.Tstim;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256)
        ys[i] = pgl_interp_frames(yf, tb0 - p.Rt + i, p.dt, p.dt_stim, p.Tstim);
    for (int i = threadIdx.x; i < p.Rt; i += 256) {
        double h = 0.0;
        for (int b = 0; b < p.Bt; ++b) h = fma(p.basis_t[(size_t)i * p.Bt + b], p.theta[(size_t)j * p.P + 1 + b], h);
        hs[i] = h;
    }
    __syncthreads();
    // thread owns 4 consecutive bins: one new y per tap, four FMAs
    const int o = threadIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // out[o+q] = sum_tau ys[Rt + o + q - tau] hs[tau-1]
    double w0 = ys[p.Rt + o - 1 + 0], w1 = ys[p.Rt + o - 1 + 1], w2 = ys[p.Rt + o - 1 + 2], w3 = ys[p.Rt + o - 1 + 3];
    for (int tau = 1; tau <= p.Rt; ++tau) {
        const double h = hs[tau - 1];
        acc[0] = fma(w0, h, acc[0]);
        acc[1] = fma(w1, h, acc[1]);
        acc[2] = fma(w2, h, acc[2]);
        acc[3] = fma(w3, h, acc[3]);
        w3 = w2; w2 = w1; w1 = w0;
        w0 = (tau < p.Rt) ? ys[p.Rt + o - 1 - tau] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long t = tb0 + o + q;
        if (t < p.t_hi && t < p.nT) p.X[t * p.xs + j] += acc[q];
    }
}

// backward, part 1: q_j[s] = sum_{tau=1..Rt} r[s+tau][j] h_j[tau-1]  -> Qb[j][s - t_lo_q]  for the bins s
// that can reach the evaluated range, s in [t_lo - Rt, t_hi);  grid = (time blocks over that range, npost)
__global__ __launch_bounds__(256) void k_sep_conv_bwd(const SepParams p, double* __restrict__ Qb,
                                                      long long s_lo, long long nS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* rs = reinterpret_cast<double*>(smem);            // [TB + Rt]: r of bins sb0 .. sb0 + TB + Rt - 1
    double* hs = rs + p.Rt + PGL_SEP_TB;
    const int j = blockIdx.y;
    const long long sb0 = s_lo + (long long)blockIdx.x * PGL_SEP_TB;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256) {
        const long long t = sb0 + i;
        rs[i] = (t >= p.t_lo && t < p.t_hi) ? p.X[t * p.xs + j] : 0.0;
    }
    for (int i = threadIdx.x; i < p.Rt; i += 256) {
        double h = 0.0;
        for (int b = 0; b < p.Bt; ++b) h = fma(p.basis_t[(size_t)i * p.Bt + b], p.theta[(size_t)j * p.P + 1 + b], h);
        hs[i] = h;
    }
    __syncthreads();
    const int o = threadIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // q[o+q'] = sum_tau rs[o + q' + tau] hs[tau-1]
    double w0 = rs[o + 1], w1 = rs[o + 2], w2 = rs[o + 3], w3 = rs[o + 4];
    for (int tau = 1; tau <= p.Rt; ++tau) {
        const double h = hs[tau - 1];
        acc[0] = fma(w0, h, acc[0]);
        acc[1] = fma(w1, h, acc[1]);
        acc[2] = fma(w2, h, acc[2]);
        acc[3] = fma(w3, h, acc[3]);
        w0 = w1; w1 = w2; w2 = w3;
        w3 = (tau < p.Rt) ? rs[o + 4 + tau] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long s = sb0 + o + q;
        if (s - s_lo < nS) Qb[(size_t)j * nS + (s - s_lo)] = (s >= 0) ? acc[q] : 0.0;
    }
}

// backward, part 2: Qf[j][f] = sum_s w(s -> f) q_j[s]  (transpose of np.interp); thread = (f, j), fixed order
__global__ __launch_bounds__(256) void k_sep_interp_T(const SepParams p, const double* __restrict__ Qb,
                                                      long long s_lo, long long nS, double* __restrict__ Qf)
{
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (f >= p.Tstim) return;
    const double ratio = p.dt_stim / p.dt;
    long long a = (long long)floor((double)(f - 1) * ratio) - 2, b = (long long)ceil((double)(f + 1) * ratio) + 2;
    if (f == p.Tstim - 1) b = s_lo + nS;                      // clamped tail: every later bin reads the last frame
    if (a < s_lo) a = s_lo;
    if (a < 0) a = 0;
    if (b > s_lo + nS) b = s_lo + nS;
    double acc = 0.0;
    for (long long s = a; s < b; ++s) {
        long long i0;
        double w;
        bool beyond;
        pgl_interp_knot(s, p.dt, p.dt_stim, p.Tstim, i0, w, beyond);
        double c = 0.0;
        if (beyond) c = (f == p.Tstim - 1) ? 1.0 : 0.0;
        else if (i0 == f) c = 1.0 - w;
        else if (i0 + 1 == f) c = w;
        if (c != 0.0) acc = fma(c, Qb[(size_t)j * nS + (s - s_lo)], acc);
    }
    Qf[(size_t)j * p.Tstim + f] = acc;
}

// backward, part 3: rho_j[tau] = sum_t r[t][j] y_j[t-tau] over one time block -> part[blk][j][tau-1]
// grid = (time blocks over [t_lo, t_hi), npost), block = 256 (thread = lag, looping if Rt > 256)
__global__ __launch_bounds__(256) void k_sep_corr(const SepParams p, double* __restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ys = reinterpret_cast<double*>(smem);            // [Rt + TB]
    double* rs = ys + p.Rt + PGL_SEP_TB;                     // [TB]
    const int j = blockIdx.y;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * PGL_SEP_TB;
    const double* yf = p.Yf + (size_t)j * p.Tstim;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256)
        ys[i] = pgl_interp_frames(yf, tb0 - p.Rt + i, p.dt, p.dt_stim, p.Tstim);
    for (int i = threadIdx.x; i < PGL_SEP_TB; i += 256) {
        const long long t = tb0 + i;
        rs[i] = (t < p.t_hi) ? p.X[t * p.xs + j] : 0.0;
    }
    __syncthreads();
    for (int tau = 1 + threadIdx.x; tau <= p.Rt; tau += 256) {
        double acc = 0.0;
        for (int i = 0; i < PGL_SEP_TB; ++i) acc = fma(rs[i], ys[p.Rt + i - tau], acc);
        part[((size_t)blockIdx.x * p.npost + j) * p.Rt + (tau - 1)] = acc;
    }
}

// d ll / d w_t[j][bt] = sum_tau basis_t[tau-1][bt] * sum_blk part[blk][j][tau-1]; grid = npost, block = 64
__global__ __launch_bounds__(64) void k_sep_wt_grad(const SepParams p, const double* __restrict__ part,
                                                    int nblk, double* __restrict__ grad)
{
    const int j = blockIdx.x;
    for (int b = 0; b < p.Bt; ++b) {
        double s = 0.0;
        for (int tau = threadIdx.x; tau < p.Rt; tau += 64) {
            double rho = 0.0;
            for (int k = 0; k < nblk; ++k) rho += part[((size_t)k * p.npost + j) * p.Rt + tau];
            s = fma(p.basis_t[(size_t)tau * p.Bt + b], rho, s);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) grad[(size_t)j * p.P + 1 + b] = s;
    }
}

// ---------------------------------------------------------------------------
// Separable stimulus at the FRAME rate (dt_stim = q dt, q integer; bkgd.py:303-340 interpolates the stimulus
// linearly between frames, basis.py:238-273 filters it causally).  The interpolated projection y_n is piecewise
// linear over q-bin frames, so the Rt-tap convolution of bin t = q F + o collapses to J = ceil(Rt / q) + 2 frame
// values:
//   I_stim[t,n] = sum_{j<J} sum_{bt} C[row(t)][j][bt] w_t[n,bt] z_n[base(F) + j],   z_n = (stim . basis_x) . w_x[n]
//   base(F) = max(F - M, 0),  M = ceil(Rt / q),  row(t) = t for t < q M (the head, where bins t - tau < 0 are
//   dropped),  q M + o from there on (periodic in the frame);  C is built once per stimulus on the host
//   (build_frame_table): C[row][j][bt] = sum_tau basis_t[tau-1][bt] * (weight of frame base + j in y(t - tau)).
// 15 multiply-adds per bin and neuron at the C5 stress shape instead of 300 taps; the frame index past the last
// frame clamps (np.interp holds the last value).  Currents and residuals travel in the slab layout of the fused
// kernels' accumulators: X[tile - tile0][post tile][r][lane], element (r, lane) = bin 16 tile + (lane >> 4) + 4 r of
// neuron 16 pt + (lane & 15).
//   forward   x[t,n]      = < C[row(t)], ZW_n >,  ZW_n[j][bt] = z_n[base + j] w_t[n,bt]      (k_sepf_fwd)
//   backward  V_n[F][j][bt] = sum_o r[qF+o,n] C[row][j][bt];  d/dw_t[n,bt] = sum_F sum_j V z_n[base + j];
//             H_n[F][j] = sum_bt V w_t[n,bt];  d/dz_n[f] = sum of the H[F][j] with base(F) + j = f   (k_sepf_bwd,
//             k_sepf_finish);  d/dw_x = (stim . basis_x)^T d/dz  (k_gemm_mfma)
// A wave owns one frame F and 64 neurons (4 post tiles): the table row is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
struct SepfParams {
    const double* __restrict__ Ctab;     // [q (M + 1)][J][BT]
    const double* __restrict__ YfT;      // [Tstim][ldy] frame-rate projections z_n (transposed; written by k_gemm_mfma)
    const double* __restrict__ theta;    // (npost, P) rows [bias, w_t(Bt), w_x(Bx), w_imp]
    double* __restrict__ X;              // slab: currents (forward, written) / residuals (backward, read)
    double* __restrict__ Hb;             // [F1 - F0 + 1][J][ldy]
    double* __restrict__ wpart;          // [workgroups of k_sepf_bwd = ceil(frames / 4)][BT][ldy]
    double* __restrict__ QvT;            // [Tstim][ldy]  d ll / d z_n[f]
    double* __restrict__ grad;           // (npost, P): the w_t columns are written by k_sepf_finish
    int P, Bt, M, q, npost, nPT, ldy, tile0, nTiles;
    long long Tstim, F0, F1;             // frames that hold bins of the tile range
    // fused stimulus backward (k_fused7<.., 3> -> k_sepf_finish_d): pieces D[base - B0][slot][post tile][5][64]
    const double* __restrict__ D;
    long long B0, B1;                    // frame bases of the first / last evaluated tile
    int SL, tilesPerChunk;
};

template <int J, int BT>
__global__ __launch_bounds__(256) void k_sepf_fwd(const SepfParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int pt = blockIdx.y * 4 + grp;
    const int n = pt * 16 + col;
    const bool vp = pt < p.nPT, vn = vp && n < p.npost;
    const long long F = p.F0 + (long long)blockIdx.x * 4 + wave;
    if (F > p.F1) return;
    const long long base = (F > p.M) ? F - p.M : 0;
    double zw[J][BT];
    {
        double w[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) w[bt] = (vn && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            long long f = base + j;
            if (f > p.Tstim - 1) f = p.Tstim - 1;
            const double z = vn ? p.YfT[(size_t)f * p.ldy + n] : 0.0;
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) zw[j][bt] = z * w[bt];
        }
    }
    const long long tb = (long long)p.tile0 * 16, te = tb + (long long)p.nTiles * 16;
    const long long t0 = F * p.q;
    const int o_lo = (int)((tb > t0) ? tb - t0 : 0), o_hi = (int)((te - t0 < p.q) ? te - t0 : p.q);
    // the table row is wave-uniform: constant address space = scalar loads, the multiply-adds take it from SGPRs
    const pgl_k_cdp crow = (pgl_k_cdp)(p.Ctab + (size_t)(((F < p.M) ? F : p.M) * p.q) * (J * BT));
    double* const xl = p.X + (vp ? (size_t)pt * 256 + col : 0);
#pragma unroll 2
    for (int o = o_lo; o < o_hi; ++o) {
        const pgl_k_cdp cr = crow + (size_t)o * (J * BT);
        double x0 = 0.0, x1 = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                if ((j * BT + bt) & 1)
                    x1 = fma(cr[j * BT + bt], zw[j][bt], x1);
                else
                    x0 = fma(cr[j * BT + bt], zw[j][bt], x0);
            }
        const long long tl = t0 + o - tb;
        if (vp) xl[(size_t)(tl >> 4) * p.nPT * 256 + (size_t)(((tl & 15) >> 2) * 64 + (tl & 3) * 16)] = x0 + x1;
    }
}

template <int J, int BT>
__global__ __launch_bounds__(256) void k_sepf_bwd(const SepfParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int pt = blockIdx.y * 4 + grp;
    const int n = pt * 16 + col;
    const bool vp = pt < p.nPT, vn = vp && n < p.npost;
    __shared__ double gws[4][BT][64];
    const long long F = p.F0 + (long long)blockIdx.x * 4 + wave;
    const bool vF = F <= p.F1;               // (no early return: the workgroup meets at a barrier below)
    const long long base = (F > p.M) ? F - p.M : 0;
    double V[J][BT];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) V[j][bt] = 0.0;
    const long long tb = (long long)p.tile0 * 16, te = tb + (long long)p.nTiles * 16;
    const long long t0 = F * p.q;
    const int o_lo = (int)((tb > t0) ? tb - t0 : 0), o_hi = vF ? (int)((te - t0 < p.q) ? te - t0 : p.q) : 0;
    const pgl_k_cdp crow = (pgl_k_cdp)(p.Ctab + (size_t)(((F < p.M) ? F : p.M) * p.q) * (J * BT));
    const double* const xl = p.X + (vp ? (size_t)pt * 256 + col : 0);
#pragma unroll 4
    for (int o = o_lo; o < o_hi; ++o) {
        const pgl_k_cdp cr = crow + (size_t)o * (J * BT);
        const long long tl = t0 + o - tb;
        const double r = vn ? xl[(size_t)(tl >> 4) * p.nPT * 256 + (size_t)(((tl & 15) >> 2) * 64 + (tl & 3) * 16)] : 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) V[j][bt] = fma(r, cr[j * BT + bt], V[j][bt]);
    }
    double w[BT], gw[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        w[bt] = (vn && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
        gw[bt] = 0.0;
    }
    const size_t fo = (size_t)(F - p.F0);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        long long f = base + j;
        if (f > p.Tstim - 1) f = p.Tstim - 1;
        const double z = (vn && vF) ? p.YfT[(size_t)f * p.ldy + n] : 0.0;
        double h = 0.0;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            h = fma(V[j][bt], w[bt], h);
            gw[bt] = fma(V[j][bt], z, gw[bt]);
        }
        if (vp && vF) p.Hb[(fo * J + j) * p.ldy + n] = h;
    }
    // d / d w_t: the four frames of the workgroup in a fixed order -> one partial per workgroup
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) gws[wave][bt][lane] = gw[bt];
    __syncthreads();
    if (wave == 0 && vp) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
            p.wpart[((size_t)blockIdx.x * BT + bt) * p.ldy + n] =
                ((gws[0][bt][lane] + gws[1][bt][lane]) + gws[2][bt][lane]) + gws[3][bt][lane];
    }
}

// blocks [0, nA): QvT[f][n] = sum of the H[F][j][n] with base(F) + j = f (frames past the last one fold into it),
// 16 frames x 64 neurons per block;  blocks [nA, ...): d ll / d w_t[n][bt] = sum_F wpart[F][bt][n] in a fixed order,
// one block per (64 neurons, bt)
template <int J, int BT>
__global__ __launch_bounds__(1024) void k_sepf_finish(const SepfParams p, const int nA, const int nG)
{
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nF = p.F1 - p.F0 + 1;
    if ((int)blockIdx.x < nA) {
        const int g = blockIdx.x % nG;
        const long long f = (long long)(blockIdx.x / nG) * 16 + wave;
        const int n = g * 64 + lane;
        if (f >= p.Tstim || n >= p.ldy) return;
        auto gather = [&](const long long fv) -> double {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {                 // frames F >= M: base = F - M
                const long long F = fv + p.M - j;
                if (F >= p.M && F >= p.F0 && F <= p.F1) a += p.Hb[((size_t)(F - p.F0) * J + j) * p.ldy + n];
            }
            if (fv < J)                                   // head frames F < M: base = 0, j = fv
                for (long long F = p.F0; F < p.M && F <= p.F1; ++F) a += p.Hb[((size_t)(F - p.F0) * J + fv) * p.ldy + n];
            return a;
        };
        double acc = gather(f);
        if (f == p.Tstim - 1) {
            long long fmax = p.F1 + 1;
            if (fmax < J - 1) fmax = J - 1;
            for (long long fv = f + 1; fv <= fmax; ++fv) acc += gather(fv);
        }
        p.QvT[(size_t)f * p.ldy + n] = acc;
    } else {
        const int b = blockIdx.x - nA;
        const int g = b % nG, bt = b / nG;
        const int n = g * 64 + lane;
        // wave w sums the workgroup partials w, w + 16, ... of k_sepf_bwd (eight independent chains: the loads of a
        // chain are a latency each)
        const long long nW = (nF + 3) / 4;
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (n < p.ldy) {
            long long F = wave;
            for (; F + 16 * 7 < nW; F += 16 * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += p.wpart[((size_t)(F + 16 * u) * BT + bt) * p.ldy + n];
            }
            for (int u = 0; F < nW; F += 16, ++u) a[u & 7] += p.wpart[((size_t)F * BT + bt) * p.ldy + n];
        }
        red[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        if (wave == 0 && n < p.npost && bt < p.Bt) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += red[w][lane];
            p.grad[(size_t)n * p.P + 1 + bt] = v;
        }
    }
}

// Fused stimulus backward, second half: k_fused7<.., 3> left D[(j', bt)][n] = sum_i A[i][(j', bt)] r[i][n] per frame base b
// (column (j', bt) belongs to frame min(b + j', Tstim - 1)), one piece per chunk that holds tiles of the base.  Here, per
// frame f and row n (one wave per frame, 16 frames per block, lanes = the <= 64 rows of the four post tiles):
//   Dt[f][bt] = sum of the pieces' columns (f - b, bt) over the bases b = f - 5 .. f (and, for the last frame, of the
//               columns that clamp onto it),
//   QvT[f][n] = sum_bt Dt w_t[n][bt]          (d ll / d z_n[f]: the d/dw_x GEMM follows),
//   wpart[block][bt][n] = sum over the block's frames of Dt z_n[f]   (d ll / d w_t: summed over the blocks by the
//               trailing blocks of k_finalize, in a fixed order).
// Which (base, slot) pieces exist follows from the launch geometry alone (pgl_sepd_first_tile): nothing is zeroed.
__global__ __launch_bounds__(1024) void k_sepf_finish_d(const SepfParams p)
{
    __shared__ double red[16][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane, pt = lane >> 4, col = lane & 15;
    const long long f = (long long)blockIdx.x * 16 + wave;
    const bool vn = n < p.ldy && pt < p.nPT;
    const long long tileE = (long long)p.tile0 + p.nTiles - 1;
    double dt[3] = {0.0, 0.0, 0.0};
    auto gather = [&](const long long fv) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const long long b = fv - j;
            if (b < p.B0 || b > p.B1) continue;
            const long long tf = pgl_sepd_first_tile(b, p.M, p.q, p.tile0);
            long long tl = ((b + 1 + p.M) * (long long)p.q + 15) / 16 - 1;
            if (tl > tileE) tl = tileE;
            const int c0 = (int)((tf - p.tile0) / p.tilesPerChunk), c1 = (int)((tl - p.tile0) / p.tilesPerChunk);
            for (int sl = 0; sl <= c1 - c0; ++sl) {
                const double* dp = p.D + ((((size_t)(b - p.B0) * p.SL + sl) * p.nPT + pt) * 5) * 64 + col;
#pragma unroll
                for (int bt = 0; bt < 3; ++bt) {
                    const int c = 3 * j + bt;
                    const int o = (c < 16) ? (c >> 2) * 64 + (c & 3) * 16 : 4 * 64 + (c - 16) * 16;
                    dt[bt] += dp[o];
                }
            }
        }
    };
    if (vn && f < p.Tstim) {
        gather(f);
        if (f == p.Tstim - 1)
            for (long long fv = f + 1; fv <= p.B1 + 5; ++fv) gather(fv);
    }
    double gw[3] = {0.0, 0.0, 0.0};
    if (vn && f < p.Tstim) {
        const bool vr = n < p.npost;
        const double z = p.YfT[(size_t)f * p.ldy + n];
        double q = 0.0;
#pragma unroll
        for (int bt = 0; bt < 3; ++bt) {
            const double w = (vr && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
            q = fma(dt[bt], w, q);
            gw[bt] = dt[bt] * z;
        }
        p.QvT[(size_t)f * p.ldy + n] = q;
    }
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) red[wave][bt][lane] = gw[bt];
    __syncthreads();
    if (wave < 3 && vn) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][wave][lane];
        p.wpart[((size_t)blockIdx.x * 3 + wave) * p.ldy + n] = v;
    }
}

// C[m][n] = sum_k A[m sam + k sak] B[n sbn + k sbk] on the f64 MFMA, any strides (8-byte loads); stored at
// C[m scm + n scn].  A workgroup = 8 waves splitting K eight ways in chunks of 32 (lane group kk owns k = 32 u + 8 kk
// + v), each on a 16 (m) x 16 NT (n) tile; the eight partial tiles are added in a fixed order through LDS.  The
// operands of the next chunk are in flight during the MFMAs of the current one (the shapes here -- a few hundred
// tiles, K in the thousands -- are bound by the latency of their load rounds, not by the MFMA rate).
// grid = (ceil(M / 16), ceil(N / (16 NT)), batch).
template <int NT>
__global__ __launch_bounds__(512) void k_gemm_mfma(const double* __restrict__ A, long long sam, long long sak,
                                                   const double* __restrict__ Bm, long long sbn, long long sbk,
                                                   double* __restrict__ C, long long scm, long long scn,
                                                   int M, int Nn, int Kd, long long sAb = 0, long long sBb = 0,
                                                   long long sCb = 0)
{
    // (a batch of problems along blockIdx.z: operand / result strides sAb, sBb, sCb)
    A += (size_t)blockIdx.z * sAb;
    Bm += (size_t)blockIdx.z * sBb;
    C += (size_t)blockIdx.z * sCb;
    constexpr int NWG = 8, KC = 8;
    __shared__ double red[NWG - 1][NT][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16 * NT;
    const double* ap = A + (size_t)((m0 + i < M) ? m0 + i : M - 1) * sam;
    const double* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = Bm + (size_t)((n0 + 16 * t + i < Nn) ? n0 + 16 * t + i : Nn - 1) * sbn;
    d4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int nU = (Kd + 4 * KC - 1) / (4 * KC);
    double a[KC], b[NT][KC];
    auto fetch = [&](const int u) {
#pragma unroll
        for (int v = 0; v < KC; ++v) {
            const int k = 4 * KC * u + KC * kk + v;
            const bool ok = k < Kd;
            const size_t kc = ok ? k : 0;
            const double av = ap[kc * sak];
            a[v] = ok ? av : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double bv = bp[t][kc * sbk];
                b[t][v] = ok ? bv : 0.0;
            }
        }
    };
    if (wave < nU) fetch(wave);
    for (int u = wave; u < nU; u += NWG) {
        double ca[KC], cb[NT][KC];
#pragma unroll
        for (int v = 0; v < KC; ++v) {
            ca[v] = a[v];
#pragma unroll
            for (int t = 0; t < NT; ++t) cb[t][v] = b[t][v];
        }
        if (u + NWG < nU) fetch(u + NWG);
#pragma unroll
        for (int v = 0; v < KC; ++v)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[v], cb[t][v], acc[t], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[t][r];
#pragma unroll
                for (int w = 0; w < NWG - 1; ++w) v += red[w][t][r][lane];
                const int m = m0 + kk + 4 * r, n = n0 + 16 * t + i;      // D[kk + 4 r][i] (see the fused kernels' epilogue)
                if (m < M && n < Nn) C[(size_t)m * scm + (size_t)n * scn] = v;
            }
    }
}

// C[m][n] = sum_k A[m][k] B[n][k] for the two thin GEMMs of the separable stimulus (0.4 GFLOP each, 25 MB streamed once:
// what matters is how the bytes travel, not the MFMA rate).  A is k-contiguous and 16-byte aligned (lda even): a lane
// loads two consecutive k per request (global_load_dwordx4), rows 16 lanes apart, so a wave reads 64-byte runs of 16
// rows and its whole share of K is ONE contiguous run per row (DRAM pages stay open) -- against single 8-byte loads that
// were bound by the texture addresser (k-contiguous) or by DRAM page misses (m-contiguous, rows 24 KB apart).  K even.
// B: BK = 1 k-contiguous and aligned like A (B[n ldb + k]);  BK = 0 n-contiguous (B[k ldb + n], 128-byte lines).
// A workgroup = 8 waves sharing one 16 x 16 NT tile, each wave a contiguous eighth of K in chunks of 8 k (lane group
// kk owns k = 8 c + 2 kk, + 1); NB chunks are requested together (one latency per batch), the eight partial tiles are
// added in a fixed order through LDS.  Stored at C[m scm + n scn].  grid = (ceil(M / 16), ceil(N / (16 NT))).
template <int NT, int BK, int NB>
__global__ __launch_bounds__(512) void k_gemm_kc(const double* __restrict__ A, long long lda,
                                                 const double* __restrict__ Bm, long long ldb,
                                                 double* __restrict__ C, long long scm, long long scn,
                                                 int M, int Nn, int Kd)
{
    constexpr int NWG = 8;
    __shared__ double red[NWG - 1][NT][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16 * NT;
    const double* ap = A + (size_t)((m0 + i < M) ? m0 + i : M - 1) * lda;
    const double* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = (n0 + 16 * t + i < Nn) ? n0 + 16 * t + i : Nn - 1;
        bp[t] = BK ? Bm + (size_t)n * ldb : Bm + n;
    }
    d4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int nC = (Kd + 7) / 8;                          // chunks of 8 k
    const int cpw = (nC + NWG - 1) / NWG;
    const int c_beg = wave * cpw, c_end = (c_beg + cpw < nC) ? c_beg + cpw : nC;
    for (int cb = c_beg; cb < c_end; cb += NB) {
        pgl_d2 a2[NB], b2[NB][NT];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int k = 8 * (cb + c) + 2 * kk;
            const bool ok = (cb + c < c_end) && k < Kd;   // (Kd is even: a pair is inside or outside as a whole)
            const size_t kc = ok ? k : 0;                  // (a clamped address for the tail)
            const pgl_d2 av = *reinterpret_cast<const pgl_d2*>(ap + kc);
            a2[c].x = ok ? av.x : 0.0;
            a2[c].y = ok ? av.y : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pgl_d2 bv;
                if (BK) {
                    bv = *reinterpret_cast<const pgl_d2*>(bp[t] + kc);
                } else {
                    bv.x = bp[t][kc * ldb];
                    bv.y = bp[t][(kc + 1) * ldb];
                }
                b2[c][t].x = ok ? bv.x : 0.0;
                b2[c][t].y = ok ? bv.y : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[c].x, b2[c][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[c].y, b2[c][t].y, acc[t], 0, 0, 0);
            }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[t][r];
#pragma unroll
                for (int w = 0; w < NWG - 1; ++w) v += red[w][t][r][lane];
                const int m = m0 + kk + 4 * r, n = n0 + 16 * t + i;
                if (m < M && n < Nn) C[(size_t)m * scm + (size_t)n * scn] = v;
            }
    }
}

// ---------------------------------------------------------------------------
// Spike-triggered average (pyglm/utils/sta.py:6-85) from the event lists:
//   A[i,l,d] = sum_t S[t,n_i] * istim[t-l, d] / sum_t S[t,n_i],  l = 0..L-1 (t-l < 0 -> 0)
// with istim the stimulus interpolated to the bin grid and divided by dt_stim/dt (sta.py:30-41).
// The reference forms a dense (nT, L*D) lag matrix and a gemv per neuron; here only the bins that
// hold spikes are touched.  grid = (ceil(L*D/256), nSel, echunks); block c of the z axis handles an
// equal share of the neuron's events and writes a partial, summed in fixed order by k_sta_finish.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sta(const int2* __restrict__ spk, const int* __restrict__ eoff,
                                             const double* __restrict__ istim, int D, int L,
                                             double* __restrict__ part)
{
    __shared__ int2 ev[256];
    const int i = blockIdx.y, c = blockIdx.z, nSel = gridDim.y, echunks = gridDim.z;
    const long long LD = (long long)L * D;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(o / D);
    const int d = (int)(o - (long long)l * D);
    const int e0 = eoff[2 * i], e1 = eoff[2 * i + 1];
    const int per = (e1 - e0 + echunks - 1) / echunks;
    const int a = e0 + c * per;
    const int b = min(e1, a + per);
    double acc = 0.0;
    for (int base = a; base < b; base += 256) {
        const int e = base + (int)threadIdx.x;
        ev[threadIdx.x] = (e < b) ? spk[e] : make_int2(0, 0);
        __syncthreads();
        const int m = min(256, b - base);
        if (o < LD) {
            for (int j = 0; j < m; ++j) {
                const int2 q = ev[j];
                const long long tt = (long long)q.x - l;
                if (tt >= 0) acc = fma((double)q.y, istim[tt * D + d], acc);
            }
        }
        __syncthreads();
    }
    if (o < LD) part[((size_t)c * nSel + i) * LD + o] = acc;
}

// A[i][o] = scale * sum_c part[c][i][o] / count_i, count_i = sum of the event counts of neuron i
// (0/0 = NaN for a silent neuron, like the reference's division, sta.py:79-80)
__global__ __launch_bounds__(256) void k_sta_finish(const double* __restrict__ part,
                                                    const int2* __restrict__ spk,
                                                    const int* __restrict__ eoff, long long LD,
                                                    int echunks, double scale, double* __restrict__ A)
{
    __shared__ double red[256];
    const int i = blockIdx.y, nSel = gridDim.y;
    double cnt = 0.0;
    for (int e = eoff[2 * i] + (int)threadIdx.x; e < eoff[2 * i + 1]; e += 256) cnt += (double)spk[e].y;
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    cnt = red[0];
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= LD) return;
    double acc = 0.0;
    for (int c = 0; c < echunks; ++c) acc += part[((size_t)c * nSel + i) * LD + o];
    A[(size_t)i * LD + o] = acc * scale / cnt;
}

// Spike-triggered average at the stimulus FRAME rate (dt_stim = q dt, q integer): np.interp makes the stimulus of bin tt the
// mix (1 - a) stim[f] + a stim[f + 1], f = tt / q, a = (tt % q) / q (the last frame beyond the end), so
//   A[i][l][:] = sum_f Wt[i][l][f] stim[f][:],   Wt[i][l][f] = scale_i * sum over the events (t, c) of neuron i of
//                c ((1 - a) [tt / q == f] + a [tt / q + 1 == f]),  tt = t - l >= 0
// -- a (nSel L) x Tstim weight matrix from the event lists (this kernel: one thread per cell, the few events that can
// reach it found by bisection in the neuron's sorted list) and ONE thin GEMM with the raw stimulus (k_gemm_kc) instead of a
// gather of L x D doubles per spike from the interpolated stimulus: 138 -> 5 ms at 64 neurons x 300 lags x 1024 pixels.
__global__ __launch_bounds__(256) void k_sta_weights(const int2* __restrict__ spk, const int* __restrict__ eoff,
                                                     const double* __restrict__ scale, int L, long long Tstim, int q,
                                                     double* __restrict__ Wt)
{
    const int i = blockIdx.z, l = blockIdx.y;
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= Tstim) return;
    const int e0 = eoff[2 * i], e1 = eoff[2 * i + 1];
    // events with tt = t - l in [(f - 1) q, (f + 1) q); the last frame also takes everything beyond it
    const long long lo_t = (f - 1) * (long long)q + l, hi_t = (f + 1 >= Tstim) ? (1ll << 62) : (f + 1) * (long long)q + l;
    int a = e0, b = e1;
    while (a < b) {                                       // first event with t >= lo_t (and t >= l)
        const int m = (a + b) >> 1;
        if ((long long)spk[m].x < lo_t) a = m + 1; else b = m;
    }
    double acc = 0.0;
    for (int e = a; e < e1; ++e) {
        const int2 ev = spk[e];
        if ((long long)ev.x >= hi_t) break;
        const long long tt = (long long)ev.x - l;
        if (tt < 0) continue;
        const long long fr = tt / q;
        const double al = (double)(tt - fr * q) / (double)q;
        double w;
        if (fr >= Tstim - 1) w = (f == Tstim - 1) ? 1.0 : 0.0;       // np.interp holds the last frame
        else w = (fr == f) ? 1.0 - al : ((fr + 1 == f) ? al : 0.0);
        acc = fma((double)ev.y, w, acc);
    }
    Wt[((size_t)i * L + l) * Tstim + f] = acc * scale[i];
}

// transpose of the uint8 count matrix: ST[n][t] = S[t][n]
__global__ void k_transpose_u8(const uint8_t* __restrict__ S, uint8_t* __restrict__ ST,
                               long long nT, int N)
{
    __shared__ uint8_t tile[64][65];
    const long long t0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int tt = i / 64, nn = i % 64;
        const long long t = t0 + tt;
        const int n = n0 + nn;
        tile[tt][nn] = (t < nT && n < N) ? S[t * N + n] : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int nn = i / 64, tt = i % 64;
        const long long t = t0 + tt;
        const int n = n0 + nn;
        if (t < nT && n < N) ST[(size_t)n * nT + t] = tile[tt][nn];
    }
}

// ---------------------------------------------------------------------------
// Leading singular pair (u_0, s_0, v_0) of a batch of matrices -- all that initialize_stim_with_sta keeps of
// np.linalg.svd(STA) (smart_init.py:66-72: the best rank-1 space x time factorisation of a neuron's spike-triggered
// average).  On the device, with this library's own kernels: the Gram matrix of the smaller side (k_gemm_mfma, batched),
// repeated squaring with a trace normalisation (G^(2^16): every other eigen-direction is down by (l_2 / l_1)^65536), the
// column with the largest diagonal entry as the eigenvector, then two steps of the alternating iteration on A itself.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double pgl_block_sum1024(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
// G[b] <- G[b] / trace(G[b]) (symmetric positive semi-definite: its eigenvalues then lie in [0, 1] and the largest is >= 1 / m)
__global__ __launch_bounds__(1024) void k_lsp_trace_scale(double* __restrict__ G, const int m, const long long sb)
{
    __shared__ double red[16];
    double* g = G + (size_t)blockIdx.x * sb;
    double t = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) t += g[(size_t)i * m + i];
    t = pgl_block_sum1024(t, red);
    if (!(t > 0.0) || t - t != 0.0) return;               // the zero matrix (no spikes) stays
    const double inv = 1.0 / t;
    for (int i = threadIdx.x; i < m * m; i += blockDim.x) g[i] *= inv;
}
// x[b] <- the column of G[b] with the largest diagonal entry, normalised (e_0 for the zero matrix)
__global__ __launch_bounds__(1024) void k_lsp_pick(const double* __restrict__ G, const int m, const long long sb,
                                                   double* __restrict__ x)
{
    __shared__ double red[16];
    __shared__ double bestv[16];
    __shared__ int besti[16];
    const double* g = G + (size_t)blockIdx.x * sb;
    double bv = -1.0;
    int bi = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double d = g[(size_t)i * m + i];
        if (d > bv) { bv = d; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bestv[threadIdx.x >> 6] = bv; besti[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = bestv[0]; bi = besti[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
    double* xo = x + (size_t)blockIdx.x * m;
    double ss = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double v = g[(size_t)i * m + bi];
        ss = fma(v, v, ss);
    }
    ss = pgl_block_sum1024(ss, red);
    const double inv = (ss > 0.0) ? 1.0 / sqrt(ss) : 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) xo[i] = (ss > 0.0) ? g[(size_t)i * m + bi] * inv : (i == 0 ? 1.0 : 0.0);
}
// v[b] <- v[b] / |v[b]|, nrm[b] = |v[b]| (a zero vector stays)
__global__ __launch_bounds__(1024) void k_lsp_normalize(double* __restrict__ v, const int len, double* __restrict__ nrm)
{
    __shared__ double red[16];
    double* x = v + (size_t)blockIdx.x * len;
    double ss = 0.0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) ss = fma(x[i], x[i], ss);
    ss = pgl_block_sum1024(ss, red);
    const double n = sqrt(ss);
    if (nrm && threadIdx.x == 0) nrm[blockIdx.x] = n;
    if (!(n > 0.0)) return;
    const double inv = 1.0 / n;
    for (int i = threadIdx.x; i < len; i += blockDim.x) x[i] *= inv;
}
