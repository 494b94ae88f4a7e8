// lock-step BFGS row kernels: k_bfgs_*
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
// ---------------------------------------------------------------------------
// Lock-step BFGS (inference/batched_bfgs.py): the per-neuron optimiser state machines of all M neurons of a shard as a
// handful of row kernels -- one workgroup per neuron row -- around the fused ll+grad launch.  The reference calls
// scipy.optimize.minimize(method="bfgs") per neuron (coord_descent.py:194-199); the row kernels run the same algorithm
// for every neuron at once: BFGS from H = I, More'-Thuente line search for the strong Wolfe conditions
// (pglm_linesearch.h: scipy's DCSRCH with scipy's constants and first trial step), termination on max|g| <= gtol or
// maxiter iterations.  Where scipy gives up ("precision loss": the search reports a warning) the row takes the best
// sufficient-decrease step of that search if there is one, else restarts once from steepest descent, then freezes.
// All state lives in ONE device block of doubles (flags and counters included), laid out by pgl_bfgs_view; the dense
// inverse Hessians H (M, P, ld) are the caller's buffer and are touched by k_bfgs_hmul only: ONE read-modify-write
// pass per accepted iteration applies the pending rank-3 update H += U V^T of the previous iteration and multiplies
// by the new gradient; a (re)started H = hscale * I is never materialised before its first update.
// ---------------------------------------------------------------------------
#include "pglm_linesearch.h"

struct BfgsView {
    int M, P;
    double *X, *g, *p, *Hg, *s, *y, *t, *Xb, *gb;    // (M, P)
    double *U, *V;                                   // (M, P, 3): H += U V^T is the BFGS update
    double *f, *fprev, *alpha, *slope, *rho, *hscale, *iters, *restarts, *active, *frozen, *acc, *upd, *stall, *ident,
           *pend, *fb, *nfev, *hk;                   // (M)
    double* ls;                                      // (PGL_LS_NDOUBLES, M): line-search state, field-major
};
#define PGL_BFGS_NVEC 9
#define PGL_BFGS_NSCAL 18
#define PGL_LS_FTOL 1e-4
#define PGL_LS_GTOL 0.9
#define PGL_LS_XTOL 1e-14
#define PGL_LS_STPMIN 1e-100
#define PGL_LS_STPMAX 1e100
__host__ __device__ inline size_t pgl_bfgs_doubles(int M, int P)
{
    return (size_t)M * P * (PGL_BFGS_NVEC + 6) + (size_t)M * (PGL_BFGS_NSCAL + PGL_LS_NDOUBLES);
}
__host__ __device__ inline BfgsView pgl_bfgs_view(double* st, int M, int P)
{
    BfgsView v;
    const size_t MP = (size_t)M * P;
    v.M = M; v.P = P;
    v.X = st; v.g = st + MP; v.p = st + 2 * MP; v.Hg = st + 3 * MP; v.s = st + 4 * MP; v.y = st + 5 * MP;
    v.t = st + 6 * MP; v.Xb = st + 7 * MP; v.gb = st + 8 * MP; v.U = st + 9 * MP; v.V = st + 12 * MP;
    double* q = st + 15 * MP;
    v.f = q; v.fprev = q + M; v.alpha = q + 2 * M; v.slope = q + 3 * M; v.rho = q + 4 * M; v.hscale = q + 5 * M;
    v.iters = q + 6 * M; v.restarts = q + 7 * M; v.active = q + 8 * M; v.frozen = q + 9 * M; v.acc = q + 10 * M;
    v.upd = q + 11 * M; v.stall = q + 12 * M; v.ident = q + 13 * M; v.pend = q + 14 * M; v.fb = q + 15 * M;
    v.nfev = q + 16 * M; v.hk = q + 17 * M;
    v.ls = q + (size_t)PGL_BFGS_NSCAL * M;
    return v;
}
#define PGL_LS_FIELDS(F) F(stp, 0) F(finit, 1) F(ginit, 2) F(gtest, 3) F(stx, 4) F(fx, 5) F(gx, 6) F(sty, 7) F(fy, 8) \
    F(gy, 9) F(stmin, 10) F(stmax, 11) F(width, 12) F(width1, 13) F(brackt, 14) F(stage, 15) F(nfev, 16) F(moved, 17)
__device__ __forceinline__ void pgl_ls_load(const BfgsView& v, int r, PglLs* s)
{
#define PGL_LS_LD(name, k) s->name = v.ls[(size_t)k * v.M + r];
    PGL_LS_FIELDS(PGL_LS_LD)
#undef PGL_LS_LD
}
__device__ __forceinline__ void pgl_ls_store(const BfgsView& v, int r, const PglLs* s)
{
#define PGL_LS_ST(name, k) v.ls[(size_t)k * v.M + r] = s->name;
    PGL_LS_FIELDS(PGL_LS_ST)
#undef PGL_LS_ST
}

// sum / max over the 256 threads of a block, result in every thread (fixed order)
// (blocks of more than 256 threads -- the one-launch iteration k_bfgs_step<1024> -- run their row loops on the first 256
//  threads only, so a row's sums are the same numbers whichever kernel computes them; the other threads pass through the
//  barriers with nothing to add)
__device__ __forceinline__ double pgl_blk_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double pgl_blk_max(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// three sums (or two sums and a maximum) behind ONE pair of barriers: the same butterflies and the same combination of
// the four waves' values as three calls of the above -- the same numbers, a third of the barriers
__device__ __forceinline__ void pgl_blk_sum3(double& a, double& b, double& c, double* red, const bool cmax = false)
{
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
        const double co = __shfl_xor(c, o, 64);
        c = cmax ? fmax(c, co) : c + co;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) {
        const int w = threadIdx.x >> 6;
        red[w] = a;
        red[4 + w] = b;
        red[8 + w] = c;
    }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
    c = cmax ? fmax(fmax(red[8], red[9]), fmax(red[10], red[11])) : (red[8] + red[9]) + (red[10] + red[11]);
}
// first column of a thread's stride-256 walk over a row of P numbers (threads beyond the first 256: none)
__device__ __forceinline__ int pgl_row_c0(const int tid, const int P) { return tid < 256 ? tid : P; }

// start of a fit: X, f, g of every row are in place; H = I (not materialised), steepest-descent direction, scipy's
// first trial step min(1, 1.01 / |g|) (its old_old_fval = f + |g| / 2), rows with max|g| <= gtol never start
__global__ __launch_bounds__(256) void k_bfgs_init(const BfgsView v, const double gtol)
{
    __shared__ double red[12];
    const int r = blockIdx.x, tid = threadIdx.x, P = v.P;
    const size_t o = (size_t)r * P;
    double gg = 0.0, gmax = 0.0;
    for (int c = tid; c < P; c += 256) {
        const double gc = v.g[o + c];
        v.Hg[o + c] = gc;
        v.p[o + c] = -gc;
        gg = fma(gc, gc, gg);
        gmax = fmax(gmax, fabs(gc));
    }
    gg = pgl_blk_sum(gg, red);
    gmax = pgl_blk_max(gmax, red);
    if (tid == 0) {
        const double f = v.f[r], fprev = f + sqrt(gg) / 2.0, slope = -gg;
        v.fprev[r] = fprev; v.slope[r] = slope; v.rho[r] = 0.0; v.hscale[r] = 1.0; v.iters[r] = 0.0; v.restarts[r] = 0.0;
        v.frozen[r] = 0.0; v.acc[r] = 0.0; v.upd[r] = 0.0; v.stall[r] = 0.0; v.ident[r] = 1.0; v.pend[r] = 0.0;
        v.fb[r] = f; v.nfev[r] = 0.0; v.hk[r] = 0.0;
        v.active[r] = gmax > gtol ? 1.0 : 0.0;
        PglLs s;
        const double a0 = pgl_ls_first_step(f, fprev, slope);
        pgl_ls_start(&s, a0, f, slope, PGL_LS_FTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
        pgl_ls_store(v, r, &s);
        v.alpha[r] = a0;
    }
}

// trial points of the listed rows: Xt[j] = X[r] + alpha[r] p[r], r = rows[j] (null: r = j)
__global__ __launch_bounds__(256) void k_bfgs_trial(const BfgsView v, const int* __restrict__ rows,
                                                    double* __restrict__ Xt)
{
    const int j = blockIdx.x, r = rows ? rows[j] : j;
    const double a = v.alpha[r];
    for (int c = threadIdx.x; c < v.P; c += 256)
        Xt[(size_t)j * v.P + c] = fma(a, v.p[(size_t)r * v.P + c], v.X[(size_t)r * v.P + c]);
}

struct BfgsPrior {
    int N, B, Dstim, kind;                // kind: 0 Gaussian, 1 group lasso on the impulse weights (priors.py:139 / 202)
    double mu_b, sg_b, stim_sigma, mu, sigma, lam;
};

// f = -(ll + log prior), g = -(grad ll + grad log prior) of one trial row x = [bias, w_stim, w_ir] (the packing that IS the
// device's theta row); fit_glm's NaN rules: f NaN -> 1e16, any NaN in a gradient row -> zero row
// (coord_descent.py:170-182).  In place: *llj -> f, g (grad) -> g.  Whole block; row loops on its first 256 threads.
__device__ __forceinline__ void pgl_bfgs_objective_row(const int P, const double* __restrict__ x, double* __restrict__ g,
                                                       double* __restrict__ llj, const BfgsPrior& q, double* red, const int tid)
{
    double lp = 0.0;
    bool bad = false;
    if (tid == 0) {                                                            // bias.py:33
        const double d = x[0] - q.mu_b;
        lp += -0.5 / (q.sg_b * q.sg_b) * d * d;
        const double gv = -(g[0] - d / (q.sg_b * q.sg_b));
        bad = bad || (gv != gv);
        g[0] = gv;
    }
    for (int c = 1 + pgl_row_c0(tid, q.Dstim); c < 1 + q.Dstim; c += 256) {    // bkgd.py:76
        const double w = x[c], is2 = 1.0 / (q.stim_sigma * q.stim_sigma);
        lp += -0.5 * is2 * w * w;
        const double gv = -(g[c] - w * is2);
        bad = bad || (gv != gv);
        g[c] = gv;
    }
    const int o = 1 + q.Dstim;
    for (int n = pgl_row_c0(tid, q.N); n < q.N; n += 256) {                    // one presynaptic group per thread
        const double* w = x + o + n * q.B;
        double* gw = g + o + n * q.B;
        if (q.kind == 1) {                                                     // priors.py:202
            double z[PGL_MAXB], ss = 0.0;
            for (int b = 0; b < q.B; ++b) {
                z[b] = (w[b] - q.mu) / q.sigma;
                ss += z[b] * z[b];
            }
            const double nrm = sqrt(ss);
            lp -= q.lam * nrm;
            for (int b = 0; b < q.B; ++b) {
                const double gv = -(gw[b] - q.lam * z[b] / nrm / q.sigma);     // 0/0 -> NaN like the host prior
                bad = bad || (gv != gv);
                gw[b] = gv;
            }
        } else {                                                               // priors.py:139
            const double is2 = 1.0 / (q.sigma * q.sigma);
            for (int b = 0; b < q.B; ++b) {
                const double d = w[b] - q.mu;
                lp += -0.5 * is2 * d * d;
                const double gv = -(gw[b] - d * is2);
                bad = bad || (gv != gv);
                gw[b] = gv;
            }
        }
    }
    const double lpt = pgl_blk_sum(lp, red);
    const bool anybad = pgl_blk_max(bad ? 1.0 : 0.0, red) > 0.0;
    if (tid == 0) {
        const double fv = -(*llj + lpt);
        *llj = (fv != fv) ? 1e16 : fv;
    }
    if (anybad)
        for (int c = pgl_row_c0(tid, P); c < P; c += 256) g[c] = 0.0;
}

// (pgl_bfgs_objective_dev: the objective of rows that are not in a search -- the starting point of a fit)
__global__ __launch_bounds__(256) void k_bfgs_objective(const int P, const double* __restrict__ Xt,
                                                        double* __restrict__ ll, double* __restrict__ grad,
                                                        const BfgsPrior q)
{
    __shared__ double red[12];
    const int j = blockIdx.x;
    pgl_bfgs_objective_row(P, Xt + (size_t)j * P, grad + (size_t)j * P, ll + j, q, red, (int)threadIdx.x);
}

// One line-search step of a row whose search is running (r: its row of the state, xt / ftj / gt: the evaluated trial):
// phi'(alpha) = g_trial . p, then the More'-Thuente state machine.  Outcomes: another trial step (alpha[r]); the trial
// satisfies the strong Wolfe conditions and the row takes it (X, f, g; s, y, rho left behind, acc = 1); or the search cannot
// make progress -- scipy stops there ("precision loss"; identical iterates up to that first warning only) -- then the best
// sufficient-decrease point of this search is taken if there is one (the trial itself, or the best step saved so far, Xb / gb
// -- judged by ITS step length, also when the search is cut off by max_trials on a call in which the trial became the best
// step), else stall = 1 (the update phase restarts or freezes the row).
__device__ __forceinline__ void pgl_bfgs_linesearch_row(const BfgsView& v, const int r, const double* __restrict__ xt,
                                                        const double* __restrict__ ftj, const double* __restrict__ gt,
                                                        const int max_trials, double* red, int* dec, const int tid)
{
    const int P = v.P;
    if (v.active[r] == 0.0) return;
    const size_t o = (size_t)r * P;
    double dp = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) dp = fma(gt[c], v.p[o + c], dp);
    dp = pgl_blk_sum(dp, red);
    if (tid == 0) {
        // (the state machine is wave-uniform scalar code; routed through a vector register index so that its ~40 doubles
        // live in VGPRs instead of spilling the scalar register file)
        int rv = r;
        asm volatile("" : "+v"(rv));
        const int r = rv;
        PglLs s;
        pgl_ls_load(v, r, &s);
        const double stp = s.stp, stx_prev = s.stx;
        const double f = *ftj;
        // an infinite objective or slope ends the search like scipy's ("WARN": its fallback search fails on inf as well);
        // NaN never arrives here (fit_glm's rule: 1e16 and a zero gradient, applied by the objective)
        int rc = PGL_LS_WARNING;
        if (f - f == 0.0 && dp - dp == 0.0)
            rc = pgl_ls_step(&s, f, dp, PGL_LS_FTOL, PGL_LS_GTOL, PGL_LS_XTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
        else s.moved = 0.0;
        if (rc == PGL_LS_EVALUATE && s.nfev >= (double)max_trials) rc = PGL_LS_WARNING;
        int src = 0;                                         // 1: take the trial, 2: take the saved best step
        const int moved = s.moved != 0.0;
        if (rc == PGL_LS_CONVERGED) src = 1;
        else if (rc == PGL_LS_WARNING) {
            // (Xb / gb / fb still hold the best step BEFORE this call -- they are only overwritten on EVALUATE -- and
            //  stx_prev is its step length: s.stx is the trial's when it has just become the best step)
            const bool okT = stp > 0.0 && f <= s.finit + stp * s.gtest && f < s.finit;
            const bool okB = stx_prev > 0.0 && v.fb[r] <= s.finit + stx_prev * s.gtest && v.fb[r] < s.finit;
            if (okT && (!okB || f <= v.fb[r])) src = 1;
            else if (okB) src = 2;
            if (src == 0) v.stall[r] = 1.0;
        } else {
            pgl_ls_store(v, r, &s);
            v.alpha[r] = s.stp;
            if (moved) v.fb[r] = f;
        }
        v.nfev[r] += 1.0;
        dec[0] = rc; dec[1] = src; dec[2] = moved;
    }
    __syncthreads();
    const int rc = dec[0], src = dec[1];
    if (rc == PGL_LS_EVALUATE) {
        if (dec[2])                                          // the trial is the best step so far: keep its point and gradient
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                v.Xb[o + c] = xt[c];
                v.gb[o + c] = gt[c];
            }
        return;
    }
    if (src == 0) return;
    const double* xs = src == 1 ? xt : v.Xb + o;
    const double* gs = src == 1 ? gt : v.gb + o;
    double sy = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
        const double xn = xs[c], gn = gs[c];
        const double s = xn - v.X[o + c], y = gn - v.g[o + c];
        v.s[o + c] = s;
        v.y[o + c] = y;
        v.X[o + c] = xn;
        v.g[o + c] = gn;
        sy = fma(s, y, sy);
    }
    sy = pgl_blk_sum(sy, red);
    if (tid == 0) {
        const double fn = src == 1 ? *ftj : v.fb[r];
        v.fprev[r] = v.f[r];
        v.f[r] = fn;
        v.acc[r] = 1.0;
        const double rho = 1.0 / sy;
        const bool u = sy > 0.0 && rho - rho == 0.0;         // curvature condition holds (always after a Wolfe step)
        v.upd[r] = u ? 1.0 : 0.0;
        v.rho[r] = u ? rho : 0.0;
    }
}

// t = H g for the listed rows that have just taken a step (acc = 1), in the same pass over H that applies the pending
// rank-3 update of the previous iteration: H <- H + U V^T (pend), t = H g.  A row whose H is still hscale * I (ident)
// is materialised here together with its first update; without a pending update it is not touched at all
// (k_bfgs_update uses t = hscale * g).  Grid (ceil(P / 32), L); a wave owns 8 rows of H, lanes run along the columns
// in 16-byte pieces (ld even).  Traffic: one read + one write of P x ld doubles per row and accepted iteration.
#define PGL_HM_ROWS 8
__global__ __launch_bounds__(256) void k_bfgs_hmul(const BfgsView v, const int* __restrict__ rows,
                                                   double* __restrict__ H, const int ld)
{
    const int j = blockIdx.y, r = rows ? rows[j] : j;
    if (v.acc[r] == 0.0) return;
    const bool ident = v.ident[r] != 0.0, pend = v.pend[r] != 0.0;
    if (ident && !pend) return;
    const int P = v.P, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wv) * PGL_HM_ROWS;
    if (i0 >= P) return;
    const double hs = v.hscale[r];
    const double* __restrict__ g = v.g + (size_t)r * P;
    const double* __restrict__ U = v.U + (size_t)r * P * 3;
    const double* __restrict__ V = v.V + (size_t)r * P * 3;
    double* Hr = H + (size_t)r * P * ld;
    double acc[PGL_HM_ROWS], u[PGL_HM_ROWS][3];
#pragma unroll
    for (int k = 0; k < PGL_HM_ROWS; ++k) {
        acc[k] = 0.0;
        const int i = min(i0 + k, P - 1);
#pragma unroll
        for (int e = 0; e < 3; ++e) u[k][e] = pend ? U[(size_t)i * 3 + e] : 0.0;
    }
    for (int c = 2 * lane; c < ld; c += 128) {
        const bool in0 = c < P, in1 = c + 1 < P;
        const double g0 = in0 ? g[c] : 0.0, g1 = in1 ? g[c + 1] : 0.0;
        double va[3], vb[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            va[e] = (pend && in0) ? V[(size_t)c * 3 + e] : 0.0;
            vb[e] = (pend && in1) ? V[(size_t)(c + 1) * 3 + e] : 0.0;
        }
        double2 h[PGL_HM_ROWS];
#pragma unroll
        for (int k = 0; k < PGL_HM_ROWS; ++k) {
            const int i = i0 + k;
            if (ident || i >= P) h[k] = make_double2(i == c ? hs : 0.0, i == c + 1 ? hs : 0.0);
            else h[k] = *reinterpret_cast<const double2*>(Hr + (size_t)i * ld + c);
        }
#pragma unroll
        for (int k = 0; k < PGL_HM_ROWS; ++k) {
            const int i = i0 + k;
            if (pend) {
                h[k].x += u[k][0] * va[0] + u[k][1] * va[1] + u[k][2] * va[2];
                h[k].y += u[k][0] * vb[0] + u[k][1] * vb[1] + u[k][2] * vb[2];
                if (i < P) *reinterpret_cast<double2*>(Hr + (size_t)i * ld + c) = h[k];
            }
            acc[k] = fma(h[k].x, g0, fma(h[k].y, g1, acc[k]));
        }
    }
#pragma unroll
    for (int k = 0; k < PGL_HM_ROWS; ++k) {
        double a = acc[k];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0 && i0 + k < P) v.t[(size_t)r * P + i0 + k] = a;
    }
}

// The same product with the inverse Hessian kept IMPLICIT: H = hscale I + sum_{j < hk} U_j V_j^T over every update so far.
// The rank-3 factors share two vectors -- U_j = (c0 s, -rho Hy, -rho s), V_j = (s, s, Hy) -- so the history holds (s_j, Hy_j)
// [row][j][2][P] and (c0_j, rho_j) [row][j][2] (k_bfgs_update appends them), and
//   U_j V_j^T g = (c0 a - rho b) s_j + (-rho a) Hy_j,   a = s_j . g,  b = Hy_j . g.
// 4 hk P numbers per row and product instead of the 2 P^2 of the dense form -- less traffic while hk <= P / 2, which is
// where fits live (C3 converges in 22 iterations; at the C5 stress shape, P = 1220, all 225 iterations read 5 x less on
// average), and no P^2 memory (wide populations).  The driver's default.  Two kernels:
//   k_bfgs_hdots: ab[row][j] = the two coefficients, one wave per (row, j), fixed-order wave reduction;
//   k_bfgs_hcomb: t = hscale g + sum_j ab[j][0] s_j + ab[j][1] Hy_j for 64 components per block; wave w of 8 takes
//                 j = w, w + 8, ..., the eight partial sums are added in wave order.
__global__ __launch_bounds__(256) void k_bfgs_hdots(const BfgsView v, const int* __restrict__ rows,
                                                    const double* __restrict__ Wh, const double* __restrict__ cs, const int Kmax,
                                                    double* __restrict__ ab)
{
    const int jr = blockIdx.y, r = rows ? rows[jr] : jr;
    if (v.acc[r] == 0.0) return;
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), P = v.P;
    if (j >= (int)v.hk[r]) return;
    const double* __restrict__ g = v.g + (size_t)r * P;
    const double* __restrict__ W = Wh + ((size_t)r * Kmax + j) * 2 * P;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    int i = lane;
    for (; i + 64 < P; i += 128) {
        const double g0 = g[i], g1 = g[i + 64];
        const double s0 = W[i], s1 = W[i + 64], h0 = W[P + i], h1 = W[P + i + 64];
        a0 = fma(s0, g0, a0);
        a1 = fma(s1, g1, a1);
        b0 = fma(h0, g0, b0);
        b1 = fma(h1, g1, b1);
    }
    if (i < P) {
        a0 = fma(W[i], g[i], a0);
        b0 = fma(W[P + i], g[i], b0);
    }
    double a = a0 + a1, b = b0 + b1;
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0) {
        const size_t q = ((size_t)r * Kmax + j) * 2;
        const double c0 = cs[q], rho = cs[q + 1];
        ab[q] = c0 * a - rho * b;
        ab[q + 1] = -rho * a;
    }
}
__global__ __launch_bounds__(512) void k_bfgs_hcomb(const BfgsView v, const int* __restrict__ rows,
                                                    const double* __restrict__ Wh, const int Kmax, const double* __restrict__ ab)
{
    __shared__ double part[8][64];
    const int jr = blockIdx.y, r = rows ? rows[jr] : jr;
    if (v.acc[r] == 0.0) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, P = v.P, K = (int)v.hk[r];
    if (K == 0) return;                                      // (no history: k_bfgs_update uses t = hscale g)
    const int i = blockIdx.x * 64 + lane;
    const bool in = i < P;
    const double* __restrict__ W = Wh + (size_t)r * Kmax * 2 * P + (in ? i : 0);
    const double* __restrict__ c = ab + (size_t)r * Kmax * 2;
    double acc0 = 0.0, acc1 = 0.0;
    int j = w;
    for (; j + 8 < K; j += 16) {
        const double* u0 = W + (size_t)j * 2 * P;
        const double* u1 = W + (size_t)(j + 8) * 2 * P;
        const double s0 = u0[0], h0 = u0[P], s1 = u1[0], h1 = u1[P];
        acc0 = fma(s0, c[2 * j], acc0);
        acc0 = fma(h0, c[2 * j + 1], acc0);
        acc1 = fma(s1, c[2 * j + 16], acc1);
        acc1 = fma(h1, c[2 * j + 17], acc1);
    }
    if (j < K) {
        const double* u0 = W + (size_t)j * 2 * P;
        acc0 = fma(u0[0], c[2 * j], acc0);
        acc0 = fma(u0[P], c[2 * j + 1], acc0);
    }
    part[w][lane] = acc0 + acc1;
    __syncthreads();
    if (w == 0 && in) {
        double t = v.hscale[r] * v.g[(size_t)r * P + i];
        #pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][lane];
        v.t[(size_t)r * P + i] = t;
    }
}

// After the line-search step (and t = H g_new for the rows that moved): the rank-3 factors of the inverse-Hessian
// update  H_new = (I - rho s y^T) H (I - rho y s^T) + rho s s^T = H + U V^T,  H_new g_new, the next direction and the
// start of its line search, restart / freeze of stalled rows, convergence flags.  init_scaling != 0: the first update
// after a (re)start is preceded by H <- (s.y / y.y) I (Nocedal & Wright (6.20); not scipy's behaviour).
__device__ __forceinline__ void pgl_bfgs_update_row(const BfgsView& v, const int r, const double gtol, const int maxiter,
                                                    const int init_scaling, double* __restrict__ Wh,
                                                    double* __restrict__ cs, const int Kmax, double* red, const int tid)
{
    const int P = v.P;
    const size_t o = (size_t)r * P;
    const bool act = v.active[r] != 0.0, a = v.acc[r] != 0.0, u = v.upd[r] != 0.0, st = v.stall[r] != 0.0;
    if (!act || (!a && !st)) return;                         // finished, or in the middle of a line search
    double* U = v.U + o * 3;
    double* V = v.V + o * 3;
    bool ident = v.ident[r] != 0.0, pend = v.pend[r] != 0.0;
    double hs = v.hscale[r];
    double restarts = v.restarts[r], iters = v.iters[r];
    bool frozen = false, again = false, hist_add = false;
    if (a) {
        const bool lazy = ident && !pend;                    // H = hs * I: k_bfgs_hmul / k_bfgs_hcomb left t alone
        iters += 1.0;
        restarts = 0.0;
        if (u) {
            const double rho = v.rho[r];
            double sc = 1.0;
            if (init_scaling && lazy) {                      // H <- (s.y / y.y) I before the first update
                double yy = 0.0;
                for (int c = pgl_row_c0(tid, P); c < P; c += 256) yy = fma(v.y[o + c], v.y[o + c], yy);
                yy = pgl_blk_sum(yy, red);
                const double gam = (1.0 / rho) / yy;
                if (gam > 0.0 && gam - gam == 0.0) { sc = gam / hs; hs = gam; }
            }
            double yHy = 0.0, vg0 = 0.0, vg2 = 0.0;
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                const double tc = lazy ? hs * v.g[o + c] : v.t[o + c];
                const double Hy = tc - sc * v.Hg[o + c];
                yHy = fma(v.y[o + c], Hy, yHy);
                vg0 = fma(v.s[o + c], v.g[o + c], vg0);
                vg2 = fma(Hy, v.g[o + c], vg2);
            }
            pgl_blk_sum3(yHy, vg0, vg2, red);
            const double c0 = (1.0 + rho * yHy) * rho;
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                const double tc = lazy ? hs * v.g[o + c] : v.t[o + c];
                const double s = v.s[o + c], Hy = tc - sc * v.Hg[o + c];
                const double u0 = c0 * s, u1 = -rho * Hy, u2 = -rho * s;
                v.Hg[o + c] = tc + (u0 * vg0 + u1 * vg0 + u2 * vg2);          // H_new g_new
                if (Wh) {                                    // implicit form: the update joins the history
                    const size_t hq = ((size_t)r * Kmax + (size_t)v.hk[r]) * 2 * P + c;
                    Wh[hq] = s;
                    Wh[hq + P] = Hy;
                } else {
                    U[3 * c] = u0; U[3 * c + 1] = u1; U[3 * c + 2] = u2;
                    V[3 * c] = s; V[3 * c + 1] = s; V[3 * c + 2] = Hy;
                }
            }
            if (Wh && tid == 0) {
                const size_t cq = ((size_t)r * Kmax + (size_t)v.hk[r]) * 2;
                cs[cq] = c0;
                cs[cq + 1] = rho;
            }
            ident = Wh ? false : lazy;                       // dense: still not materialised -- hs * I + U V^T at the next pass
            pend = Wh ? false : true;
            hist_add = Wh != nullptr;
        } else {
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) v.Hg[o + c] = lazy ? hs * v.g[o + c] : v.t[o + c];
            ident = lazy;
            pend = false;
        }
        __syncthreads();
    } else {                                                 // stalled line search: restart once, then freeze
        again = restarts == 0.0;
        if (again) restarts = 1.0;
        else frozen = true;
    }
    double sl = 0.0, gg = 0.0, gmax = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
        const double gc = v.g[o + c];
        sl = fma(-v.Hg[o + c], gc, sl);
        gg = fma(gc, gc, gg);
        gmax = fmax(gmax, fabs(gc));
    }
    pgl_blk_sum3(sl, gg, gmax, red, true);
    const bool newls = a || again;
    const bool reset = newls && (again || !(sl < 0.0));      // restart / not a descent direction: H = I
    if (newls)
        for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
            const double gc = v.g[o + c];
            if (reset) v.Hg[o + c] = gc;
            v.p[o + c] = reset ? -gc : -v.Hg[o + c];
        }
    if (tid == 0) {
        v.iters[r] = iters;
        v.restarts[r] = restarts;
        if (frozen) v.frozen[r] = 1.0;
        if (reset) { ident = true; pend = false; hs = 1.0; }
        v.hk[r] = reset ? 0.0 : v.hk[r] + (hist_add ? 1.0 : 0.0);
        v.ident[r] = ident ? 1.0 : 0.0;
        v.pend[r] = pend ? 1.0 : 0.0;
        v.hscale[r] = hs;
        const bool go = !frozen && gmax > gtol && iters < (double)maxiter;
        if (newls && go) {
            const double slope = reset ? -gg : sl;
            const double f = v.f[r];
            const double fprev = again ? f + sqrt(gg) / 2.0 : v.fprev[r];
            v.fprev[r] = fprev;
            v.slope[r] = slope;
            PglLs s;
            const double a0 = pgl_ls_first_step(f, fprev, slope);
            pgl_ls_start(&s, a0, f, slope, PGL_LS_FTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
            pgl_ls_store(v, r, &s);
            v.alpha[r] = a0;
            v.fb[r] = f;
        }
        v.active[r] = go ? 1.0 : 0.0;
        v.acc[r] = 0.0;                                      // cleared for the next launch
        v.upd[r] = 0.0;
        v.stall[r] = 0.0;
    }
}

// t = H g of ONE row from its update history inside a block of NT threads (NT / 64 waves): the two kernels above in one
// workgroup -- wave w takes the dots of the updates j = w, w + NT / 64, ... exactly as a wave of k_bfgs_hdots does, then
// NT / 512 groups of eight waves each combine 64 components exactly as a block of k_bfgs_hcomb does: the same numbers.
// The merged iteration kernel uses it while the history of a row is short (4 hk P numbers through one workgroup).
template <int NT>
__device__ __forceinline__ void pgl_bfgs_hist_row(const BfgsView& v, const int r, const double* __restrict__ Wh,
                                                  const double* __restrict__ cs, const int Kmax, double* __restrict__ ab,
                                                  double (*part)[64], const int tid)
{
    if (v.acc[r] == 0.0) return;
    const int P = v.P, K = (int)v.hk[r];
    if (K == 0) return;                                      // (no history: the update uses t = hscale g)
    const int lane = tid & 63, wv = tid >> 6;
    const double* __restrict__ g = v.g + (size_t)r * P;
    for (int j = wv; j < K; j += NT / 64) {
        const double* __restrict__ W = Wh + ((size_t)r * Kmax + j) * 2 * P;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int i = lane;
        for (; i + 64 < P; i += 128) {
            const double g0 = g[i], g1 = g[i + 64];
            const double s0 = W[i], s1 = W[i + 64], h0 = W[P + i], h1 = W[P + i + 64];
            a0 = fma(s0, g0, a0);
            a1 = fma(s1, g1, a1);
            b0 = fma(h0, g0, b0);
            b1 = fma(h1, g1, b1);
        }
        if (i < P) {
            a0 = fma(W[i], g[i], a0);
            b0 = fma(W[P + i], g[i], b0);
        }
        double a = a0 + a1, b = b0 + b1;
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane == 0) {
            const size_t q = ((size_t)r * Kmax + j) * 2;
            const double c0 = cs[q], rho = cs[q + 1];
            ab[q] = c0 * a - rho * b;
            ab[q + 1] = -rho * a;
        }
    }
    __syncthreads();                                         // the coefficients are in memory for the whole workgroup
    constexpr int NVB = (NT >= 512) ? NT / 512 : 1;          // groups of eight waves
    constexpr int WPG = (NT >= 512) ? 8 : NT / 64;           // (a 256-thread block: one group of four -- not used)
    static_assert(NT >= 512 && NT % 512 == 0, "pgl_bfgs_hist_row: whole groups of eight waves");
    const int vb = wv / WPG, w = wv % WPG;
    const int nblk = (P + 63) / 64;
    const double* __restrict__ c = ab + (size_t)r * Kmax * 2;
    for (int b0 = 0; b0 < nblk; b0 += NVB) {
        const int blk = b0 + vb;
        const int i = blk * 64 + lane;
        const bool in = blk < nblk && i < P;
        const double* __restrict__ W = Wh + (size_t)r * Kmax * 2 * P + (in ? i : 0);
        double acc0 = 0.0, acc1 = 0.0;
        int j = w;
        for (; j + 8 < K; j += 16) {
            const double* u0 = W + (size_t)j * 2 * P;
            const double* u1 = W + (size_t)(j + 8) * 2 * P;
            const double s0 = u0[0], h0 = u0[P], s1 = u1[0], h1 = u1[P];
            acc0 = fma(s0, c[2 * j], acc0);
            acc0 = fma(h0, c[2 * j + 1], acc0);
            acc1 = fma(s1, c[2 * j + 16], acc1);
            acc1 = fma(h1, c[2 * j + 17], acc1);
        }
        if (j < K) {
            const double* u0 = W + (size_t)j * 2 * P;
            acc0 = fma(u0[0], c[2 * j], acc0);
            acc0 = fma(u0[P], c[2 * j + 1], acc0);
        }
        part[wv][lane] = acc0 + acc1;
        __syncthreads();
        if (w == 0 && in) {
            double t = v.hscale[r] * v.g[(size_t)r * P + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) t += part[vb * 8 + q][lane];
            v.t[(size_t)r * P + i] = t;
        }
        __syncthreads();
    }
}

// One iteration of the lock-step optimiser for the listed rows, one workgroup per row: the phases an evaluation is followed
// by, selected by `phases` so that the same code serves the one-launch form and the split form around the multi-workgroup
// inverse-Hessian kernels (k_bfgs_hdots + k_bfgs_hcomb for long histories, k_bfgs_hmul for dense matrices):
//   PGL_STEP_LS     priors + NaN rules on the evaluation's (ll, grad) (have_prior; else f, g arrive final), line-search step
//   PGL_STEP_HIST   t = H g from the update history of a row that has just taken a step (NT >= 512)
//   PGL_STEP_UPDATE BFGS update / restart / freeze / convergence, next direction and first step; then the row's next trial
//                   point into Xt_next[pos_next[r]] (the list of the NEXT launch: pos_next null = same positions) and its
//                   active flag into flags_out[r] (host-visible memory: the driver reads it without a copy kernel)
// The rows are the evaluation's list (rows[j], null: j); Xt / ft / gt are indexed by list position.
#define PGL_STEP_LS 1
#define PGL_STEP_HIST 2
#define PGL_STEP_UPDATE 4
struct BfgsStepArgs {
    const int* rows;
    const double* Xt;
    double* ft;
    double* gt;
    BfgsPrior q;
    int have_prior, max_trials, maxiter, init_scaling, Kmax, phases;
    double gtol;
    double* Wh;
    double* cs;
    double* ab;
    const int* pos_next;
    double* Xt_next;
    double* flags_out;
};
template <int NT>
__global__ __launch_bounds__(NT) void k_bfgs_step(const BfgsView v, const BfgsStepArgs a)
{
    __shared__ double red[12];
    __shared__ int dec[3];
    __shared__ double part[(NT >= 512) ? NT / 64 : 1][64];
    const int j = blockIdx.x, r = a.rows ? a.rows[j] : j, tid = threadIdx.x, P = v.P;
    if (a.phases & PGL_STEP_LS) {
        if (a.have_prior && v.active[r] != 0.0)
            pgl_bfgs_objective_row(P, a.Xt + (size_t)j * P, a.gt + (size_t)j * P, a.ft + j, a.q, red, tid);
        __syncthreads();                                     // f, g of the trial are in memory for the whole workgroup
        pgl_bfgs_linesearch_row(v, r, a.Xt + (size_t)j * P, a.ft + j, a.gt + (size_t)j * P, a.max_trials, red, dec, tid);
        __syncthreads();
    }
    if constexpr (NT >= 512) {
        if ((a.phases & PGL_STEP_HIST) && a.Wh) {
            pgl_bfgs_hist_row<NT>(v, r, a.Wh, a.cs, a.Kmax, a.ab, part, tid);
            __syncthreads();
        }
    }
    if (a.phases & PGL_STEP_UPDATE) {
        pgl_bfgs_update_row(v, r, a.gtol, a.maxiter, a.init_scaling, a.Wh, a.cs, a.Kmax, red, tid);
        __syncthreads();
        if (a.Xt_next) {
            const int jn = a.pos_next ? a.pos_next[r] : j;
            if (jn >= 0) {
                const double al = v.alpha[r];
                for (int c = pgl_row_c0(tid, P); c < P; c += 256)
                    a.Xt_next[(size_t)jn * P + c] = fma(al, v.p[(size_t)r * P + c], v.X[(size_t)r * P + c]);
            }
        }
        if (a.flags_out && tid == 0)
            __hip_atomic_store(a.flags_out + r, v.active[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
