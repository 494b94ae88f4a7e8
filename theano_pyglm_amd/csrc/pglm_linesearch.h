// Line search of the lock-step BFGS (inference/batched_bfgs.py) as a reverse-communication state machine: one call per
// evaluated trial step, no callbacks -- which is what lets all neurons of a shard advance through their own searches in
// lock step, one fused ll+grad launch per trial.
//
// The reference fits every neuron with scipy.optimize.minimize(method="bfgs") (coord_descent.py:194-199), whose line
// search is MINPACK-2's DCSRCH / DCSTEP: the safeguarded cubic / quadratic interpolation search for the strong Wolfe
// conditions of  J. J. More' and D. J. Thuente, "Line search algorithms with guaranteed sufficient decrease", ACM TOMS
// 20 (1994) 286-307, called with ftol = 1e-4, gtol = 0.9, xtol = 1e-14, step bounds [1e-100, 1e100] and the first trial
// step  min(1, 1.01 * 2 (f_k - f_{k-1}) / slope).  This file restates that published algorithm (sections 2-4 of the paper:
// the auxiliary function psi of stage 1, the four interpolation cases, the interval update and its bisection safeguard)
// in plain C so that the per-neuron iterates of the lock-step fit are scipy's up to rounding
// (tests/test_host_logic.py compiles it with gcc and compares the step sequence with scipy's own search).
//
// Plain C subset, usable from host and device code.
#ifndef PGLM_LINESEARCH_H
#define PGLM_LINESEARCH_H

#if defined(__HIPCC__)
#define PGL_LS_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define PGL_LS_FN static inline
#endif

#define PGL_LS_EVALUATE 0   // evaluate phi, phi' at s->stp and call pgl_ls_step again
#define PGL_LS_CONVERGED 1  // the step just evaluated satisfies the strong Wolfe conditions
#define PGL_LS_WARNING 2    // no further progress possible (rounding, xtol, step bound): s->stx is the best step seen

#define PGL_LS_NDOUBLES 18

typedef struct {
    double stp;                      // the trial step to evaluate next (EVALUATE) / the accepted step (CONVERGED)
    double finit, ginit, gtest;      // phi(0), phi'(0), ftol * phi'(0)
    double stx, fx, gx;              // best step so far, its value and derivative
    double sty, fy, gy;              // the other end point of the interval of uncertainty
    double stmin, stmax;             // bounds for the next trial step
    double width, width1;            // interval widths of the last two iterations (bisection safeguard)
    double brackt, stage;            // flags (stored as doubles: the optimiser state is one block of doubles)
    double nfev;                     // trial steps evaluated in this search
    double moved;                    // 1 after a call in which the evaluated step became the best step stx
} PglLs;

PGL_LS_FN double pgl_ls_abs(double x) { return x < 0.0 ? -x : x; }
PGL_LS_FN double pgl_ls_max3(double a, double b, double c)
{
    const double m = a > b ? a : b;
    return m > c ? m : c;
}
PGL_LS_FN double pgl_ls_sqrt(double x) { return __builtin_sqrt(x); }

// begin a search along a descent direction: phi(0) = f0, phi'(0) = g0 < 0, first trial step stp0 in [stpmin, stpmax]
PGL_LS_FN void pgl_ls_start(PglLs* s, double stp0, double f0, double g0, double ftol, double stpmin, double stpmax)
{
    s->stp = stp0;
    s->finit = f0; s->ginit = g0; s->gtest = ftol * g0;
    s->stx = 0.0; s->fx = f0; s->gx = g0;
    s->sty = 0.0; s->fy = f0; s->gy = g0;
    s->stmin = 0.0;
    s->stmax = stp0 + 4.0 * stp0;
    s->width = stpmax - stpmin;
    s->width1 = 2.0 * s->width;
    s->brackt = 0.0; s->stage = 1.0; s->nfev = 0.0; s->moved = 0.0;
}

// One safeguarded interpolation step (section 4 of the paper).  (stx, fx, dx): best step; (sty, fy, dy): other end point;
// (stp, fp, dp): the trial just evaluated.  Updates the interval and returns the next trial step.
PGL_LS_FN double pgl_ls_interpolate(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy,
                                    double stp, double fp, double dp, double* brackt, double stpmin, double stpmax)
{
    const double sgnd = dp * (*dx / pgl_ls_abs(*dx));
    double stpf;
    if (fp > *fx) {
        // case 1: higher value -- the minimum is bracketed; cubic through both points, quadratic through (fx, dx, fp);
        // take the cubic step if it is closer to stx, else the average of the two
        const double theta = 3.0 * (*fx - fp) / (stp - *stx) + *dx + dp;
        const double sc = pgl_ls_max3(pgl_ls_abs(theta), pgl_ls_abs(*dx), pgl_ls_abs(dp));
        double gamma = sc * pgl_ls_sqrt((theta / sc) * (theta / sc) - (*dx / sc) * (dp / sc));
        if (stp < *stx) gamma = -gamma;
        const double p = (gamma - *dx) + theta, q = ((gamma - *dx) + gamma) + dp, r = p / q;
        const double stpc = *stx + r * (stp - *stx);
        const double stpq = *stx + ((*dx / ((*fx - fp) / (stp - *stx) + *dx)) / 2.0) * (stp - *stx);
        stpf = (pgl_ls_abs(stpc - *stx) < pgl_ls_abs(stpq - *stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
        *brackt = 1.0;
    } else if (sgnd < 0.0) {
        // case 2: lower value, derivatives of opposite sign -- bracketed; cubic and secant step, the one farther from stp
        const double theta = 3.0 * (*fx - fp) / (stp - *stx) + *dx + dp;
        const double sc = pgl_ls_max3(pgl_ls_abs(theta), pgl_ls_abs(*dx), pgl_ls_abs(dp));
        double gamma = sc * pgl_ls_sqrt((theta / sc) * (theta / sc) - (*dx / sc) * (dp / sc));
        if (stp > *stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + *dx, r = p / q;
        const double stpc = stp + r * (*stx - stp);
        const double stpq = stp + (dp / (dp - *dx)) * (*stx - stp);
        stpf = (pgl_ls_abs(stpc - stp) > pgl_ls_abs(stpq - stp)) ? stpc : stpq;
        *brackt = 1.0;
    } else if (pgl_ls_abs(dp) < pgl_ls_abs(*dx)) {
        // case 3: lower value, same sign, derivative magnitude decreases -- the cubic step only if the cubic tends to
        // infinity in the direction of the step (or its minimum lies beyond stp); secant step otherwise
        const double theta = 3.0 * (*fx - fp) / (stp - *stx) + *dx + dp;
        const double sc = pgl_ls_max3(pgl_ls_abs(theta), pgl_ls_abs(*dx), pgl_ls_abs(dp));
        const double d2 = (theta / sc) * (theta / sc) - (*dx / sc) * (dp / sc);
        double gamma = sc * pgl_ls_sqrt(d2 > 0.0 ? d2 : 0.0);
        if (stp > *stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = (gamma + (*dx - dp)) + gamma, r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (*stx - stp);
        else if (stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - *dx)) * (*stx - stp);
        if (*brackt != 0.0) {
            stpf = (pgl_ls_abs(stpc - stp) < pgl_ls_abs(stpq - stp)) ? stpc : stpq;
            const double lim = stp + 0.66 * (*sty - stp);
            if (stp > *stx) stpf = stpf < lim ? stpf : lim;
            else stpf = stpf > lim ? stpf : lim;
        } else {
            stpf = (pgl_ls_abs(stpc - stp) > pgl_ls_abs(stpq - stp)) ? stpc : stpq;
            stpf = stpf < stpmax ? stpf : stpmax;
            stpf = stpf > stpmin ? stpf : stpmin;
        }
    } else {
        // case 4: lower value, same sign, derivative magnitude does not decrease -- cubic through stp and sty when
        // bracketed, else the bound
        if (*brackt != 0.0) {
            const double theta = 3.0 * (fp - *fy) / (*sty - stp) + *dy + dp;
            const double sc = pgl_ls_max3(pgl_ls_abs(theta), pgl_ls_abs(*dy), pgl_ls_abs(dp));
            double gamma = sc * pgl_ls_sqrt((theta / sc) * (theta / sc) - (*dy / sc) * (dp / sc));
            if (stp > *sty) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + *dy, r = p / q;
            stpf = stp + r * (*sty - stp);
        } else if (stp > *stx) stpf = stpmax;
        else stpf = stpmin;
    }
    // the interval that contains a minimiser (value selects, no stores through selected pointers: the state stays in
    // registers on the device)
    const int hi = fp > *fx, sw = !hi && sgnd < 0.0;
    const double nsty = hi ? stp : (sw ? *stx : *sty), nfy = hi ? fp : (sw ? *fx : *fy), ndy = hi ? dp : (sw ? *dx : *dy);
    const double nstx = hi ? *stx : stp, nfx = hi ? *fx : fp, ndx = hi ? *dx : dp;
    *sty = nsty; *fy = nfy; *dy = ndy;
    *stx = nstx; *fx = nfx; *dx = ndx;
    return stpf;
}

// phi(s->stp) = f, phi'(s->stp) = g have been evaluated: test, update the interval, choose the next trial step.
PGL_LS_FN int pgl_ls_step(PglLs* s, double f, double g, double ftol, double gtol, double xtol, double stpmin,
                          double stpmax)
{
    const double stp = s->stp;
    const double ftest = s->finit + stp * s->gtest;
    const int brackt = s->brackt != 0.0;
    s->nfev += 1.0;
    s->moved = 0.0;
    if (s->stage == 1.0 && f <= ftest && g >= 0.0) s->stage = 2.0;
    // warnings first (in DCSRCH's order; the last one that applies is the one reported), then convergence
    int warn = 0;
    if (brackt && (stp <= s->stmin || stp >= s->stmax)) warn = 1;
    if (brackt && s->stmax - s->stmin <= xtol * s->stmax) warn = 1;
    if (stp == stpmax && f <= ftest && g <= s->gtest) warn = 1;
    if (stp == stpmin && (f > ftest || g >= s->gtest)) warn = 1;
    if (f <= ftest && pgl_ls_abs(g) <= gtol * (-s->ginit)) return PGL_LS_CONVERGED;
    if (warn) return PGL_LS_WARNING;
    const double stx_before = s->stx;
    double stpn, br = s->brackt;
    if (s->stage == 1.0 && f <= s->fx && f > ftest) {
        // stage 1: interpolate the auxiliary function psi(a) = phi(a) - phi(0) - ftol phi'(0) a
        double fxm = s->fx - s->stx * s->gtest, fym = s->fy - s->sty * s->gtest;
        double gxm = s->gx - s->gtest, gym = s->gy - s->gtest;
        stpn = pgl_ls_interpolate(&s->stx, &fxm, &gxm, &s->sty, &fym, &gym, stp, f - stp * s->gtest, g - s->gtest, &br,
                                  s->stmin, s->stmax);
        s->fx = fxm + s->stx * s->gtest; s->fy = fym + s->sty * s->gtest;
        s->gx = gxm + s->gtest; s->gy = gym + s->gtest;
    } else {
        stpn = pgl_ls_interpolate(&s->stx, &s->fx, &s->gx, &s->sty, &s->fy, &s->gy, stp, f, g, &br, s->stmin, s->stmax);
    }
    s->brackt = br;
    s->moved = (s->stx == stp && stx_before != stp) ? 1.0 : 0.0;
    if (br != 0.0) {
        // the interval must shrink by a third over two iterations, else bisect
        if (pgl_ls_abs(s->sty - s->stx) >= 0.66 * s->width1) stpn = s->stx + 0.5 * (s->sty - s->stx);
        s->width1 = s->width;
        s->width = pgl_ls_abs(s->sty - s->stx);
        s->stmin = s->stx < s->sty ? s->stx : s->sty;
        s->stmax = s->stx > s->sty ? s->stx : s->sty;
    } else {
        s->stmin = stpn + 1.1 * (stpn - s->stx);
        s->stmax = stpn + 4.0 * (stpn - s->stx);
    }
    stpn = stpn > stpmin ? stpn : stpmin;
    stpn = stpn < stpmax ? stpn : stpmax;
    // no further progress possible: fall back on the best step (the next call then reports the warning)
    if ((br != 0.0 && (stpn <= s->stmin || stpn >= s->stmax)) ||
        (br != 0.0 && s->stmax - s->stmin <= xtol * s->stmax)) stpn = s->stx;
    s->stp = stpn;
    if (!(stpn == stpn) || stpn - stpn != 0.0) return PGL_LS_WARNING;       // NaN / inf step (scipy: "WARN")
    return PGL_LS_EVALUATE;
}

// the first trial step of a search: scipy's  min(1, 1.01 * 2 (f - f_prev) / slope), 1 when that is not positive
PGL_LS_FN double pgl_ls_first_step(double f, double f_prev, double slope)
{
    double a = 1.0;
    if (slope != 0.0) {
        a = 1.01 * 2.0 * (f - f_prev) / slope;
        if (a > 1.0) a = 1.0;
        if (!(a > 0.0)) a = 1.0;           // zero, negative or NaN
    }
    return a;
}

#endif
