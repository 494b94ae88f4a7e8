/*
 * C restatement of the reference's population-GLM ll + gradient dataflow
 * ("B1") -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 * Nothing under theano_pyglm_amd/ links or calls this file.
 *
 * It mirrors the op sequence the reference's Theano graph executes per neuron
 * (SURVEY.md §2 K2-K7), on features materialised once like data['fS']
 * (pyglm/components/impulse.py:114-130):
 *   I_imp = sum_b fS*w            impulse.py:58
 *   I_net = I_imp . W_eff         glm.py:33-39
 *   lam   = nlin(bias+I_stim+I_net)   glm.py:43-45, nlin.py:25/43
 *   ll    = sum(-dt*lam + log(lam)*S[:,n])   glm.py:52
 *   grad  = T.grad(ll, [bias, w_stim, w_ir]) coord_descent.py:27-30
 * Parity: checked against oracle/glm_oracle.py (tests/test_oracle.py), which is
 * itself pinned as described in that file's header.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
static double sigmoid(double x)
{
    double e = exp(-fabs(x));
    return x >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
}

/* basis.py:201-236: fS[t,n,b] = sum_{tau=1..R} S[t-tau,n] ibasis[tau-1,b] (spike-driven form) */
void oracle_features(const uint8_t* S, int64_t nT, int N, const double* ibasis, int R, int B,
                     double* fS)
{
    memset(fS, 0, sizeof(double) * (size_t)nT * N * B);
    for (int64_t s = 0; s < nT; ++s)
        for (int n = 0; n < N; ++n) {
            const double c = S[s * N + n];
            if (c == 0.0) continue;
            const int64_t e = (s + 1 + R < nT) ? s + 1 + R : nT;
            for (int64_t t = s + 1; t < e; ++t) {
                const double* ph = ibasis + (size_t)(t - s - 1) * B;
                double* o = fS + ((size_t)t * N + n) * B;
                for (int b = 0; b < B; ++b) o[b] += c * ph[b];
            }
        }
}

/* One neuron, reference dataflow with materialised intermediates.
 * theta = [bias, w_stim(Dstim), w(N*B)], grad same layout (may be NULL).
 * scratch: 3*nT doubles.  Returns ll. */
double oracle_ll_grad_neuron(int n, const uint8_t* S, int64_t nT, int N, int B, const double* fS,
                             const double* fstim, int Dstim, const double* theta,
                             const double* Weff_col, int nlin, double dt, double* grad,
                             double* scratch)
{
    const int K = N * B;
    const double bias = theta[0];
    const double* ws = theta + 1;
    const double* w = theta + 1 + Dstim;
    double* x = scratch;
    double* r = scratch + nT;
    double ll = 0.0;
    /* forward: I_imp (per presyn) then dot with W_eff, like the Theano elemwise+sum, gemv */
    for (int64_t t = 0; t < nT; ++t) {
        const double* f = fS + (size_t)t * K;
        double inet = 0.0;
        for (int np = 0; np < N; ++np) {
            double iimp = 0.0;
            for (int b = 0; b < B; ++b) iimp += f[np * B + b] * w[np * B + b];
            inet += iimp * Weff_col[np];
        }
        double istim = 0.0;
        for (int j = 0; j < Dstim; ++j) istim += fstim[(size_t)t * Dstim + j] * ws[j];
        x[t] = bias + istim + inet;
    }
    for (int64_t t = 0; t < nT; ++t) {
        const double s = S[t * N + n];
        if (nlin == 1) {
            const double lam = softplus(x[t]);
            ll += -dt * lam + log(lam) * s;
            r[t] = (-dt + s / lam) * sigmoid(x[t]);
        } else {
            const double lam = exp(x[t]);
            ll += -dt * lam + x[t] * s;
            r[t] = -dt * lam + s;
        }
    }
    if (grad) {
        memset(grad, 0, sizeof(double) * (size_t)(1 + Dstim + K));
        double gb = 0.0;
        for (int64_t t = 0; t < nT; ++t) {
            const double rt = r[t];
            gb += rt;
            const double* f = fS + (size_t)t * K;
            double* g = grad + 1 + Dstim;
            for (int k = 0; k < K; ++k) g[k] += rt * f[k];
            for (int j = 0; j < Dstim; ++j) grad[1 + j] += rt * fstim[(size_t)t * Dstim + j];
        }
        grad[0] = gb;
        for (int np = 0; np < N; ++np)
            for (int b = 0; b < B; ++b) grad[1 + Dstim + np * B + b] *= Weff_col[np];
    }
    return ll;
}

/* Population: loop n over [n_lo,n_hi) (population.py:80-86).  threads>1 uses OpenMP over neurons. */
void oracle_ll_grad(int n_lo, int n_hi, const uint8_t* S, int64_t nT, int N, int B,
                    const double* fS, const double* fstim, int Dstim, const double* theta,
                    const double* Weff, int nlin, double dt, double* ll_out, double* grad_out,
                    int threads)
{
    const int P = 1 + Dstim + N * B;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int n = n_lo; n < n_hi; ++n) {
        double* scratch = (double*)malloc(sizeof(double) * 2 * (size_t)nT);
        double* wcol = (double*)malloc(sizeof(double) * N);
        for (int np = 0; np < N; ++np) wcol[np] = Weff[(size_t)np * N + n];
        ll_out[n - n_lo] = oracle_ll_grad_neuron(
            n, S, nT, N, B, fS, fstim, Dstim, theta + (size_t)(n - n_lo) * P, wcol, nlin, dt,
            grad_out ? grad_out + (size_t)(n - n_lo) * P : NULL, scratch);
        free(scratch);
        free(wcol);
    }
}
