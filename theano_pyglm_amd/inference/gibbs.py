"""
Collapsed Gibbs update of one column (A[:,n], W[:,n]) of the network -- counterpart of
pyglm/inference/gibbs.py:775-1250 (CollapsedGibbsNetworkColumnUpdate), i.e. the
"synth_mcmc inner ll" path.

What runs on the GPU: the impulse currents I_imp of all presynaptic neurons for the
column's post-synaptic neuron (once per column, gibbs.py:812-833), the "other" current
(a rank-1 downdate of the resident total I_net instead of the reference's full gemv per
pair, gibbs.py:835-864) and the batched ll at the 10 Gauss-Hermite nodes + w=0
(gibbs.py:910-937, 1002-1032).  What stays on the host: the Gauss-Hermite marginal,
the Bernoulli draw of A and the draw of W.  The reference samples W with adaptive
rejection sampling from the un-vendored `hips` package; here W is drawn by the
inverse-CDF sampler the reference also carries (gibbs.py:1068-1084) on a refined grid.
"""
import numpy as np
from scipy.special import logsumexp

from theano_pyglm_amd.inference.log_sum_exp import log_sum_exp_sample


class CollapsedGibbsNetworkColumnUpdate(object):
    def __init__(self, rng=None):
        self.DEG_GAUSS_HERMITE = 10
        self.GAUSS_HERMITE_ABSCISSAE, self.GAUSS_HERMITE_WEIGHTS = \
            np.polynomial.hermite.hermgauss(self.DEG_GAUSS_HERMITE)     # gibbs.py:787-789
        self.rng = np.random if rng is None else rng
        self.n_grid = 64

    def preprocess(self, population):
        """gibbs.py:791-810."""
        self.population = population
        self.network = population.network
        self.glm = population.glm
        w = self.network.weights
        self.mu_w = w.prior.mu
        self.sigma_w = w.prior.sigma
        if getattr(w, 'refractory_prior', None) is not None:
            self.mu_w_ref = w.refractory_prior.mu
            self.sigma_w_ref = w.refractory_prior.sigma
        else:
            self.mu_w_ref, self.sigma_w_ref = self.mu_w, self.sigma_w

    # -- reference-shaped helpers (host arrays in / out) -----------------------------
    def _precompute_vars(self, x, n_post):
        """gibbs.py:812-833: I_bias, I_stim, I_imp (nT,N), p_A."""
        pop = self.population
        xn = x['glms'][n_post]
        h = pop._handle(pop._current)
        I_bias = self.glm.bias_model.I_bias(xn['bias'])
        w = self.glm.imp_model.flat_weights(xn['imp']).reshape(pop.N, -1)
        I_imp = h.impulse_currents(w)
        if self.glm.Dstim > 0:
            I_stim = pop.stim_features().dot(self.glm.bkgd_model.flat_weights(xn['bkgd']))
        else:
            I_stim = 0.0
        return I_bias, I_stim, I_imp, self.network.graph.pA

    def _precompute_other_current(self, x, I_imp, n_pre, n_post):
        """gibbs.py:835-864: I_net with A[n_pre,n_post] = 0."""
        col = self.network.W_eff(x['net'])[:, n_post].copy()
        col[n_pre] = 0.0
        return I_imp.dot(col)

    def _glm_ll(self, n_pre, n_post, w, x, I_bias, I_stim, I_imp, I_net_other):
        """gibbs.py:910-937 for one weight or an array of weights."""
        pop = self.population
        ws = np.atleast_1d(np.asarray(w, dtype=float))
        ll = np.zeros(len(ws))
        for data in pop.data_sequences:
            pop.set_data(data)
            h = pop._handle(data)
            stim = None if np.isscalar(I_stim) else I_stim
            bias = I_bias + (I_stim if np.isscalar(I_stim) else 0.0)
            ll += h.ll_from_current(n_post, bias, stim, I_net_other, I_imp[:, n_pre], ws)
        return ll if np.ndim(w) else float(ll[0])

    # -- the collapsed draw -----------------------------------------------------------
    def _marginal(self, log_L):
        """log G = logsumexp(log_L + log(omega_i / sqrt(pi)))  (gibbs.py:1015-1022)."""
        log_L = np.where(np.isnan(log_L), -np.inf, log_L)
        wl = log_L + np.log(self.GAUSS_HERMITE_WEIGHTS / np.sqrt(np.pi))
        wl = np.where(np.isnan(wl), -np.inf, wl)
        return logsumexp(wl)

    def _inverse_cdf_sample_w(self, mu_w, sigma_w, ws, log_L):
        """gibbs.py:1068-1084."""
        lp = -0.5 / sigma_w ** 2 * (ws - mu_w) ** 2 + log_L
        p = np.exp(lp - logsumexp(lp))
        F = np.concatenate(([0.0], np.cumsum(0.5 * (p[1:] + p[:-1]) * np.diff(ws))))
        F = F / F[-1]
        return float(np.interp(self.rng.random_sample(), F, ws))

    def update(self, x, n_post):
        """gibbs.py:1229-1250 with the device-resident inner loop: resample column n_post."""
        pop = self.population
        N = pop.N
        if len(pop.data_sequences) != 1:
            raise Exception("device-resident column update supports one data sequence")
        pop.set_data(pop.data_sequences[0])
        h = pop._handle(pop._current)
        A = np.asarray(x['net']['graph']['A'])
        W = np.asarray(x['net']['weights']['W'], dtype=float).reshape(N, N)
        xn = x['glms'][n_post]
        h.gibbs_prepare(n_post, self.glm.theta_row(xn), (A * W)[:, n_post])
        p_A = self.network.graph.pA
        stats = []
        for n_pre in self.rng.permutation(N):
            mu_w, sigma_w = (self.mu_w_ref, self.sigma_w_ref) if n_pre == n_post \
                else (self.mu_w, self.sigma_w)
            aw_cur = float(A[n_pre, n_post] * W[n_pre, n_post])
            W_nns = np.sqrt(2) * sigma_w * self.GAUSS_HERMITE_ABSCISSAE + mu_w      # gibbs.py:1004
            ll = h.gibbs_ll(n_pre, aw_cur, np.concatenate((W_nns, [0.0])))
            log_L, ll_noA = ll[:-1], ll[-1]
            log_G = self._marginal(log_L)
            if not np.isfinite(log_G):
                raise Exception("log_G not finie")
            with np.errstate(divide='ignore'):
                log_pr_A = np.log(p_A[n_pre, n_post]) + log_G
                log_pr_noA = np.log(1.0 - p_A[n_pre, n_post]) + ll_noA
            if np.isnan(log_pr_noA):
                log_pr_noA = -np.inf
            a_new = log_sum_exp_sample([log_pr_noA, log_pr_A], self.rng)           # gibbs.py:1041
            if a_new == 1:
                grid = mu_w + sigma_w * np.linspace(-4.0, 4.0, self.n_grid)
                w_new = self._inverse_cdf_sample_w(mu_w, sigma_w, grid,
                                                   h.gibbs_ll(n_pre, aw_cur, grid))
            else:
                w_new = mu_w + sigma_w * self.rng.standard_normal()                 # gibbs.py:1060-1062
            h.gibbs_update(n_pre, a_new * w_new - aw_cur)
            A[n_pre, n_post] = a_new
            W[n_pre, n_post] = w_new
            stats.append((int(n_pre), float(log_G), float(ll_noA)))
        x['net']['graph']['A'] = A
        x['net']['weights']['W'] = W.ravel()
        return stats
