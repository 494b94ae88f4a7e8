"""
MCMC over the population GLM -- counterpart of pyglm/inference/gibbs.py for the models on the hot
path: the HMC updates of bias / stimulus / impulse parameters (gibbs.py:164-773), the collapsed
Gibbs update of one column (A[:,n], W[:,n]) of the network (gibbs.py:775-1250, the "synth_mcmc
inner ll" path), `initialize_updates` and `gibbs_sample` (gibbs.py:2413-2560).

HMC blocks.  Given the network, neuron n's bias / stimulus / impulse parameters enter only ll_n,
so the reference's per-neuron updates are conditionally independent across neurons.  Each block
update therefore advances all N chains in lock step: one leapfrog step of every neuron is ONE
batched population ll+grad evaluation on the device (the fused kernel) instead of N Theano calls.
`update(x, n)` keeps the reference's per-neuron form.  HMC and ARS themselves are in-repo
(inference/hmc.py, inference/ars.py) because the reference's come from the un-vendored `hips`.

Collapsed network column update:

What runs on the GPU: the impulse currents I_imp of all presynaptic neurons for the
column's post-synaptic neuron (once per column, gibbs.py:812-833), the "other" current
(a rank-1 downdate of the resident total I_net instead of the reference's full gemv per
pair, gibbs.py:835-864) and the batched ll at the 10 Gauss-Hermite nodes + w=0
(gibbs.py:910-937, 1002-1032).  What stays on the host: the Gauss-Hermite marginal,
the Bernoulli draw of A and the draw of W: adaptive rejection sampling started from the
quadrature nodes (gibbs.py:1087-1126; each extra abscissa is one more device inner-ll call), or the
inverse-CDF sampler the reference also carries (gibbs.py:1068-1084) on a refined grid.
"""
import copy
import time

import numpy as np
from scipy.special import logsumexp

from theano_pyglm_amd.components.impulse import DirichletImpulses
from theano_pyglm_amd.inference.ars import adaptive_rejection_sample
from theano_pyglm_amd.inference.hmc import hmc_lockstep, adapt_step_size
from theano_pyglm_amd.inference.log_sum_exp import log_sum_exp_sample
from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars


def _columns(syms, prefix=()):
    """{path: (lo, hi)} of every leaf of `syms` in the packed vector (sorted-key DFS order)."""
    out, cur = {}, [0]

    def walk(d, path):
        for k in sorted(d):
            v = d[k]
            if isinstance(v, dict):
                walk(v, path + (k,))
            else:
                size = int(np.prod(v.shape))
                out[path + (k,)] = (cur[0], cur[0] + size)
                cur[0] += size
    walk(syms, prefix)
    return out


class _HmcBlockUpdate(object):
    """HMC on one component's differentiable variables of every neuron (lock step)."""
    key = None
    n_steps = 2

    def __init__(self, rng=None):
        self.avg_accept_rate = 0.9
        self.step_sz = 0.1
        self.rng = np.random if rng is None else rng
        self.n_evals = 0

    def preprocess(self, population):
        self.population = population
        self.glm = population.glm
        self.syms = population.glm_syms()
        self.block_syms = self.syms.get(self.key, {})
        cols = _columns(self.syms)
        mine = [v for p, v in cols.items() if p[0] == self.key]
        self.lo = min(v[0] for v in mine) if mine else 0
        self.hi = max(v[1] for v in mine) if mine else 0
        self.block_cols = dict((p[1:], v) for p, v in cols.items() if p[0] == self.key)

    def _pack(self, xn):
        return packdict(get_vars(self.block_syms, xn[self.key]))

    def _adapt(self, accepted):
        for a in accepted:                              # the reference's recursion, neuron by neuron
            self.step_sz, self.avg_accept_rate = adapt_step_size(self.step_sz, self.avg_accept_rate, a)

    def _neg_lp_grad(self, x, n_lo, n_hi):
        lp, G = self.population.compute_lp_grad_packed(x, n_lo, n_hi)
        self.n_evals += 1
        U = np.where(np.isfinite(lp), -lp, np.inf)
        return U, -np.nan_to_num(G, nan=0.0, posinf=0.0, neginf=0.0)

    def update_range(self, x, n_lo, n_hi):
        if self.hi == self.lo:
            return x                                    # nothing to sample (e.g. NoStimulus)
        glms = x['glms']
        Q0, shapes = [], None
        for n in range(n_lo, n_hi):
            q, shapes = self._pack(glms[n])
            Q0.append(q)
        Q0 = np.array(Q0)

        def UG(Q):
            for i, n in enumerate(range(n_lo, n_hi)):
                set_vars(self.block_syms, glms[n][self.key], unpackdict(Q[i].copy(), shapes))
            U, G = self._neg_lp_grad(x, n_lo, n_hi)
            return U, G[:, self.lo:self.hi]

        Q1, acc, _ = hmc_lockstep(UG, self.step_sz, self.n_steps, Q0, rng=self.rng)
        for i, n in enumerate(range(n_lo, n_hi)):
            set_vars(self.block_syms, glms[n][self.key], unpackdict(Q1[i].copy(), shapes))
        self._adapt(acc)
        return x

    def update(self, x, n):
        """the reference's per-neuron form (gibbs.py:284-322 etc.)."""
        return self.update_range(x, n, n + 1)

    def update_all(self, x):
        return self.update_range(x, 0, self.population.N)


def _single_handle(population):
    """(handle, data) when the population holds exactly one data sequence, else (None, None)."""
    if len(population.data_sequences) != 1:
        return None, None
    data = population.data_sequences[0]
    population.set_data(data)
    return population._handle(data), data


class HmcBiasUpdate(_HmcBlockUpdate):
    """gibbs.py:164-322: 10 leapfrog steps on the bias.

    Fast path (one data sequence): the flat feature-weight matrix is built once per update and only
    its bias column changes between leapfrog steps; the prior -0.5/sigma^2 (b-mu)^2 (bias.py:33) and
    its derivative are two numpy expressions.  Same energies as the generic path, without packing
    128 state dicts per evaluation."""
    key = 'bias'
    n_steps = 10

    def update_range(self, x, n_lo, n_hi):
        pop = self.population
        h, _ = _single_handle(pop)
        if h is None:
            return _HmcBlockUpdate.update_range(self, x, n_lo, n_hi)
        glms = x['glms']
        theta = pop.theta_matrix(x, n_lo, n_hi)
        Weff = pop.W_eff(x)
        mu, sg = float(self.glm.bias_model.mu_bias), float(self.glm.bias_model.sig_bias)

        def UG(Q):
            theta[:, 0] = Q[:, 0]
            ll, g = h.ll_grad(theta, Weff, n_lo, n_hi)
            self.n_evals += 1
            lp = ll - 0.5 / sg ** 2 * (Q[:, 0] - mu) ** 2
            dlp = g[:, 0] - (Q[:, 0] - mu) / sg ** 2
            U = np.where(np.isfinite(lp), -lp, np.inf)
            return U, -np.nan_to_num(dlp, nan=0.0, posinf=0.0, neginf=0.0)[:, None]

        Q0 = theta[:, :1].copy()
        Q1, acc, _ = hmc_lockstep(UG, self.step_sz, self.n_steps, Q0, rng=self.rng)
        for i, n in enumerate(range(n_lo, n_hi)):
            glms[n]['bias']['bias'] = Q1[i].copy()
        self._adapt(acc)
        return x


class HmcBkgdUpdate(_HmcBlockUpdate):
    """gibbs.py:324-447: 2 leapfrog steps on the stimulus weights; no-op without a stimulus."""
    key = 'bkgd'
    n_steps = 2


class HmcImpulseUpdate(_HmcBlockUpdate):
    """gibbs.py:449-567: 2 leapfrog steps on all basis weights w_ir of the neuron."""
    key = 'imp'
    n_steps = 2


class HmcDirichletImpulseUpdate(_HmcBlockUpdate):
    """gibbs.py:569-773: for every existing edge n_pre -> n_post, 2 leapfrog steps on g_{n_pre}
    (one presynaptic neuron at a time, conditioning on the others); without an edge g is redrawn
    from its Gamma(alpha, 1) prior.  Lock step: round r handles the r-th incoming edge of every
    post-synaptic neuron at once; neurons with fewer edges sit the round out."""
    key = 'imp'
    n_steps = 2

    def update_range(self, x, n_lo, n_hi):
        N, imp = self.population.N, self.glm.imp_model
        A = np.asarray(x['net']['graph']['A']).reshape(N, N)
        glms = x['glms']
        posts = list(range(n_lo, n_hi))
        edges = [np.nonzero(A[:, n])[0] for n in posts]
        for n, e in zip(posts, edges):                              # gibbs.py:765-769
            for n_pre in np.setdiff1d(np.arange(N), e):
                glms[n]['imp']['g_%d' % n_pre] = self.rng.gamma(imp.alpha, np.ones(imp.B))
        h, _ = _single_handle(self.population)
        fast = h is not None
        if fast:
            theta = self.population.theta_matrix(x, n_lo, n_hi)      # after the prior redraws above
            Weff = self.population.W_eff(x)
            D = self.glm.Dstim
        for r in range(max([len(e) for e in edges] + [0])):
            active = np.array([r < len(e) for e in edges])
            pre = [int(e[r]) if r < len(e) else 0 for e in edges]
            names = ['g_%d' % k for k in pre]
            Q0 = np.array([np.asarray(glms[n]['imp'][nm], dtype=float) for n, nm in zip(posts, names)])
            if fast:
                # only the B flat weights beta = |g|/sum|g| of (n_pre -> n_post) change: update that
                # block of the resident theta matrix, chain the device gradient through the
                # normalisation (impulse.py:286-291) and add the Gamma prior (impulse.py:320-322)
                c0 = np.array([1 + D + k * imp.B for k in pre])
                rows = np.arange(len(posts))[:, None]
                colsb = c0[:, None] + np.arange(imp.B)[None, :]

                def UG(Q):
                    ga = np.abs(Q)
                    sm = ga.sum(axis=1, keepdims=True)
                    theta[rows, colsb] = np.where(active[:, None], ga / sm, theta[rows, colsb])
                    ll, g = h.ll_grad(theta, Weff, n_lo, n_hi)
                    self.n_evals += 1
                    gb = g[rows, colsb]
                    dg = np.sign(Q) * (gb * sm - np.sum(gb * ga, axis=1, keepdims=True)) / sm ** 2
                    with np.errstate(divide='ignore', invalid='ignore'):
                        lp = ll + np.sum((imp.alpha - 1.0) * np.log(ga) - ga, axis=1)
                        dlp = dg + (imp.alpha - 1.0) / Q - np.sign(Q)
                    U = np.where(np.isfinite(lp), -lp, np.inf)
                    return U, -np.nan_to_num(dlp, nan=0.0, posinf=0.0, neginf=0.0)
            else:
                cols = [self.block_cols[(nm,)] for nm in names]

                def UG(Q):
                    for i, n in enumerate(posts):
                        if active[i]:
                            glms[n]['imp'][names[i]] = Q[i].copy()
                    U, G = self._neg_lp_grad(x, n_lo, n_hi)
                    return U, np.array([G[i, c[0]:c[1]] for i, c in enumerate(cols)])

            Q1, acc, _ = hmc_lockstep(UG, self.step_sz, self.n_steps, Q0, active=active, rng=self.rng)
            for i, n in enumerate(posts):
                glms[n]['imp'][names[i]] = Q1[i].copy()
            if fast:                                  # leave theta at the accepted state
                ga = np.abs(Q1)
                theta[rows, colsb] = ga / ga.sum(axis=1, keepdims=True)
            self._adapt(acc[active])
        return x


class CollapsedGibbsNetworkColumnUpdate(object):
    def __init__(self, rng=None, w_sampler='ars'):
        self.w_sampler = w_sampler                      # 'ars' (gibbs.py:1054) | 'inverse_cdf' (:1053)
        self.n_ars_evals = 0
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
            I_stim = pop.stim_features().dot(self.glm.bkgd_model.dense_weights(xn['bkgd']))
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

    def _adaptive_rejection_sample_w(self, ll_of_w, mu_w, sigma_w, ws, log_L, ll_of_ws=None):
        """gibbs.py:1087-1126: ARS on log N(w; mu_w, sigma_w) + ll(w), started from the quadrature
        nodes with finite, moderate values; the density is shifted by its maximum over the nodes.
        ll_of_w(w) is one more device inner-ll evaluation of the pair.

        ll_of_ws (optional, vector of <= 14 weights -> vector of ll): with T ~ 10^5 bins the posterior
        of a weight is far narrower than the spacing of the quadrature nodes, and an ARS started from
        them spends ~9 evaluations = device launches per draw on rejections near the mode.  A launch of
        14 candidate weights costs what a launch of one does, so the hull is first refined around its
        maximum -- 14 abscissae between the neighbours of the best node, repeated while the density
        still drops by more than e^3 to a neighbour -- and the rejection loop starts from a tight hull.
        Any set of abscissae gives an exact sampler (the reference's own ARS lives in the un-vendored
        hips package: its abscissa sequence is not pinned)."""
        log_post = -0.5 / sigma_w ** 2 * (ws - mu_w) ** 2 + log_L
        Z = np.amax(log_post[np.isfinite(log_post)])
        valid = np.isfinite(log_post) & (log_post > -1e8) & (log_post < 1e8)
        xs, vs = np.asarray(ws, dtype=float)[valid], (log_post - Z)[valid]

        def f(w):
            self.n_ars_evals += 1
            ll = ll_of_w(w)
            v = -0.5 / sigma_w ** 2 * (w - mu_w) ** 2 + ll - Z
            return v if np.isfinite(v) else -np.inf

        if ll_of_ws is not None and len(xs) >= 2:
            order = np.argsort(xs)
            xs, vs = xs[order], vs[order]
            for _ in range(4):
                i = int(np.argmax(vs))
                lo = xs[i - 1] if i > 0 else xs[i] - (xs[i + 1] - xs[i])
                hi = xs[i + 1] if i + 1 < len(xs) else xs[i] + (xs[i] - xs[i - 1])
                v_lo = vs[i - 1] if i > 0 else -np.inf
                v_hi = vs[i + 1] if i + 1 < len(xs) else -np.inf
                if vs[i] - min(v_lo, v_hi) < 3.0 or not hi - lo > 1e-9 * max(1.0, abs(xs[i])):
                    break
                cand = np.linspace(lo, hi, 16)[1:-1]
                cand = cand[np.abs(cand - xs[i]) > 1e-12 * max(1.0, abs(xs[i]))]
                self.n_ars_evals += 1                                       # one launch
                vc = -0.5 / sigma_w ** 2 * (cand - mu_w) ** 2 + np.asarray(ll_of_ws(cand), dtype=float) - Z
                ok = np.isfinite(vc)
                xs = np.concatenate((xs, cand[ok]))
                vs = np.concatenate((vs, vc[ok]))
                order = np.argsort(xs)
                xs, vs = xs[order], vs[order]
        return float(adaptive_rejection_sample(f, xs, vs, (-np.inf, np.inf),
                                               stepsz=sigma_w / 2.0, rng=self.rng))

    def update_all(self, x, cols=None):
        """One pair (n_pre -> n_post) of EVERY column per device launch.  Given the rest of the state
        the columns are conditionally independent (the reference maps them over its engines and merges
        the columns afterwards, parallel_gibbs.py:24-37, 162-165); each column visits its presynaptic
        neurons in its own random order (gibbs.py:1238).  Step j of the sweep evaluates the 10
        Gauss-Hermite nodes + w = 0 (gibbs.py:1002-1032) of pair (perm_c[j], c) for all columns c in
        one launch (pgl_gibbs_ll_cols), draws A for all of them with numpy vector arithmetic, runs ARS
        only for the columns whose edge came up, and applies the rank-1 current updates in one launch.
        `cols`: the post-synaptic columns to resample (default all; a rank's shard in multi-GPU runs)."""
        pop = self.population
        N = pop.N
        h = _SequenceSum(pop)
        cols = np.arange(N) if cols is None else np.asarray(cols, dtype=int)
        nc = len(cols)
        A = np.array(x['net']['graph']['A']).reshape(N, N)
        W = np.array(x['net']['weights']['W'], dtype=float).reshape(N, N)
        h.gibbs_prepare_all(pop.theta_matrix(x), A * W)
        pA = np.asarray(self.network.graph.pA, dtype=float) * np.ones((N, N))
        with np.errstate(divide='ignore'):
            log_pA, log_pnA = np.log(pA), np.log(1.0 - pA)
        log_gh = np.log(self.GAUSS_HERMITE_WEIGHTS / np.sqrt(np.pi))
        absc = np.sqrt(2) * self.GAUSS_HERMITE_ABSCISSAE
        perms = np.array([self.rng.permutation(N) for _ in range(nc)])
        self.last_stats = []
        for j in range(N):
            n_pre = perms[:, j]
            ref = n_pre == cols
            mu = np.where(ref, self.mu_w_ref, self.mu_w)
            sg = np.where(ref, self.sigma_w_ref, self.sigma_w)
            W_nns = sg[:, None] * absc[None, :] + mu[:, None]                   # gibbs.py:1004
            probes = np.concatenate((W_nns, np.zeros((nc, 1))), axis=1)
            aw_cur = (A[n_pre, cols] * W[n_pre, cols]).astype(float)
            ll = h.gibbs_ll_cols(cols, n_pre, aw_cur, probes)
            log_L, ll_noA = ll[:, :-1], ll[:, -1]
            wl = log_L + log_gh[None, :]                                         # gibbs.py:1015-1022
            wl[np.isnan(wl)] = -np.inf
            mx = wl.max(axis=1)
            if not np.all(np.isfinite(mx)):
                raise Exception("log_G not finie")
            log_G = mx + np.log(np.exp(wl - mx[:, None]).sum(axis=1))
            log_pr_A = log_pA[n_pre, cols] + log_G
            log_pr_noA = log_pnA[n_pre, cols] + ll_noA
            log_pr_noA[np.isnan(log_pr_noA)] = -np.inf
            m2 = np.maximum(log_pr_noA, log_pr_A)
            if not np.all(np.isfinite(m2)):
                raise Exception("log_sum_exp_sample: no finite entry")
            p0 = np.exp(log_pr_noA - m2)                                         # log_sum_exp.py:4-37
            a_new = np.where(self.rng.random_sample(nc) * (p0 + np.exp(log_pr_A - m2)) < p0, 0, 1)
            w_new = mu + sg * self.rng.standard_normal(nc)                      # gibbs.py:1060-1062
            for i in np.nonzero(a_new)[0]:
                c1, p1, aw1 = cols[i:i + 1], n_pre[i:i + 1], aw_cur[i:i + 1]
                if self.w_sampler == 'ars':
                    w_new[i] = self._adaptive_rejection_sample_w(
                        lambda w: float(h.gibbs_ll_cols(c1, p1, aw1, np.array([[w]]))[0, 0]),
                        mu[i], sg[i], W_nns[i], log_L[i],
                        ll_of_ws=lambda wv: h.gibbs_ll_cols(c1, p1, aw1, np.asarray(wv)[None, :])[0])
                else:
                    grid = mu[i] + sg[i] * np.linspace(-4.0, 4.0, self.n_grid)
                    w_new[i] = self._inverse_cdf_sample_w(mu[i], sg[i], grid,
                                                          h.gibbs_ll_cols(c1, p1, aw1, grid[None, :])[0])
            delta = a_new * w_new - aw_cur
            nz = delta != 0.0
            if np.any(nz):
                h.gibbs_update_cols(cols[nz], n_pre[nz], delta[nz])
            A[n_pre, cols] = a_new
            W[n_pre, cols] = w_new
            self.last_stats.append((n_pre.copy(), log_G, ll_noA.copy()))
        x['net']['graph']['A'] = A.astype(np.asarray(x['net']['graph']['A']).dtype)
        x['net']['weights']['W'] = W.ravel()
        return x

    def update(self, x, n_post):
        """gibbs.py:1229-1250 with the device-resident inner loop: resample column n_post.
        The per-pair host work is kept to scalar arithmetic (16 384 pairs per sweep at N = 128)."""
        pop = self.population
        N = pop.N
        h = _SequenceSum(pop)
        A = np.asarray(x['net']['graph']['A'])
        W = np.asarray(x['net']['weights']['W'], dtype=float).reshape(N, N)
        xn = x['glms'][n_post]
        h.gibbs_prepare(n_post, self.glm.theta_row(xn), (A * W)[:, n_post])
        with np.errstate(divide='ignore'):
            log_pA = np.log(np.asarray(self.network.graph.pA, dtype=float)[:, n_post])
            log_pnA = np.log(1.0 - np.asarray(self.network.graph.pA, dtype=float)[:, n_post])
        log_gh = np.log(self.GAUSS_HERMITE_WEIGHTS / np.sqrt(np.pi))
        # quadrature nodes + the w = 0 probe, for off-diagonal and refractory (diagonal) priors
        nodes = {}
        for key, (m, sg) in (('off', (self.mu_w, self.sigma_w)), ('ref', (self.mu_w_ref, self.sigma_w_ref))):
            W_nns = np.sqrt(2) * sg * self.GAUSS_HERMITE_ABSCISSAE + m                  # gibbs.py:1004
            nodes[key] = (m, sg, W_nns, np.concatenate((W_nns, [0.0])))
        rnd = self.rng.random_sample
        stats = []
        for n_pre in self.rng.permutation(N):
            mu_w, sigma_w, W_nns, probes = nodes['ref' if n_pre == n_post else 'off']
            aw_cur = float(A[n_pre, n_post] * W[n_pre, n_post])
            ll = h.gibbs_ll(n_pre, aw_cur, probes)
            log_L, ll_noA = ll[:-1], float(ll[-1])
            wl = log_L + log_gh                                      # gibbs.py:1015-1022
            wl[np.isnan(wl)] = -np.inf
            mx = wl.max()
            if not np.isfinite(mx):
                raise Exception("log_G not finie")
            log_G = float(mx + np.log(np.exp(wl - mx).sum()))
            log_pr_A = log_pA[n_pre] + log_G
            log_pr_noA = log_pnA[n_pre] + ll_noA
            if log_pr_noA != log_pr_noA:                             # NaN (lam underflow at w = 0)
                log_pr_noA = -np.inf
            # log_sum_exp_sample over {no edge, edge} (gibbs.py:1041, log_sum_exp.py:4-37)
            m2 = max(log_pr_noA, log_pr_A)
            if not np.isfinite(m2):
                raise Exception("log_sum_exp_sample: no finite entry")
            p0 = np.exp(log_pr_noA - m2)
            a_new = 0 if rnd() * (p0 + np.exp(log_pr_A - m2)) < p0 else 1
            if a_new == 1 and self.w_sampler == 'ars':
                w_new = self._adaptive_rejection_sample_w(
                    lambda w: h.gibbs_ll(n_pre, aw_cur, np.array([w]))[0], mu_w, sigma_w, W_nns, log_L,
                    ll_of_ws=lambda wv: h.gibbs_ll(n_pre, aw_cur, np.asarray(wv)))
            elif a_new == 1:
                grid = mu_w + sigma_w * np.linspace(-4.0, 4.0, self.n_grid)
                w_new = self._inverse_cdf_sample_w(mu_w, sigma_w, grid,
                                                   h.gibbs_ll(n_pre, aw_cur, grid))
            else:
                w_new = mu_w + sigma_w * self.rng.standard_normal()                 # gibbs.py:1060-1062
            h.gibbs_update(n_pre, a_new * w_new - aw_cur)
            A[n_pre, n_post] = a_new
            W[n_pre, n_post] = w_new
            stats.append((int(n_pre), log_G, ll_noA))
        x['net']['graph']['A'] = A
        x['net']['weights']['W'] = W.ravel()
        return stats


class _SequenceSum(object):
    """The device state of the collapsed column update for ALL data sequences of a population: the inner ll
    is the sum over `population.data_sequences` (gibbs.py:899-903, 931-935) -- one resident handle per
    sequence, every prepare / rank-1 update applied to each of them, the (columns x weights) ll blocks added."""

    def __init__(self, population):
        if not population.data_sequences:
            raise Exception("No data sequence has been added")
        self.hs = []
        for data in population.data_sequences:
            population.set_data(data)
            self.hs.append(population._handle(data))

    def gibbs_prepare_all(self, theta, Weff):
        for h in self.hs:
            h.gibbs_prepare_all(theta, Weff)

    def gibbs_ll_cols(self, n_post, n_pre, aw_cur, w):
        out = self.hs[0].gibbs_ll_cols(n_post, n_pre, aw_cur, w)
        for h in self.hs[1:]:
            out = out + h.gibbs_ll_cols(n_post, n_pre, aw_cur, w)
        return out

    def gibbs_update_cols(self, n_post, n_pre, delta):
        for h in self.hs:
            h.gibbs_update_cols(n_post, n_pre, delta)

    def gibbs_prepare(self, n_post, theta_n, weff_col):
        for h in self.hs:
            h.gibbs_prepare(n_post, theta_n, weff_col)

    def gibbs_ll(self, n_pre, aw_cur, w):
        out = self.hs[0].gibbs_ll(n_pre, aw_cur, w)
        for h in self.hs[1:]:
            out = out + h.gibbs_ll(n_pre, aw_cur, w)
        return out

    def gibbs_update(self, n_pre, delta):
        for h in self.hs:
            h.gibbs_update(n_pre, delta)


def initialize_updates(population, rng=None, w_sampler='ars'):
    """gibbs.py:2413-2473: the latent-variable samplers of the reference are no-ops for models
    without latent components (the only ones on this path); every population gets the bias,
    stimulus, impulse and network-column updates, in this order."""
    serial_updates = []
    imp_cls = HmcDirichletImpulseUpdate if isinstance(population.glm.imp_model, DirichletImpulses) \
        else HmcImpulseUpdate
    parallel_updates = [HmcBiasUpdate(rng), HmcBkgdUpdate(rng), imp_cls(rng)]
    g = population.network.graph
    if hasattr(g, 'pA') and 'A' in population.get_variables()['net']['graph']:
        parallel_updates.append(CollapsedGibbsNetworkColumnUpdate(rng, w_sampler=w_sampler))
    for u in parallel_updates:
        u.preprocess(population)
    return serial_updates, parallel_updates


def gibbs_sample(population, N_samples=1000, x0=None, init_from_mle=True, callback=None,
                 lockstep=True, rng=None, verbose=True):
    """gibbs.py:2475-2560.  Returns the list of sampled states (x0 first).

    x0 None: prior draw, optionally replaced by a MAP fit of `standard_glm` on the same data
    converted to this model (gibbs.py:2490-2507).  lockstep=False runs every update neuron by
    neuron exactly in the reference's order."""
    N = population.model['N']
    dt = population.model['dt']
    rng = np.random if rng is None else rng
    if x0 is None:
        x0 = population.sample(rng if rng is not np.random else None)
        if init_from_mle and isinstance(population.glm.imp_model, DirichletImpulses):
            from theano_pyglm_amd.inference.coord_descent import coord_descent
            from theano_pyglm_amd.models.model_factory import make_model, convert_model
            from theano_pyglm_amd.population import Population
            if verbose:
                print("Initializing with coordinate descent")
            mle_model = make_model('standard_glm', N=N, dt=dt)
            mle_popn = Population(mle_model, device=population.device)
            for data in population.data_sequences:
                mle_popn.add_data(data)          # own handle: the MAP model has its own impulse basis
            mle_x0 = coord_descent(mle_popn, x0=mle_popn.sample(rng if rng is not np.random else None),
                                   maxiter=1, batched='torch')
            x0 = convert_model(mle_popn, mle_model, mle_x0, population, population.model, x0)
            mle_popn.release_data()              # the MAP population's own handles (resident features)
    serial_updates, parallel_updates = initialize_updates(population, rng)
    x = x0
    x_smpls = [copy.deepcopy(x0)]      # (the reference stores x0 itself, which its in-place updates then overwrite)
    start = time.time()
    for smpl in range(N_samples):
        if callback is not None:
            callback(x)
        lp = population.compute_log_p(x)
        stop = time.time()
        if verbose:
            print("Gibbs iteration %d. Iter/s = %f. Log prob: %.3f" % (smpl, 1.0 / max(stop - start, 1e-9), lp))
        start = stop
        for upd in parallel_updates:
            if lockstep:
                upd.update_all(x)
            else:
                for n in range(N):
                    upd.update(x, n)
        for upd in serial_updates:
            upd.update(x)
        x_smpls.append(copy.deepcopy(x))
    return x_smpls
