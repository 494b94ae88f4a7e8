"""CPU tests of the in-repo samplers that replace the un-vendored `hips` package
(gibbs.py:14-15): the invariant distributions are pinned with Kolmogorov-Smirnov tests."""
import numpy as np
from scipy import stats

from theano_pyglm_amd.inference.ars import adaptive_rejection_sample
from theano_pyglm_amd.inference.hmc import hmc, hmc_lockstep, adapt_step_size


def test_ars_gaussian_from_quadrature_nodes():
    """the reference's use: start from the 10 Gauss-Hermite nodes (gibbs.py:1087-1126)."""
    rng = np.random.RandomState(0)
    mu, sig = 0.7, 1.3
    f = lambda w: -0.5 * (w - mu) ** 2 / sig ** 2
    xs = np.sqrt(2) * 2.0 * np.polynomial.hermite.hermgauss(10)[0]
    draws, evals = [], 0
    for _ in range(1500):
        t, ne = adaptive_rejection_sample(f, xs, f(xs), (-np.inf, np.inf), stepsz=1.0, rng=rng,
                                          return_evals=True)
        draws.append(t)
        evals += ne
    assert stats.kstest(draws, 'norm', args=(mu, sig)).pvalue > 1e-3
    assert evals / 1500.0 < 2.0                      # the 10-node hull is already tight


def test_ars_needs_bracketing_and_bounded_domain():
    rng = np.random.RandomState(1)
    # all starting points left of the mode: the sampler must extend to the right
    f = lambda w: -0.5 * (w - 5.0) ** 2
    d = [adaptive_rejection_sample(f, [-1.0, 0.0], [f(-1.0), f(0.0)], stepsz=0.5, rng=rng)
         for _ in range(800)]
    assert stats.kstest(d, 'norm', args=(5.0, 1.0)).pvalue > 1e-3
    # Gamma(3,1) on (0, inf): finite left edge, skewed
    g = lambda w: 2.0 * np.log(w) - w if w > 0 else -np.inf
    d = [adaptive_rejection_sample(g, [0.5, 2.0, 6.0], [g(0.5), g(2.0), g(6.0)], (0.0, np.inf),
                                   stepsz=1.0, rng=rng) for _ in range(800)]
    assert min(d) > 0 and stats.kstest(d, 'gamma', args=(3.0,)).pvalue > 1e-3
    # bounded on both sides: truncated exponential (linear log density -> exact hull)
    e = lambda w: -2.0 * w
    d = [adaptive_rejection_sample(e, [0.1, 0.5, 0.9], [e(0.1), e(0.5), e(0.9)], (0.0, 1.0), rng=rng)
         for _ in range(800)]
    cdf = lambda t: (1 - np.exp(-2 * np.asarray(t))) / (1 - np.exp(-2.0))
    assert 0 <= min(d) and max(d) <= 1 and stats.kstest(d, cdf).pvalue > 1e-3


def test_ars_drops_nonfinite_starting_values():
    rng = np.random.RandomState(2)
    f = lambda w: -0.5 * w * w
    xs = np.array([-3.0, -1.0, 0.5, 2.0, 30.0])
    vs = np.array([f(-3.0), f(-1.0), f(0.5), f(2.0), -np.inf])
    d = [adaptive_rejection_sample(f, xs, vs, rng=rng) for _ in range(500)]
    assert stats.kstest(d, 'norm').pvalue > 1e-3


def test_hmc_single_chain_invariant_distribution():
    rng = np.random.RandomState(3)
    C = np.array([[1.0, 0.6], [0.6, 2.0]])
    Ci = np.linalg.inv(C)
    U = lambda q: 0.5 * q.dot(Ci).dot(q)
    gU = lambda q: Ci.dot(q)
    q, step, rate = np.zeros(2), 0.3, 0.9
    out = []
    for i in range(4000):
        q, step, rate = hmc(U, gU, step, 8, q, adaptive_step_sz=True, avg_accept_rate=rate, rng=rng)
        out.append(q.copy())
    out = np.array(out[500:])
    assert np.allclose(out.mean(0), 0, atol=0.15)
    assert np.allclose(np.cov(out.T), C, atol=0.3)
    assert 1e-3 <= step <= 1.0 and 0.5 < rate <= 1.0
    assert hmc(U, gU, 0.1, 3, np.ones(2), rng=rng).shape == (2,)       # non-adaptive form


def test_hmc_lockstep_matches_independent_targets():
    rng = np.random.RandomState(4)
    M = 6
    mus = np.linspace(-2, 2, M)[:, None] * np.ones((M, 3))
    sig = np.linspace(0.5, 1.5, M)[:, None]
    calls = [0]

    def UG(Q):
        calls[0] += 1
        Z = (Q - mus) / sig
        return 0.5 * np.sum(Z * Z, axis=1), Z / sig

    Q = np.zeros((M, 3))
    active = np.array([True] * (M - 1) + [False])
    tr = []
    for i in range(2500):
        Q, acc, ne = hmc_lockstep(UG, 0.35, 5, Q, active=active, rng=rng)
        assert ne == 6 and not acc[-1]
        tr.append(Q.copy())
    tr = np.array(tr[300:])
    assert np.all(tr[:, -1, :] == 0.0)                          # frozen chain never moves
    for m in range(M - 1):
        assert abs(tr[:, m, :].mean() - mus[m, 0]) < 0.15
        assert abs(tr[:, m, :].std() - sig[m, 0]) < 0.15
    assert calls[0] == 2500 * 6


def test_step_size_controller():
    s, r = adapt_step_size(0.1, 0.95, True)
    assert np.isclose(s, 0.102) and np.isclose(r, 0.95 * 0.95 + 0.05)
    s, r = adapt_step_size(0.1, 0.5, False)
    assert np.isclose(s, 0.098) and np.isclose(r, 0.475)
    assert adapt_step_size(1.0, 0.99, True)[0] == 1.0 and adapt_step_size(1e-3, 0.1, False)[0] == 1e-3
