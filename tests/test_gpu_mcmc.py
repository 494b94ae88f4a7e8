"""
GPU tests of the MCMC updates (inference/gibbs.py): the HMC blocks driven by the batched device
ll+grad, ARS for the synaptic weights on top of the device inner-ll, and the gibbs_sample driver
(counterparts of gibbs.py:164-773, 1087-1126, 2413-2560).  Sampler trajectories cannot be compared
with the reference (un-vendored `hips`, unseeded streams) -- the tests pin invariants:
the sampled conditional matches a Laplace approximation built from the device gradient, ARS and
the reference's inverse-CDF sampler agree in distribution, state stays consistent with the oracle.
"""
import copy

import numpy as np
import pytest
from scipy import stats

from theano_pyglm_amd.harness.generate_synth_data import make_dataset
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference import gibbs as G
from theano_pyglm_amd.models.model_factory import make_model, convert_model
from theano_pyglm_amd.population import Population

from tests.test_gpu_population import oracle_log_p

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def std3():
    model, popn, data = make_dataset('standard_glm', 3, 8.0, seed=5)
    x_map = cd.coord_descent(popn, x0=popn.sample(np.random.RandomState(6)), maxiter=1, batched='torch')
    return model, popn, data, x_map


def test_hmc_bias_block_matches_laplace(std3):
    """Lock-step HMC on the biases: the chain's mean / spread equal the conditional posterior's
    mode / curvature (Laplace: near-Gaussian for thousands of bins), per neuron."""
    model, popn, data, x_map = std3
    x = copy.deepcopy(x_map)
    upd = G.HmcBiasUpdate(rng=np.random.RandomState(7))
    upd.preprocess(popn)
    assert (upd.lo, upd.hi) == (0, 1)
    # Laplace approximation from the device gradient: d lp / d bias by central differences of grad
    eps = 1e-3
    sd = np.zeros(3)
    for n in range(3):
        xp, xm = copy.deepcopy(x), copy.deepcopy(x)
        xp['glms'][n]['bias']['bias'] = x['glms'][n]['bias']['bias'] + eps
        xm['glms'][n]['bias']['bias'] = x['glms'][n]['bias']['bias'] - eps
        H = (popn.compute_grad(xp, n)[0] - popn.compute_grad(xm, n)[0]) / (2 * eps)
        assert H < 0
        sd[n] = 1.0 / np.sqrt(-H)
    mode = np.array([x['glms'][n]['bias']['bias'][0] for n in range(3)])
    tr = []
    for it in range(400):
        upd.update_all(x)
        tr.append([x['glms'][n]['bias']['bias'][0] for n in range(3)])
    tr = np.array(tr[50:])
    assert upd.n_evals == 400 * 11                     # 1 + n_steps batched evaluations per transition
    assert 1e-3 <= upd.step_sz <= 1.0 and upd.avg_accept_rate > 0.3
    for n in range(3):
        assert abs(tr[:, n].mean() - mode[n]) < 0.5 * sd[n]
        assert 0.6 * sd[n] < tr[:, n].std() < 1.5 * sd[n]
    # the other blocks were not touched
    for n in range(3):
        assert np.array_equal(x['glms'][n]['imp']['w_ir'], x_map['glms'][n]['imp']['w_ir'])


def test_hmc_impulse_block_and_per_neuron_form(std3):
    model, popn, data, x_map = std3
    x = copy.deepcopy(x_map)
    upd = G.HmcImpulseUpdate(rng=np.random.RandomState(8))
    upd.preprocess(popn)
    assert (upd.lo, upd.hi) == (1, 1 + 3 * 5)
    lp0 = popn.compute_log_p(x)
    upd.update(x, 1)                                   # per-neuron form: only neuron 1 may move
    assert np.array_equal(x['glms'][0]['imp']['w_ir'], x_map['glms'][0]['imp']['w_ir'])
    assert np.array_equal(x['glms'][2]['imp']['w_ir'], x_map['glms'][2]['imp']['w_ir'])
    moved = 0
    for it in range(60):
        upd.update_all(x)
        moved += 1
    lp1 = popn.compute_log_p(x)
    assert np.isfinite(lp1) and lp1 > lp0 - 60.0       # stays in the typical set around the mode
    assert not np.array_equal(x['glms'][1]['imp']['w_ir'], x_map['glms'][1]['imp']['w_ir'])
    assert np.allclose(lp1, oracle_log_p(popn, data, x)[0], rtol=1e-9)
    bk = G.HmcBkgdUpdate(rng=np.random.RandomState(9))
    bk.preprocess(popn)
    assert bk.lo == bk.hi and bk.update_all(x) is x and bk.n_evals == 0    # NoStimulus: no-op


@pytest.fixture(scope='module')
def swm4():
    def tame(x):
        x['net']['weights']['W'] = 0.3 * np.asarray(x['net']['weights']['W'])
    model = make_model('sparse_weighted_model', N=4, dt=0.001)
    model['network']['graph']['rho'] = 0.4
    model, popn, data = make_dataset(model, 4, 8.0, seed=14, adjust=tame, check=False)
    popn.add_data(data)
    return model, popn, data


def test_ars_weight_draw_matches_inverse_cdf(swm4):
    """W[n_pre,n_post] | A=1 drawn by ARS over the device inner ll agrees in distribution with the
    reference's inverse-CDF sampler (gibbs.py:1068-1084) on a fine grid."""
    model, popn, data = swm4
    x = copy.deepcopy(data['vars'])
    upd = G.CollapsedGibbsNetworkColumnUpdate(rng=np.random.RandomState(15))
    upd.preprocess(popn)
    n_post, n_pre = 1, 2
    h = popn._handle(popn._current)
    A = np.asarray(x['net']['graph']['A'])
    W = np.asarray(x['net']['weights']['W'], float).reshape(4, 4)
    h.gibbs_prepare(n_post, popn.glm.theta_row(x['glms'][n_post]), (A * W)[:, n_post])
    aw = float(A[n_pre, n_post] * W[n_pre, n_post])
    mu_w, sigma_w = upd.mu_w, upd.sigma_w
    W_nns = np.sqrt(2) * sigma_w * upd.GAUSS_HERMITE_ABSCISSAE + mu_w
    log_L = h.gibbs_ll(n_pre, aw, W_nns)
    ars = np.array([upd._adaptive_rejection_sample_w(lambda w: h.gibbs_ll(n_pre, aw, np.array([w]))[0],
                                                     mu_w, sigma_w, W_nns, log_L)
                    for _ in range(200)])
    assert upd.n_ars_evals < 200 * 12                   # a handful of extra abscissae per draw (sharp posterior)
    # with the batched refinement of the hull around the mode (14 weights per launch): fewer launches per draw
    n0 = upd.n_ars_evals
    ars2 = np.array([upd._adaptive_rejection_sample_w(lambda w: h.gibbs_ll(n_pre, aw, np.array([w]))[0],
                                                      mu_w, sigma_w, W_nns, log_L,
                                                      ll_of_ws=lambda wv: h.gibbs_ll(n_pre, aw, np.asarray(wv)))
                     for _ in range(200)])
    assert upd.n_ars_evals - n0 <= n0
    ars = np.concatenate((ars, ars2))
    assert stats.ks_2samp(ars[:200], ars2).pvalue > 1e-3
    grid = mu_w + sigma_w * np.linspace(-5.0, 5.0, 801)
    ll_grid = np.concatenate([h.gibbs_ll(n_pre, aw, grid[i:i + 16]) for i in range(0, 801, 16)])
    icdf = np.array([upd._inverse_cdf_sample_w(mu_w, sigma_w, grid, ll_grid) for _ in range(4000)])
    assert stats.ks_2samp(ars, icdf).pvalue > 1e-3
    # the likelihood really shapes the draw: it is not the prior
    assert icdf.std() < 0.8 * sigma_w


def test_gibbs_sample_driver(swm4):
    model, popn, data = swm4
    rng = np.random.RandomState(16)
    seen = []
    smpls = G.gibbs_sample(popn, N_samples=4, x0=None, init_from_mle=True, callback=seen.append,
                           rng=rng, verbose=False)
    assert len(smpls) == 5 and len(seen) == 4
    x = smpls[-1]
    A = np.asarray(x['net']['graph']['A'])
    assert A.shape == (4, 4) and set(np.unique(A)) <= {0, 1}
    assert np.asarray(x['net']['weights']['W']).shape == (16,)
    lps = [popn.compute_log_p(s) for s in smpls]
    assert np.all(np.isfinite(lps))
    assert np.allclose(lps[-1], oracle_log_p(popn, data, x)[0], rtol=1e-9)
    # the converted MAP initialisation (model_factory.convert_model) is a valid state of this model
    x0 = smpls[0]
    for n in range(4):
        for k in range(4):
            g = x0['glms'][n]['imp']['g_%d' % k]
            assert g.shape == (5,) and np.all(g > 0) and np.isclose(g.sum(), popn.glm.imp_model.alpha * 5)
    # sampling moves every block
    assert not np.array_equal(smpls[1]['glms'][0]['bias']['bias'], smpls[0]['glms'][0]['bias']['bias']) \
        or not np.array_equal(smpls[2]['glms'][0]['bias']['bias'], smpls[0]['glms'][0]['bias']['bias'])
    # neuron-by-neuron order of the reference gives a valid chain too
    s2 = G.gibbs_sample(popn, N_samples=1, x0=copy.deepcopy(x), lockstep=False,
                        rng=np.random.RandomState(17), verbose=False)
    assert np.isfinite(popn.compute_log_p(s2[-1]))


def test_dirichlet_impulse_update_respects_graph(swm4):
    model, popn, data = swm4
    x = copy.deepcopy(data['vars'])
    A = np.zeros((4, 4), dtype=np.int8)
    A[0, 1] = A[2, 1] = A[3, 3] = 1
    x['net']['graph']['A'] = A
    x0 = copy.deepcopy(x)
    upd = G.HmcDirichletImpulseUpdate(rng=np.random.RandomState(18))
    upd.preprocess(popn)
    upd.update_all(x)
    assert upd.n_evals == 2 * 3                         # two lock-step rounds (max in-degree 2) x (1+2) evals
    # no edge -> fresh prior draw (changes with probability one); edge -> HMC (moves or stays)
    assert not np.array_equal(x['glms'][0]['imp']['g_1'], x0['glms'][0]['imp']['g_1'])
    assert not np.array_equal(x['glms'][1]['imp']['g_1'], x0['glms'][1]['imp']['g_1'])
    assert np.isfinite(popn.compute_log_p(x))
    for n in range(4):
        for k in range(4):
            assert x['glms'][n]['imp']['g_%d' % k].shape == (5,)


def test_collapsed_gibbs_over_two_data_sequences():
    """gibbs.py:899-903, 931-935: the inner ll of the collapsed (A, W) column update is the SUM over the
    population's data sequences.  Two sequences of different length on one population: the batched and the
    single-column device paths against the oracle's inner ll summed over both, a full column sweep, and the
    gibbs_sample driver on the two-sequence population."""
    from oracle import glm_oracle as O

    def tame(x):
        x['net']['weights']['W'] = 0.3 * np.asarray(x['net']['weights']['W'])
    N = 4
    model = make_model('sparse_weighted_model', N=N, dt=0.001)
    model['network']['graph']['rho'] = 0.4
    model, popn, d1 = make_dataset(model, N, 5.0, seed=21, adjust=tame, check=False)
    _, _, d2 = make_dataset(model, N, 3.0, seed=22, adjust=tame, check=False)
    popn.add_data(d1)
    popn.add_data(d2)
    assert len(popn.data_sequences) == 2
    x = copy.deepcopy(d1['vars'])
    A = np.asarray(x['net']['graph']['A']).reshape(N, N).astype(float)
    W = np.asarray(x['net']['weights']['W'], float).reshape(N, N)
    glm = popn.glm
    rng = np.random.RandomState(23)

    def oracle_inner(n_pre, n_post, ws):
        out = np.zeros(len(ws))
        xn = x['glms'][n_post]
        beta = glm.imp_model.flat_weights(xn['imp']).reshape(N, -1)
        for d in (d1, d2):
            S = np.asarray(d['S'], float)
            I_imp = O.impulse_currents(O.convolve_with_basis_fft(S, glm.imp_model.ibasis), beta)
            I_other = O.other_current(I_imp, A, W, n_pre, n_post)
            out += O.mcmc_inner_ll(ws, glm.bias_model.I_bias(xn['bias']), 0.0, I_other, I_imp[:, n_pre],
                                   S[:, n_post], glm.dt, glm.nlin_model.kind)
        return out

    hs = G._SequenceSum(popn)
    assert len(hs.hs) == 2
    # batched path: one pair of every column per launch
    hs.gibbs_prepare_all(popn.theta_matrix(x), A * W)
    cols, n_pre = np.arange(N), np.array([1, 2, 3, 0])
    aw = (A * W)[n_pre, cols]
    ws = rng.standard_normal((N, 11))
    ll = hs.gibbs_ll_cols(cols, n_pre, aw, ws)
    for c in range(N):
        assert np.allclose(ll[c], oracle_inner(n_pre[c], c, ws[c]), rtol=1e-10), c
    # single-column path
    n_post = 2
    hs.gibbs_prepare(n_post, glm.theta_row(x['glms'][n_post]), (A * W)[:, n_post])
    for k in (0, 2, 3):
        got = hs.gibbs_ll(k, float((A * W)[k, n_post]), ws[0])
        assert np.allclose(got, oracle_inner(k, n_post, ws[0]), rtol=1e-10), k
    # rank-1 updates reach both handles: changing one pair == re-preparing with the new weights
    delta = 0.37
    hs.gibbs_prepare_all(popn.theta_matrix(x), A * W)
    hs.gibbs_update_cols(np.array([1]), np.array([3]), np.array([delta]))
    AW2 = (A * W).copy()
    AW2[3, 1] += delta
    ll_upd = hs.gibbs_ll_cols(cols, n_pre, AW2[n_pre, cols], ws)
    hs.gibbs_prepare_all(popn.theta_matrix(x), AW2)
    assert np.allclose(ll_upd, hs.gibbs_ll_cols(cols, n_pre, AW2[n_pre, cols], ws), rtol=1e-11)
    # a sweep of all columns (lock step) and one reference-order column update on the two sequences
    upd = G.CollapsedGibbsNetworkColumnUpdate(rng=np.random.RandomState(24))
    upd.preprocess(popn)
    xs = copy.deepcopy(x)
    upd.update_all(xs)
    st = upd.update(xs, 0)
    assert len(st) == N
    lp = popn.compute_log_p(xs)
    lp1, _ = oracle_log_p(popn, d1, xs)
    _, ll2 = oracle_log_p(popn, d2, xs)
    assert np.isclose(lp, lp1 + np.sum(ll2), rtol=1e-9)
    smpls = G.gibbs_sample(popn, N_samples=2, x0=copy.deepcopy(xs), rng=np.random.RandomState(25), verbose=False)
    assert len(smpls) == 3 and np.isfinite(popn.compute_log_p(smpls[-1]))
    popn.release_data()
