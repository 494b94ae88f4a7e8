"""
GPU tests of the host mirror (Population / Glm / coord_descent / Gibbs inner loop)
against the CPU oracle -- they read like the reference's own scripts (test/synth_map.py,
test/generate_synth_data.py) with assertions instead of plots.
"""
import copy

import numpy as np
import pytest

from oracle import glm_oracle as O
from theano_pyglm_amd.harness.generate_synth_data import make_dataset
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.gibbs import CollapsedGibbsNetworkColumnUpdate
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars

pytestmark = pytest.mark.gpu


def oracle_log_p(popn, data, x):
    """compute_log_p restated with the oracle (population.py:34-86)."""
    glm = popn.glm
    N = popn.N
    S = np.asarray(data['S'], dtype=float)
    fS = O.convolve_with_basis_fft(S, glm.imp_model.ibasis)
    fstim = None
    if glm.Dstim > 0:           # oracle features from the raw stimulus (bkgd.py:122-154 / 303-340)
        bk = glm.bkgd_model
        if hasattr(bk, 'ibasis_x'):
            fstim = O.spatiotemporal_stim_features(np.asarray(data['stim'], float), data['dt_stim'], glm.dt,
                                                   S.shape[0], bk.ibasis_x, bk.ibasis_t)
        else:
            fstim = O.basis_stim_features(np.asarray(data['stim'], float), data['dt_stim'], glm.dt,
                                          S.shape[0], bk.ibasis)
    Weff = popn.W_eff(x)
    lp = popn.network.log_p(x['net'])
    lls = []
    for n in range(N):
        xn = x['glms'][n]
        lp += glm.log_prior(xn)
        w = glm.imp_model.flat_weights(xn['imp']).reshape(N, -1)
        ws = glm.bkgd_model.dense_weights(xn['bkgd']) if fstim is not None else None
        lls.append(O.glm_ll(n, S, fS, w, Weff[:, n], glm.bias_model.I_bias(xn['bias']), glm.dt,
                            glm.nlin_model.kind, fstim, ws))
    return lp + np.sum(lls), np.array(lls)


def fd_grad(f, v, eps=1e-6):
    g = np.zeros_like(v)
    for i in range(len(v)):
        e = np.zeros_like(v)
        e[i] = eps
        g[i] = (f(v + e) - f(v - e)) / (2 * eps)
    return g


@pytest.fixture(scope='module')
def std4():
    return make_dataset('standard_glm', 4, 6.0, seed=3)       # also runs the lam==lam_sim check


def test_generate_synth_data_invariant(std4):
    """test/generate_synth_data.py:125-129 through the device path (asserted in make_dataset)."""
    model, popn, data = std4
    assert data['S'].shape == (6000, 4) and data['S'].sum() > 50


def test_compute_log_p_standard(std4):
    model, popn, data = std4
    x = popn.sample(np.random.RandomState(5))
    lp = popn.compute_log_p(x)
    lp0, lls = oracle_log_p(popn, data, x)
    assert np.allclose(lp, lp0, rtol=1e-10)
    assert np.allclose(popn.compute_ll_vector(x), lls, rtol=1e-10)


def test_compute_grad_matches_fd_and_oracle(std4):
    """compute_grad == -grad_nlp (coord_descent.py:61-80) checked by central differences of
    the oracle log posterior."""
    model, popn, data = std4
    x = popn.sample(np.random.RandomState(6))
    n = 2
    syms = popn.glm_syms()
    v0, shapes = packdict(get_vars(syms, x['glms'][n]))
    assert v0.size == 1 + 5 * 4                                 # [bias, w_ir]
    g = popn.compute_grad(x, n)

    def lp_of(v):
        x2 = copy.deepcopy(x)
        set_vars(syms, x2['glms'][n], unpackdict(v, shapes))
        return oracle_log_p(popn, data, x2)[0]

    g_fd = fd_grad(lp_of, v0)
    assert np.max(np.abs(g - g_fd)) < 1e-4 * max(1.0, np.max(np.abs(g_fd)))


def test_fit_glm_and_coord_descent(std4):
    model, popn, data = std4
    x0 = popn.sample(np.random.RandomState(7))
    lp0 = popn.compute_log_p(x0)
    x_seq = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched=False)
    lp_seq = popn.compute_log_p(x_seq)
    x_bat = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched=True)
    lp_bat = popn.compute_log_p(x_bat)
    assert lp_seq > lp0 + 1.0 and lp_bat > lp0 + 1.0
    # both optimisers reach the same concave optimum
    assert abs(lp_seq - lp_bat) < 1e-2 * max(1.0, abs(lp_seq) * 1e-3)
    # GPU-resident optimizer state (torch plumbing around the same HIP ll+grad)
    x_t = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    lp_t = popn.compute_log_p(x_t)
    assert abs(lp_t - lp_bat) < 1e-6 * max(1.0, abs(lp_bat))
    # and the oracle agrees on the fitted state
    assert np.allclose(lp_bat, oracle_log_p(popn, data, x_bat)[0], rtol=1e-9)


def test_nan_semantics(std4):
    """A zero impulse-weight group makes the group-lasso gradient NaN (priors.py:202);
    fit_glm zeroes it like coord_descent.py:179-180 and still returns."""
    model, popn, data = std4
    x = popn.sample(np.random.RandomState(8))
    x['glms'][0]['imp']['w_ir'] = np.zeros_like(x['glms'][0]['imp']['w_ir'])
    g = popn.compute_grad(x, 0)
    assert np.any(np.isnan(g))
    prms = cd.prep_first_order_glm_inference(popn)
    nv = popn.extract_vars(x, 0)
    res = cd.fit_glm(nv, 0, prms, maxiter=3)
    assert np.all(np.isfinite(res.x))


def test_sparse_weighted_model_log_p_and_gibbs():
    model, popn, data = make_dataset('sparse_weighted_model', 6, 4.0, seed=11)
    rng = np.random.RandomState(12)
    x = popn.sample(rng)
    lp = popn.compute_log_p(x)
    lp0, _ = oracle_log_p(popn, data, x)
    assert np.allclose(lp, lp0, rtol=1e-10)
    # Dirichlet chain rule: packed gradient vs central differences of the oracle
    n = 1
    syms = popn.glm_syms()
    v0, shapes = packdict(get_vars(syms, x['glms'][n]))
    assert v0.size == 1 + 6 * 5
    g = popn.compute_grad(x, n)

    def lp_of(v):
        x2 = copy.deepcopy(x)
        set_vars(syms, x2['glms'][n], unpackdict(v, shapes))
        return oracle_log_p(popn, data, x2)[0]

    g_fd = fd_grad(lp_of, v0)
    assert np.max(np.abs(g - g_fd)) < 1e-4 * max(1.0, np.max(np.abs(g_fd)))
    # collapsed Gibbs inner loop: reference-shaped helpers vs oracle quadrature inputs
    upd = CollapsedGibbsNetworkColumnUpdate(rng=np.random.RandomState(13))
    upd.preprocess(popn)
    n_post, n_pre = 2, 4
    I_bias, I_stim, I_imp, p_A = upd._precompute_vars(x, n_post)
    I_other = upd._precompute_other_current(x, I_imp, n_pre, n_post)
    W_nns, _ = O.gauss_hermite_nodes(upd.mu_w, upd.sigma_w)
    ll_dev = upd._glm_ll(n_pre, n_post, W_nns, x, I_bias, I_stim, I_imp, I_other)
    fS = O.convolve_with_basis_fft(np.asarray(data['S'], float), popn.glm.imp_model.ibasis)
    w = popn.glm.imp_model.flat_weights(x['glms'][n_post]['imp']).reshape(6, -1)
    I_imp0 = O.impulse_currents(fS, w)
    A = np.asarray(x['net']['graph']['A'], float)
    Wm = np.asarray(x['net']['weights']['W']).reshape(6, 6)
    I_other0 = O.other_current(I_imp0, A, Wm, n_pre, n_post)
    ll_ref = O.mcmc_inner_ll(W_nns, I_bias, 0.0, I_other0, I_imp0[:, n_pre],
                             np.asarray(data['S'], float)[:, n_post], popn.glm.dt, 'explinear')
    assert np.allclose(ll_dev, ll_ref, rtol=1e-9)
    # a full column update keeps the state consistent: device log_p == oracle log_p afterwards
    stats = upd.update(x, n_post)
    assert len(stats) == 6
    assert set(np.unique(x['net']['graph']['A'])) <= {0, 1}
    assert np.allclose(popn.compute_log_p(x), oracle_log_p(popn, data, x)[0], rtol=1e-9)


def test_device_stimulus_features(golden):
    """pgl_set_stimulus (interp + projection + causal temporal filtering on the GPU) against the
    reference's convolve_with_low_rank_2d_basis golden vectors and against the oracle."""
    from theano_pyglm_amd import _lib
    stim, ibx, ibt = golden['lr2d_stim'], golden['lr2d_ibasis_x'], golden['lr2d_ibasis_t']
    T = stim.shape[0]
    d = _lib.DeviceGlm(2, T, 3, 300, 'exp', 0.001)
    d.set_stimulus(stim, 0.001, ibt, ibx, layout=0)               # dt_stim == dt: no interpolation
    f = d.get_stim_features().reshape(T, 3, 3)                    # (t, bt, bx)
    assert np.max(np.abs(np.transpose(f, (0, 2, 1)) - golden['lr2d_fstim'])) < 1e-12
    d.set_stimulus(stim, 0.001, ibt, None, layout=1)              # BasisStimulus layout d*B+b
    f1 = d.get_stim_features()
    assert np.max(np.abs(f1 - O.basis_stim_features(stim, 0.001, 0.001, T, ibt))) < 1e-12
    d.close()
    # with interpolation from a coarse grid (dt_stim = 0.1 s), ragged end
    rng = np.random.RandomState(3)
    nT = 2345
    coarse = rng.randn(24, 3)
    d = _lib.DeviceGlm(2, nT, 3, 300, 'exp', 0.001)
    d.set_stimulus(coarse, 0.1, ibt, ibx, layout=0)
    ref = O.spatiotemporal_stim_features(coarse, 0.1, 0.001, nT, ibx, ibt)
    assert np.max(np.abs(d.get_stim_features() - ref)) < 1e-11
    d.close()


def test_spatiotemporal_glm():
    def tame(x):                               # keep rates finite under the exp nonlinearity
        for xn in x['glms']:
            xn['bias']['bias'] = np.array([2.0])
            xn['bkgd']['w_x'] = xn['bkgd']['w_x'] * 0.2
            xn['bkgd']['w_t'] = xn['bkgd']['w_t'] * 0.2

    model, popn, data = make_dataset('spatiotemporal_glm', 4, 3.0, seed=21, check=True, adjust=tame)
    assert data['S'].sum() > 5
    x = popn.sample(np.random.RandomState(22))
    tame(x)
    lp = popn.compute_log_p(x)
    lp0, _ = oracle_log_p(popn, data, x)
    assert np.isfinite(lp) and np.allclose(lp, lp0, rtol=1e-10)
    n = 3
    syms = popn.glm_syms()
    v0, shapes = packdict(get_vars(syms, x['glms'][n]))
    assert v0.size == 1 + 3 + 3 + 3 * 4         # [bias, w_t, w_x, w_ir]  
    g = popn.compute_grad(x, n)

    def lp_of(v):
        x2 = copy.deepcopy(x)
        set_vars(syms, x2['glms'][n], unpackdict(v, shapes))
        return oracle_log_p(popn, data, x2)[0]

    g_fd = fd_grad(lp_of, v0, eps=1e-6)
    assert np.max(np.abs(g - g_fd)) < 2e-4 * max(1.0, np.max(np.abs(g_fd)))


def test_basis_stimulus_model():
    """standard_glm with the BasisStimulus background (bkgd.py:45-169): device-built features,
    log p and packed gradient [bias, w_stim, w_ir] against the oracle."""
    from theano_pyglm_amd.models import templates
    tmpl = templates.standard_glm()
    tmpl['bkgd']['type'] = 'basis'
    model, popn, data = make_dataset(tmpl, 3, 4.0, seed=31, check=True)
    assert popn.glm.Dstim == 3 and popn.stim_features().shape == (4000, 3)
    x = popn.sample(np.random.RandomState(32))
    lp = popn.compute_log_p(x)
    lp0, _ = oracle_log_p(popn, data, x)
    assert np.allclose(lp, lp0, rtol=1e-10)
    n = 1
    syms = popn.glm_syms()
    v0, shapes = packdict(get_vars(syms, x['glms'][n]))
    assert v0.size == 1 + 3 + 5 * 3
    g = popn.compute_grad(x, n)

    def lp_of(v):
        x2 = copy.deepcopy(x)
        set_vars(syms, x2['glms'][n], unpackdict(v, shapes))
        return oracle_log_p(popn, data, x2)[0]

    g_fd = fd_grad(lp_of, v0, eps=1e-7)
    assert np.max(np.abs(g - g_fd)) < 1e-3 * max(1.0, np.max(np.abs(g_fd)))
    # the GPU-resident batched optimizer handles the stimulus weights too
    x_t = cd.coord_descent(popn, x0=copy.deepcopy(x), maxiter=1, batched='torch')
    assert popn.compute_log_p(x_t) > lp


def test_wide_spatiotemporal_stimulus_sliced():
    """spatiotemporal_glm with a wide stimulus (D_stim = 256, identity spatial basis ->
    3*256 = 768 stimulus columns + 3*N impulse columns > 640): device feature build and the
    sliced ll+grad path against the oracle (a small version of the C5 stress variant)."""
    from theano_pyglm_amd.models import templates
    tmpl = templates.spatiotemporal_glm()
    tmpl['bkgd']['D_stim'] = 256
    tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': 256}
    tmpl['bkgd']['sigma'] = 0.05

    def tame(x):
        for xn in x['glms']:
            xn['bias']['bias'] = np.array([2.0])

    model, popn, data = make_dataset(tmpl, 4, 1.5, seed=41, check=True, adjust=tame)
    assert popn.glm.bkgd_model.separable and popn.glm.Dstim == 3 + 256     # rank-1 path: theta carries [w_t, w_x]
    x = popn.sample(np.random.RandomState(42))
    tame(x)
    lp = popn.compute_log_p(x)
    lp0, _ = oracle_log_p(popn, data, x)
    assert np.isfinite(lp) and np.allclose(lp, lp0, rtol=1e-10)
    # packed gradient [bias, w_t(3), w_x(256), w_ir(12)] along a random direction vs oracle differences
    n = 2
    syms = popn.glm_syms()
    v0, shapes = packdict(get_vars(syms, x['glms'][n]))
    assert v0.size == 1 + 3 + 256 + 12
    g = popn.compute_grad(x, n)
    d = np.random.RandomState(43).randn(v0.size)
    d /= np.linalg.norm(d)

    def lp_of(v):
        x2 = copy.deepcopy(x)
        set_vars(syms, x2['glms'][n], unpackdict(v, shapes))
        return oracle_log_p(popn, data, x2)[0]

    eps = 1e-6
    fd = (lp_of(v0 + eps * d) - lp_of(v0 - eps * d)) / (2 * eps)
    assert abs(fd - g.dot(d)) < 1e-4 * max(1.0, abs(fd))


def test_sta_matches_oracle():
    """pgl_sta (event-list STA on the device) against the dense lag-matrix restatement of
    pyglm/utils/sta.py, with interpolation from a coarse stimulus grid, a neuron subset, a
    multi-spike bin, spikes inside the first L bins and a silent neuron."""
    from theano_pyglm_amd.utils.sta import sta
    rng = np.random.RandomState(8)
    nT, N, D, L = 5000, 5, 3, 120
    S = (rng.rand(nT, N) < 0.03).astype(float)
    S[3, 0] = 2.0
    S[nT - 1, 4] = 1.0
    S[:, 2] = 0.0
    stim = rng.randn(nT // 100, D)
    data = {'S': S, 'dt': 0.001, 'dt_stim': 0.1, 'N': N, 'T': nT * 0.001}
    A = sta(stim, data, L, Ns=[0, 2, 4])
    A0 = O.sta(stim, S, 0.001, 0.1, L, [0, 2, 4])
    assert A.shape == (3, L, D)
    assert np.all(np.isnan(A[1])) and np.all(np.isnan(A0[1]))
    assert np.allclose(A[[0, 2]], A0[[0, 2]], rtol=1e-11, atol=1e-14)
    assert np.allclose(sta(stim, data, L, Ns=3)[0], O.sta(stim, S, 0.001, 0.1, L, [3])[0], rtol=1e-11, atol=1e-14)
    # wide stimulus, all neurons (many output tiles, one event chunk)
    stim2 = rng.randn(nT // 10, 200)
    data2 = {'S': S[:, [0, 1]], 'dt': 0.001, 'dt_stim': 0.01}
    assert np.allclose(sta(stim2, data2, 30), O.sta(stim2, S[:, [0, 1]], 0.001, 0.01, 30, [0, 1]),
                       rtol=1e-11, atol=1e-14)
    # the two device forms -- frame-rate (weights from the event lists + one thin GEMM with the raw stimulus; taken for
    # an integer dt_stim / dt and an even number of frames) and bin-rate (gather from the interpolated stimulus; dev
    # option 94 = 7, odd frame counts, other ratios) -- agree; spikes behind the last frame hold it (np.interp)
    from theano_pyglm_amd import _lib
    h = _lib.DeviceGlm(N, nT, 1, 1, 'exp', 0.001)
    h.set_spikes(S)
    for st, dts in ((stim, 0.1), (stim[:37], 0.1), (stim[:30], 0.1), (rng.randn(700, 4), 0.007)):
        Af = h.sta(st, dts, L, Ns=[0, 1, 3, 4])
        h.set_option(94, 7)
        Ab = h.sta(st, dts, L, Ns=[0, 1, 3, 4])
        h.set_option(94, 0)
        assert np.allclose(Af, Ab, rtol=1e-11, atol=1e-14)
        assert np.allclose(Af, O.sta(st, S, 0.001, dts, L, [0, 1, 3, 4]), rtol=1e-11, atol=1e-14)
    h.close()


def test_initialize_with_sta():
    """smart_init.initialize_stim_with_sta on the spatiotemporal and basis-stimulus models: device
    STA + host factorisation equal the oracle's, and the warm start beats the prior draw."""
    from theano_pyglm_amd.inference import smart_init
    from theano_pyglm_amd.models import templates

    def tame(x):
        for xn in x['glms']:
            xn['bias']['bias'] = np.array([2.0])
            xn['bkgd']['w_x'] = np.array([0.8, -0.4, 0.3]) * (1 + 0.1 * xn['n'])
            xn['bkgd']['w_t'] = np.array([1.0, 0.3, 0.1])
            xn['imp']['w_ir'] = xn['imp']['w_ir'] * 0.2

    model, popn, data = make_dataset('spatiotemporal_glm', 3, 20.0, seed=41, adjust=tame)
    x0 = popn.sample(np.random.RandomState(42))
    ll_prior_draw = popn.compute_ll(x0)
    smart_init.initialize_with_data(popn, data, x0)
    bk = popn.glm.bkgd_model
    A0 = O.sta(np.asarray(data['stim'], float), np.asarray(data['S'], float), data['dt'], data['dt_stim'],
               bk.ibasis_t.shape[0], range(3))
    for n in range(3):
        w0 = O.sta_stim_weights(A0[n], 'spatiotemporal', bk.ibasis_t, bk.ibasis_x)
        w = x0['glms'][n]['bkgd']
        # the singular pair is defined up to a common sign
        assert np.allclose(np.outer(w['w_t'], w['w_x']), np.outer(w0['w_t'], w0['w_x']), rtol=1e-7, atol=1e-10)
    assert popn.compute_ll(x0) > ll_prior_draw
    # basis-stimulus model
    tmpl = templates.standard_glm()
    tmpl['bkgd']['type'] = 'basis'
    model, popn, data = make_dataset(tmpl, 3, 6.0, seed=43)
    x0 = popn.sample(np.random.RandomState(44))
    smart_init.initialize_stim_with_sta(popn, data, x0, Ns=1)
    bk = popn.glm.bkgd_model
    A0 = O.sta(np.asarray(data['stim'], float), np.asarray(data['S'], float), data['dt'],
               data['dt_stim'], bk.ibasis.shape[0], [1])
    assert np.allclose(x0['glms'][1]['bkgd']['w_stim'],
                       O.sta_stim_weights(A0[0], 'basis', bk.ibasis)['w_stim'], rtol=1e-8, atol=1e-12)
    assert np.isfinite(popn.compute_log_p(x0))


def test_map_with_cross_validation(std4):
    """harness/synth_map_with_xv: group-lasso lam grid (synth_harness.py:76), three resident data
    handles (train / held-out / all), hyper-parameters switched on the host only."""
    from theano_pyglm_amd.harness.synth_map_with_xv import get_xv_models, run_xv
    model, popn, data = std4
    models = get_xv_models(make_model('standard_glm', N=4, dt=0.001))
    assert [m['impulse']['prior']['lam'] for m in models] == [0.5, 1.0, 2.0, 3.0, 5.0, 7.5, 10.0]
    assert len(get_xv_models({'impulse': {}})) == 1               # setting absent -> the model itself
    popn2 = Population(make_model('standard_glm', N=4, dt=0.001))
    popn2.add_data(data)
    best_x, best_ind, tr, xv, tot = run_xv(popn2, data, models[:4], rng=np.random.RandomState(5), verbose=False)
    assert 0 <= best_ind < 4 and np.all(np.isfinite(xv)) and np.all(np.isfinite(tr))
    assert best_ind == int(np.argmax(xv))
    assert popn2.data_sequences == [data] or popn2.data_sequences[0] is data
    # held-out ll of every fitted model beats a prior draw; the prior scalar really changed
    popn2.set_hyperparameters(models[best_ind])
    assert popn2.glm.imp_model.prior.lam == models[best_ind]['impulse']['prior']['lam']
    x_rand = popn2.sample(np.random.RandomState(6))
    assert popn2.compute_ll(best_x) > popn2.compute_ll(x_rand)
    assert len(popn2._handles) == 1                               # the train / held-out handles were released
    popn2.release_data()
    assert popn2._handles == []


def test_separable_stimulus_matches_dense_and_oracle():
    """pgl_set_stimulus_separable (rank-1 stimulus kept as w_t, w_x on the device: frame-rate GEMM + 1-D
    convolutions, bkgd.py:214-227) against the dense feature path (pgl_set_stimulus) and the oracle:
    identity and non-identity spatial bases, interpolation from a coarse grid with a clamped tail,
    a neuron sub-range and a time range."""
    from tests import helpers as H
    from theano_pyglm_amd import _lib
    rng = np.random.RandomState(77)
    N, nT, D, Bt, Rt = 6, 3000, 40, 3, 300
    ibt = H.golden()['lr2d_ibasis_t']
    assert ibt.shape == (Rt, Bt)
    for Bx, dt_stim, Tstim in ((D, 0.1, 25), (5, 0.013, 231)):       # second: frames end before the recording does
        stim = rng.randn(Tstim, D)
        ibx = None if Bx == D else rng.randn(D, Bx)
        p = H.Problem(N, nT, H.st_ibasis(), kind='exp', seed=5, w_scale=0.02)
        w_t, w_x = 0.3 * rng.randn(N, Bt), 0.3 * rng.randn(N, Bx)
        dense = p.device()
        dense.set_stimulus(stim, dt_stim, ibt, ibx, layout=0)
        th_d = np.concatenate((p.theta[:, :1], np.einsum('nt,nx->ntx', w_t, w_x).reshape(N, -1), p.theta[:, 1:]), axis=1)
        ll_d, g_d = dense.ll_grad(th_d, p.Weff)
        sep = p.device()
        sep.set_stimulus_separable(stim, dt_stim, ibt, ibx)
        assert sep.P == 1 + Bt + Bx + N * 3
        # dt_stim = 100 dt: the frame-rate kernels (5 frame values per bin); dt_stim = 13 dt would need 26: tap-rate kernels
        assert sep.info()['stim_path'] == (2 if dt_stim == 0.1 else 1)
        th_s = np.concatenate((p.theta[:, :1], w_t, w_x, p.theta[:, 1:]), axis=1)
        ll_s, g_s = sep.ll_grad(th_s, p.Weff)
        assert np.allclose(ll_s, ll_d, rtol=1e-11)
        G = g_d[:, 1:1 + Bt * Bx].reshape(N, Bt, Bx)
        g_chain = np.concatenate((g_d[:, :1], np.einsum('ntx,nx->nt', G, w_x), np.einsum('ntx,nt->nx', G, w_t),
                                  g_d[:, 1 + Bt * Bx:]), axis=1)
        assert H.rel_err(g_s, g_chain) < 1e-10
        ll_only, _ = sep.ll_grad(th_s, p.Weff, want_grad=False)
        assert np.allclose(ll_only, ll_s, rtol=1e-13)
        # oracle on the dense features
        fst = O.spatiotemporal_stim_features(stim, dt_stim, 0.001, nT, np.eye(D) if ibx is None else ibx, ibt)
        p.fstim, p.Dstim, p.P, p.theta = fst, Bt * Bx, th_d.shape[1], th_d
        ll0, g0 = p.oracle_ll_grad()
        assert np.allclose(ll_s, ll0, rtol=1e-10)
        # neuron sub-range and time range (partial sums add up)
        a, b = sep.ll_grad(th_s[2:5], p.Weff, 2, 5)
        assert np.allclose(a, ll_s[2:5], rtol=1e-12) and H.rel_err(b, g_s[2:5]) < 1e-11
        acc_l, acc_g = 0.0, 0.0
        for lo, hi in ((0, 1008), (1008, 3000)):
            sep.set_time_range(lo, hi)
            a, b = sep.ll_grad(th_s, p.Weff)
            acc_l, acc_g = acc_l + a, acc_g + b
        assert np.allclose(acc_l, ll_s, rtol=1e-11) and H.rel_err(acc_g, g_s) < 1e-10
        # state read-back (I_stim by the separable path)
        sep.set_time_range(0, nT)
        lam, inet, istim = sep.state(1, th_s[1], p.Weff[:, 1])
        assert np.max(np.abs(istim - fst.dot(th_d[1, 1:1 + Bt * Bx]))) < 1e-10
        with pytest.raises(_lib.PglError):
            sep.get_stim_features()
        dense.close()
        sep.close()


def test_separable_stimulus_frame_rate_kernels():
    """The frame-rate form of the separable stimulus (k_sepf_fwd / k_sepf_bwd / k_sepf_finish + the MFMA GEMMs, impulse
    columns on resident tiles through the slab-input forms of k_fused7 / k_fused5) against the tap-rate kernels of the same handle
    (dev option 94) and against the oracle on the dense features (bkgd.py:214-227, 303-340; basis.py:238-273): one to
    four post tiles, q = 100 / 50 / 150 bins per frame (5, 8 and 4 frame values per bin), one to three temporal bases,
    recordings that end before / behind the last frame, time ranges that cut frames."""
    from tests import helpers as H
    from theano_pyglm_amd import _lib
    rng = np.random.RandomState(78)
    ibt3 = H.golden()['lr2d_ibasis_t']
    cases = ((6, 3000, 24, 0.1, 25, 3, None),        # frames end at bin 2400: clamped tail
             (40, 4000, 32, 0.05, 90, 2, 7),         # q = 50 -> 8 frame values; non-identity spatial basis; frames outlast the bins
             (64, 2512, 16, 0.15, 17, 1, None),      # q = 150 -> 4 frame values, recording not a multiple of the frame
             (20, 1600, 8, 0.1, 16, 3, None),
             (100, 1808, 12, 0.1, 18, 3, None),      # seven post tiles: the two-pass kernel with the slab-input pass 1
             (128, 1600, 6, 0.05, 40, 2, 4))
    for N, nT, D, dt_stim, Tstim, Bt, Bx_ in cases:
        Bx = D if Bx_ is None else Bx_
        ibt = np.ascontiguousarray(ibt3[:, :Bt])
        stim = rng.randn(Tstim, D)
        ibx = None if Bx_ is None else rng.randn(D, Bx)
        p = H.Problem(N, nT, H.st_ibasis(), kind='exp', seed=5 + N, w_scale=0.02)
        w_t, w_x = 0.3 * rng.randn(N, Bt), 0.3 * rng.randn(N, Bx) / np.sqrt(Bx / 8.0)
        sep = p.device()
        sep.set_stimulus_separable(stim, dt_stim, ibt, ibx)
        th_s = np.concatenate((p.theta[:, :1], w_t, w_x, p.theta[:, 1:]), axis=1)
        assert sep.info()['stim_path'] == 2 and sep.info()['kernel_version'] == (7 if N <= 64 else 5)
        ll_f, g_f = sep.ll_grad(th_s, p.Weff)
        ll_only, _ = sep.ll_grad(th_s, p.Weff, want_grad=False)
        assert np.allclose(ll_only, ll_f, rtol=1e-13)
        # oracle on the dense features
        fst = O.spatiotemporal_stim_features(stim, dt_stim, 0.001, nT, np.eye(D) if ibx is None else ibx, ibt)
        th_d = np.concatenate((p.theta[:, :1], np.einsum('nt,nx->ntx', w_t, w_x).reshape(N, -1), p.theta[:, 1:]), axis=1)
        p.fstim, p.Dstim, p.P, p.theta = fst, Bt * Bx, th_d.shape[1], th_d
        ll0, g0 = p.oracle_ll_grad()
        G = g0[:, 1:1 + Bt * Bx].reshape(N, Bt, Bx)
        g_chain = np.concatenate((g0[:, :1], np.einsum('ntx,nx->nt', G, w_x), np.einsum('ntx,nt->nx', G, w_t),
                                  g0[:, 1 + Bt * Bx:]), axis=1)
        assert np.allclose(ll_f, ll0, rtol=1e-10), (N, np.max(np.abs(ll_f - ll0) / np.abs(ll0)))
        for sl in (slice(0, 1), slice(1, 1 + Bt), slice(1 + Bt, 1 + Bt + Bx), slice(1 + Bt + Bx, None)):
            assert H.rel_err(g_f[:, sl], g_chain[:, sl]) < 1e-9, (N, sl)
        # neuron sub-range and time ranges that cut stimulus frames: partial sums add up
        lo_n, hi_n = N // 3, N // 3 + max(1, N // 2)
        a, b = sep.ll_grad(th_s[lo_n:hi_n], p.Weff, lo_n, hi_n)
        assert np.allclose(a, ll_f[lo_n:hi_n], rtol=1e-12) and H.rel_err(b, g_f[lo_n:hi_n]) < 1e-11
        acc_l, acc_g = 0.0, 0.0
        cut1, cut2 = 16 * 21, 16 * 77
        for lo, hi in ((0, cut1), (cut1, cut2), (cut2, nT)):
            sep.set_time_range(lo, hi)
            a, b = sep.ll_grad(th_s, p.Weff)
            acc_l, acc_g = acc_l + a, acc_g + b
        assert np.allclose(acc_l, ll_f, rtol=1e-11) and H.rel_err(acc_g, g_f) < 1e-10
        if N in (40, 100):
            # a neuron LIST (pgl_ll_grad_list_dev) through the frame-rate path == the rows of the range call
            import torch
            idx = np.array([N - 1, 3, N // 2, 17, 0] + list(range(20, 20 + (70 if N == 100 else 10))), dtype=np.int32)
            d_idx = torch.from_numpy(idx).cuda()
            d_th = torch.from_numpy(np.ascontiguousarray(th_s[idx])).cuda()
            d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
            d_ll = torch.zeros(len(idx), dtype=torch.float64, device='cuda')
            d_g = torch.zeros((len(idx), th_s.shape[1]), dtype=torch.float64, device='cuda')
            torch.cuda.synchronize()
            sep.set_time_range(0, nT)
            sep.ll_grad_list_dev(d_idx.data_ptr(), len(idx), d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
            sep.sync()
            assert np.allclose(d_ll.cpu().numpy(), ll_f[idx], rtol=1e-12), N
            assert np.max(np.abs(d_g.cpu().numpy() - g_f[idx])) < 1e-10 * np.max(np.abs(g_f)), N
        sep.set_time_range(0, nT - 37)
        a37, b37 = sep.ll_grad(th_s, p.Weff)
        # the stimulus current through the slab (option 94 = 3) instead of inside the forward contraction (<= 64 neurons)
        sep.set_option(94, 3)
        a37b, b37b = sep.ll_grad(th_s, p.Weff)
        assert np.allclose(a37, a37b, rtol=1e-12) and H.rel_err(b37, b37b) < 1e-11
        # the tap-rate kernels on the same handle
        sep.set_option(94, 2)
        assert sep.info()['stim_path'] == 1
        a37t, b37t = sep.ll_grad(th_s, p.Weff)
        assert np.allclose(a37, a37t, rtol=1e-12) and H.rel_err(b37, b37t) < 1e-11
        sep.set_time_range(0, nT)
        ll_t, g_t = sep.ll_grad(th_s, p.Weff)
        assert np.allclose(ll_f, ll_t, rtol=1e-12) and H.rel_err(g_f, g_t) < 1e-11
        sep.close()


def test_two_data_sequences_log_p_and_lockstep_map():
    """population.py:80-86 sums the likelihood over every data sequence (add_data twice).  log p is the oracle's
    prior + sum over both sequences; the lock-step optimizer (one device handle per sequence, both on one
    stream) reaches the optimum of the sequential reference-style fit on the same two sequences."""
    model, popn, data1 = make_dataset('standard_glm', 4, 5.0, seed=31)
    _, _, data2 = make_dataset('standard_glm', 4, 3.0, seed=32)
    popn.add_data(data2)
    assert len(popn.data_sequences) == 2
    x0 = popn.sample(np.random.RandomState(33))
    lp = popn.compute_log_p(x0)
    lp1, ll1 = oracle_log_p(popn, data1, x0)
    lp2, ll2 = oracle_log_p(popn, data2, x0)
    assert np.isclose(lp, lp1 + np.sum(ll2), rtol=1e-10)          # the prior counts once
    x_seq = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched=False)
    x_t = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    lp_seq, lp_t = popn.compute_log_p(x_seq), popn.compute_log_p(x_t)
    assert lp_t > lp + 1.0
    assert abs(lp_t - lp_seq) < 1e-2 * max(1.0, abs(lp_seq) * 1e-3)
    st = popn.last_fit_stats
    assert st['converged_gtol'] + st['stalled'] + st['maxiter'] == 4
    popn.release_data()


def test_lockstep_row_kernels_follow_scipy_bfgs(std4):
    """The lock-step optimizer (HIP row kernels: pgl_bfgs_init / trial / objective / linesearch / hmul / update_dev) runs the
    algorithm scipy runs for the reference's fit_glm (coord_descent.py:161-204): BFGS from H = I with the More'-Thuente
    strong-Wolfe search and scipy's first trial step.  Per neuron: the same number of iterations and line-search steps
    as the sequential scipy fit of that neuron (rounding may move a stop by one iteration), objectives equal to 1e-9 --
    for the group-lasso prior (standard_glm), a Gaussian impulse prior, a BasisStimulus model (stimulus block of the prior
    kernel) and a neuron sub-range; a NaN gradient (zero impulse group under the group lasso, priors.py:202) takes
    fit_glm's path (zero gradient: the fit stops where it started)."""
    from theano_pyglm_amd.inference import coord_descent as cd
    from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
    model, popn, data = std4
    cases = [(popn, data, 0, 4), (popn, data, 1, 3)]
    mg = make_model('standard_glm', N=4, dt=0.001)
    mg['impulse']['prior'] = {'type': 'gaussian', 'mu': 0.0, 'sigma': 2.0}
    pg = Population(mg)
    pg.add_data(dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed')))
    cases.append((pg, data, 0, 4))
    mb = make_model('standard_glm', N=4, dt=0.001)
    mb['bkgd'] = {'type': 'basis', 'D_stim': 2, 'dt_max': 0.3, 'basis': mb['bkgd']['basis']}
    pb = Population(mb)
    dstim = dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed'))
    dstim['stim'] = np.random.RandomState(8).randn(int(round(data['T'] / 0.1)), 2)
    dstim['dt_stim'] = 0.1
    pb.add_data(dstim)
    cases.append((pb, dstim, 0, 4))
    same_nit = total = 0
    for k, (pp, dd, lo, hi) in enumerate(cases):
        x0 = pp.sample(np.random.RandomState(40 + k))
        if k == 0:
            x0['glms'][2]['imp']['w_ir'][5:10] = 0.0          # zero group: NaN prior gradient at the start
        xa = copy.deepcopy(x0)
        fa, ita, eva = fit_glms_batched_torch(pp, xa, n_lo=lo, n_hi=hi)
        sa = dict(pp.last_fit_stats)
        assert sa['bookkeeping'] == 'hip row kernels'
        assert sa['converged_gtol'] + sa['stalled'] + sa['maxiter'] == hi - lo
        prms = cd.prep_first_order_glm_inference(pp)
        for i, n in enumerate(range(lo, hi)):
            nv = pp.extract_vars(copy.deepcopy(x0), n)
            res = cd.fit_glm(nv, n, prms)
            nit, nls = sa['per_neuron']['iterations'][i], sa['per_neuron']['line_search_steps'][i]
            assert abs(res.fun - fa[i]) <= 1e-9 * abs(res.fun), (k, n, res.fun, fa[i])
            # scipy: nfev = the evaluation at the start + one per line-search step
            assert abs(res.nit - nit) <= 1 and abs((res.nfev - 1) - nls) <= 2, (k, n, res.nit, nit, res.nfev, nls)
            same_nit += int(res.nit == nit and res.nfev - 1 == nls)
            total += 1
            assert np.allclose(pp.glm.theta_row(xa['glms'][n]), pp.glm.theta_row(nv['glm']), rtol=1e-5, atol=1e-7)
        if k == 0:
            assert sa['per_neuron']['iterations'][2] == 0 and np.array_equal(xa['glms'][2]['imp']['w_ir'], x0['glms'][2]['imp']['w_ir'])
    assert same_nit >= total - 3, (same_nit, total)            # almost always the very same count
    pg.release_data()
    pb.release_data()


def test_separable_stimulus_frame_rate_randomised_shapes():
    """Randomised shapes for the frame-rate stimulus kernels against the tap-rate kernels of the same handle: frame
    lengths from 2 to 150 bins, temporal filters of 20 to 300 taps with 1 to 4 bases (3 to 8 frame values per bin, both
    template instantiations), odd and even numbers of spatial bases (both GEMM kernels), stimulus shorter and longer
    than the recording, random neuron ranges and time ranges."""
    from tests import helpers as H
    rng = np.random.RandomState(2024)
    done = fused = 0
    while done < 14:
        Rt = int(rng.choice([20, 47, 100, 233, 300]))
        q = int(rng.randint(max(2, -(-Rt // 6)), 151))
        if -(-Rt // q) + 2 > 8:
            continue
        Bt = int(rng.randint(1, 5))
        N = int(rng.choice([1, 5, 16, 17, 33, 64, 80, 128]))
        nT = int(rng.randint(20, 190)) * 16 + int(rng.randint(0, 16))
        D = int(rng.randint(2, 24))
        Bx = D if rng.rand() < 0.5 else int(rng.randint(1, 12))
        Tstim = max(2, int(nT / q * rng.choice([0.5, 1.0, 1.3])) + int(rng.randint(0, 3)))
        ibt = rng.randn(Rt, Bt) / np.sqrt(Rt)
        ibx = None if Bx == D else rng.randn(D, Bx)
        stim = rng.randn(Tstim, D)
        p = H.Problem(N, nT, H.st_ibasis(), kind='exp' if rng.rand() < 0.5 else 'explinear', seed=100 + done, w_scale=0.02,
                      bias_mu=1.0)
        dev = p.device()
        dev.set_stimulus_separable(stim, q * 0.001, ibt, ibx)
        assert dev.info()['stim_path'] == 2, (Rt, q, Bt)
        th = np.concatenate((p.theta[:, :1], 0.3 * rng.randn(N, Bt), 0.3 * rng.randn(N, Bx) / np.sqrt(Bx), p.theta[:, 1:]), axis=1)
        n_lo = int(rng.randint(0, N))
        n_hi = int(rng.randint(n_lo + 1, N + 1))
        t_lo = 16 * int(rng.randint(0, nT // 32))
        t_hi = int(rng.randint(t_lo + 1, nT + 1))
        dev.set_time_range(t_lo, t_hi)
        ll_f, g_f = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
        case = (Rt, q, Bt, N, nT, D, Bx, Tstim, n_lo, n_hi, t_lo, t_hi)
        # option 94 = 3: the stimulus current through the slab; where the default put it into the forward contraction
        # (<= 64 listed neurons, <= 3 temporal bases, <= 5 frame values, frames of >= 16 bins) the bits differ
        dev.set_option(94, 3)
        ll_b, g_b = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
        assert np.allclose(ll_f, ll_b, rtol=1e-11, atol=1e-12) and H.rel_err(g_f, g_b) < 1e-10, case
        fused += int(not np.array_equal(g_f, g_b))
        dev.set_option(94, 2)
        assert dev.info()['stim_path'] == 1
        ll_t, g_t = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
        assert np.allclose(ll_f, ll_t, rtol=1e-11, atol=1e-12), case
        assert H.rel_err(g_f, g_t) < 1e-10, case
        assert np.all(np.isfinite(ll_f)) and np.any(g_f[:, 1:1 + Bt] != 0.0) and np.any(g_f[:, 1 + Bt:1 + Bt + Bx] != 0.0), case
        print("frame-rate vs tap-rate, case", case, "max rel grad diff %.1e" % H.rel_err(g_f, g_t))
        dev.close()
        done += 1
    assert fused >= 3, fused


def test_lockstep_map_separable_stimulus_row_kernels_and_lists():
    """spatiotemporal_glm with a wide stimulus (separable device path at the frame rate): the lock-step optimizer runs on
    the HIP row kernels with neuron LISTS (pgl_ll_grad_list_dev through the frame-rate stimulus kernels) and ends where
    the sequential scipy fits (neuron ranges) end; a list call equals the range call row by row."""
    from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch, _Packing
    from theano_pyglm_amd.models import templates
    import torch
    tmpl = templates.spatiotemporal_glm()
    tmpl['bkgd']['D_stim'] = 40
    tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': 40}
    tmpl['bkgd']['sigma'] = 0.05

    def tame(x):
        for xn in x['glms']:
            xn['bias']['bias'] = np.array([2.0])

    N = 6
    model, popn_gen, data = make_dataset(tmpl, N, 6.0, seed=43, check=True, adjust=tame)
    popn_gen.release_data()
    # the fitted model: a wider impulse prior (the template's N(0, 0.001) needs more than 225 BFGS iterations)
    tmpl_fit = copy.deepcopy(tmpl)
    tmpl_fit['impulse']['sigma'] = 0.5
    popn = Population(make_model(tmpl_fit, N=N, dt=0.001))
    popn.add_data(dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed')))
    assert popn.glm.bkgd_model.separable
    h = popn._handle(popn._current)
    assert h.info()['stim_path'] == 2
    pk = _Packing(popn, torch, [h])
    assert pk.identity and pk.list_launch
    # a neuron list == the range call, row by row
    x = popn.sample(np.random.RandomState(44))
    tame(x)
    th = popn.theta_matrix(x)
    W = popn.W_eff(x)
    ll_r, g_r = h.ll_grad(th, W)
    idx = np.array([4, 1, 5], dtype=np.int32)
    d_idx = torch.from_numpy(idx).cuda()
    d_th = torch.from_numpy(np.ascontiguousarray(th[idx])).cuda()
    d_W = torch.from_numpy(np.ascontiguousarray(W)).cuda()
    d_ll = torch.zeros(3, dtype=torch.float64, device='cuda')
    d_g = torch.zeros((3, th.shape[1]), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    h.ll_grad_list_dev(d_idx.data_ptr(), 3, d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    h.sync()
    assert np.allclose(d_ll.cpu().numpy(), ll_r[idx], rtol=1e-12)
    assert np.max(np.abs(d_g.cpu().numpy() - g_r[idx])) < 1e-10 * np.max(np.abs(g_r))
    # row kernels with neuron lists against sequential scipy fits through the host-pointer API (neuron ranges)
    from theano_pyglm_amd.inference import coord_descent as cd
    x0 = popn.sample(np.random.RandomState(45))
    tame(x0)
    xa = copy.deepcopy(x0)
    fa, ita, eva = fit_glms_batched_torch(popn, xa)
    sa = dict(popn.last_fit_stats)
    assert sa['bookkeeping'] == 'hip row kernels'
    assert sa['neuron_evaluations'] < sa['evaluations'] * N                # lists: finished neurons drop out of the launches
    assert sa['converged_gtol'] + sa['stalled'] == N
    prms = cd.prep_first_order_glm_inference(popn)
    for n in (0, 3, 5):
        nv = popn.extract_vars(copy.deepcopy(x0), n)
        res = cd.fit_glm(nv, n, prms)
        if res.status == 2 and res.nit <= 2:
            # neuron 3 starts at an objective of 4e21 (exp nonlinearity, prior draw of the stimulus weights): scipy's first
            # line search meets inf, its fallback search fails and the fit stops where it started ("precision loss");
            # the lock-step fit takes the best sufficient-decrease point of the stuck search and carries on to the optimum
            assert n == 3 and res.fun > 1e20 and fa[n] < 0.0 and sa['per_neuron']['iterations'][n] > 20, (n, res.fun, fa[n])
            assert np.max(np.abs(popn.compute_grad(xa, n))) < 1e-3
            continue
        # (~100+ BFGS iterations: list launches sum the listed rows in another order than range launches, the two
        #  trajectories drift apart by rounding and may stop a few iterations apart -- at the same optimum)
        assert abs(res.fun - fa[n]) <= 1e-8 * abs(res.fun), (n, res.fun, fa[n], sa)
        assert abs(res.nit - sa['per_neuron']['iterations'][n]) <= max(3, res.nit // 20), (n, res.nit, sa['per_neuron'])
        assert np.allclose(popn.glm.theta_row(xa['glms'][n]), popn.glm.theta_row(nv['glm']), rtol=1e-3, atol=1e-5)
    popn.release_data()


def test_separable_stimulus_wide_population_short_lists_and_shards():
    """spatiotemporal_glm with N * B > 320 impulse columns (N = 128, B = 3) and a separable stimulus: neuron lists and
    neuron shards SHORTER than 49 neurons stay on the frame-rate path (the feature row is too long for k_fused7, so the
    slab-input form of the two-pass kernel serves any number of post tiles) and equal the whole-population call row by
    row; the lock-step MAP fit of a 28-neuron shard and the default sweep of the whole population (its launch lists shrink
    below 49 as neurons finish) run through it."""
    import torch
    from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch, _Packing
    from theano_pyglm_amd.inference import coord_descent as cd
    from theano_pyglm_amd.models import templates
    N, nT, D = 128, 8000, 24
    tmpl = templates.spatiotemporal_glm()
    tmpl['bkgd']['D_stim'] = D
    tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
    tmpl['bkgd']['sigma'] = 0.05
    tmpl['bkgd']['separable'] = True
    tmpl['impulse']['sigma'] = 0.5
    rng = np.random.default_rng(77)
    S = np.minimum(rng.poisson(20.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
    stim = rng.standard_normal((nT // 100, D))
    popn = Population(make_model(tmpl, N=N, dt=0.001))
    popn.add_data({'S': S, 'N': N, 'dt': 0.001, 'T': nT * 0.001, 'stim': stim, 'dt_stim': 0.1})
    h = popn._handle(popn._current)
    assert h.info()['stim_path'] == 2 and h.info(100, 128)['stim_path'] == 2 and h.info(5, 6)['stim_path'] == 2
    pk = _Packing(popn, torch, [h])
    assert pk.identity and pk.list_launch
    x = popn.sample(np.random.RandomState(3))
    for xn in x['glms']:
        xn['bias']['bias'] = np.array([1.0])
        xn['imp']['w_ir'] = 0.02 * np.asarray(xn['imp']['w_ir']) / 0.5
    th = popn.theta_matrix(x)
    W = popn.W_eff(x)
    ll_r, g_r = h.ll_grad(th, W)
    assert np.all(np.isfinite(ll_r))
    # shards shorter than 49 neurons (1, 2 and 3 post tiles)
    for lo, hi in ((100, 128), (7, 20), (60, 101)):
        ll_s, g_s = h.ll_grad(th[lo:hi], W, lo, hi)
        assert np.allclose(ll_s, ll_r[lo:hi], rtol=1e-11) and np.max(np.abs(g_s - g_r[lo:hi])) < 1e-9 * np.max(np.abs(g_r))
    # neuron lists of 1 .. 48 neurons
    for cnt in (1, 17, 48):
        idx = np.sort(np.random.RandomState(cnt).permutation(N)[:cnt]).astype(np.int32)
        d_idx = torch.from_numpy(idx).cuda()
        d_th = torch.from_numpy(np.ascontiguousarray(th[idx])).cuda()
        d_W = torch.from_numpy(np.ascontiguousarray(W)).cuda()
        d_ll = torch.zeros(cnt, dtype=torch.float64, device='cuda')
        d_g = torch.zeros((cnt, th.shape[1]), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        h.ll_grad_list_dev(d_idx.data_ptr(), cnt, d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
        h.sync()
        assert np.allclose(d_ll.cpu().numpy(), ll_r[idx], rtol=1e-11), cnt
        assert np.max(np.abs(d_g.cpu().numpy() - g_r[idx])) < 1e-9 * np.max(np.abs(g_r)), cnt
    # lock-step fit of a short shard, and the default sweep with shrinking lists
    lp0, _ = popn.compute_lp_grad_packed(x)
    xs = copy.deepcopy(x)
    nlp_s, _, _ = fit_glms_batched_torch(popn, xs, n_lo=100, n_hi=128, maxiter=40)
    assert np.all(nlp_s < -lp0[100:128])
    xm = cd.coord_descent(popn, x0=copy.deepcopy(x), maxiter=1)
    st = popn.last_fit_stats
    assert st['bookkeeping'] == 'hip row kernels' and st['neuron_evaluations'] < st['evaluations'] * N, st
    assert popn.compute_log_p(xm) > popn.compute_log_p(x)
    popn.release_data()


def test_all_f64_epilogue_option():
    """PGL_OPT_EPI_F64 (run-time switch of the single-precision exp(-x) correction): same ll and gradient to 1e-12 in
    the regime where the correction is used (currents > 12), bit-identical outside it."""
    import torch
    from tests import helpers as Hh
    from theano_pyglm_amd import _lib
    for bias_mu, N, nT in ((20.0, 40, 3000), (2.0, 40, 3000), (20.0, 130, 1500)):
        p = Hh.Problem(N, nT, Hh.std_ibasis(), seed=7, bias_mu=bias_mu, w_scale=0.3)
        d = p.device()
        ll0, g0 = d.ll_grad(p.theta, p.Weff)
        d.set_option(_lib.OPT_EPI_F64, 1)
        ll1, g1 = d.ll_grad(p.theta, p.Weff)
        d.set_option(_lib.OPT_EPI_F64, 0)
        ll2, g2 = d.ll_grad(p.theta, p.Weff)
        assert np.array_equal(ll0, ll2) and np.array_equal(g0, g2)
        assert np.allclose(ll0, ll1, rtol=1e-12, atol=0) and Hh.rel_err(g0, g1) < 1e-12
        if bias_mu < 12:
            assert np.array_equal(ll0, ll1) and np.array_equal(g0, g1)
        ll_or, g_or = p.oracle_ll_grad(0, 3)
        assert np.allclose(ll1[:3], ll_or, rtol=1e-10) and Hh.rel_err(g1[:3], g_or) < 1e-9
        d.close()


def test_bfgs_hmul_kernel_against_dense_algebra():
    """pgl_bfgs_hmul_dev -- the one pass over the dense inverse Hessians per accepted iteration: for the listed rows with
    acc = 1 it applies the pending rank-3 update H <- H + U V^T and returns t = H g; a row still at hscale * I is
    materialised together with its first update and left alone without one; rows without acc (or not listed) are not
    touched.  Odd and even P (padded leading dimension), P below and above one column sweep of a wave."""
    import torch
    from tests import helpers as Hh
    p = Hh.Problem(3, 200, Hh.std_ibasis(), seed=1)
    d = p.device()
    gen = torch.Generator(device='cuda').manual_seed(5)
    for M, P in ((6, 37), (5, 130), (4, 641)):
        ld = P + (P & 1)
        n = d.bfgs_state_doubles(M, P)
        st = torch.zeros(n, dtype=torch.float64, device='cuda')
        MP = M * P
        g = st[MP:2 * MP].view(M, P)
        t = st[6 * MP:7 * MP].view(M, P)
        U, V = st[9 * MP:12 * MP].view(M, P, 3), st[12 * MP:15 * MP].view(M, P, 3)
        sc = st[15 * MP:].view(-1, M)
        hscale, acc, ident, pend = sc[5], sc[10], sc[13], sc[14]
        g.copy_(torch.randn(M, P, dtype=torch.float64, device='cuda', generator=gen))
        U.copy_(torch.randn(M, P, 3, dtype=torch.float64, device='cuda', generator=gen))
        V.copy_(torch.randn(M, P, 3, dtype=torch.float64, device='cuda', generator=gen))
        t.fill_(-7.0)
        H = torch.randn(M, P, ld, dtype=torch.float64, device='cuda', generator=gen)
        H0 = H.clone()
        # rows: 0 dense + pending, 1 dense no pending, 2 identity + pending, 3 identity no pending (lazy), 4.. acc = 0
        acc.copy_(torch.tensor([1, 1, 1, 1] + [0] * (M - 4), dtype=torch.float64))
        ident.copy_(torch.tensor([0, 0, 1, 1] + [0] * (M - 4), dtype=torch.float64))
        pend.copy_(torch.tensor([1, 0, 1, 0] + [1] * (M - 4), dtype=torch.float64))
        hscale.copy_(torch.tensor([1.0, 1.0, 0.37, 2.0] + [1.0] * (M - 4), dtype=torch.float64))
        rows = torch.tensor([3, 0, 2, 1] + list(range(4, M)), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        d.bfgs_hmul_dev(st.data_ptr(), M, P, rows.data_ptr(), M, H.data_ptr(), ld)
        d.sync()
        eye = torch.eye(P, dtype=torch.float64, device='cuda')
        UV = torch.bmm(U, V.transpose(1, 2))
        want0 = H0[0, :, :P] + UV[0]
        assert torch.allclose(H[0, :, :P], want0, rtol=1e-13, atol=1e-13)
        assert torch.allclose(t[0], want0 @ g[0], rtol=1e-11, atol=1e-11)
        assert torch.equal(H[1], H0[1]) and torch.allclose(t[1], H0[1, :, :P] @ g[1], rtol=1e-11, atol=1e-11)
        want2 = 0.37 * eye + UV[2]
        assert torch.allclose(H[2, :, :P], want2, rtol=1e-13, atol=1e-13)
        assert torch.allclose(t[2], want2 @ g[2], rtol=1e-11, atol=1e-11)
        assert torch.equal(H[3], H0[3]) and bool((t[3] == -7.0).all())          # lazy identity: not touched
        for m in range(4, M):
            assert torch.equal(H[m], H0[m]) and bool((t[m] == -7.0).all())
    d.close()


def test_sta_factorisation_on_the_gpu_matches_lapack_svd():
    """smart_init.leading_singular_pairs (pgl_leading_singular_pairs: batched Gram matrices, repeated squaring, two alternating
    steps -- the library's own kernels) against
    np.linalg.svd -- what the reference calls per neuron (smart_init.py:68-72): leading singular value to 1e-12, the vectors
    up to the pair's common sign, for wide (300 x 1024 STA), tall and noise-only matrices; same values as the host routine."""
    from theano_pyglm_amd.inference.smart_init import leading_singular_pairs, leading_singular_pair
    rng = np.random.default_rng(11)
    for shape, planted in (((300, 1024), 2.0), ((300, 1024), 0.0), ((40, 7), 1.0), ((12, 300), 0.3)):
        S = rng.standard_normal((6,) + shape)
        for i in range(6):
            S[i] += planted * np.outer(rng.standard_normal(shape[0]), rng.standard_normal(shape[1]))
        U, Sig, V = leading_singular_pairs(S)
        for i in range(6):
            u0, s0, v0 = np.linalg.svd(S[i], full_matrices=False)
            sg = np.sign(u0[:, 0].dot(U[i]))
            gap = s0[0] / s0[1]
            tol = 1e-9 / max(gap - 1.0, 1e-3)            # (the vectors of a nearly degenerate pair are ill-conditioned)
            assert abs(Sig[i] - s0[0]) <= 1e-12 * s0[0]
            assert np.max(np.abs(U[i] - sg * u0[:, 0])) < tol and np.max(np.abs(V[i] - sg * v0[0])) < tol
            assert U[i][np.argmax(np.abs(U[i]))] > 0
            uh, sh, vh = leading_singular_pair(S[i])
            assert abs(sh - Sig[i]) <= 1e-12 * sh and np.max(np.abs(uh - U[i])) < tol and np.max(np.abs(vh - V[i])) < tol


def test_sta_factors_without_leaving_the_device():
    """initialize_stim_with_sta on a wide stimulus: the spike-triggered averages stay on the device (pgl_sta with A_out = NULL)
    and only their leading singular pairs come back (pgl_leading_singular_pairs with A = NULL) -- the same pairs as from the
    averages copied out; the kept buffer is consumed by the call (a second call without a new pgl_sta fails loudly)."""
    from theano_pyglm_amd import _lib
    rng = np.random.default_rng(5)
    N, nT, D, L = 6, 40000, 96, 50
    S = np.minimum(rng.poisson(30.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
    stim = rng.standard_normal((nT // 100, D))
    d = _lib.DeviceGlm(N, nT, 1, 1, 'exp', 0.001)
    d.set_spikes(S)
    A = d.sta(stim, 0.1, L)
    U0, S0, V0 = d.leading_singular_pairs(A)
    shape = d.sta(stim, 0.1, L, keep_on_device=True)
    assert shape == A.shape
    U1, S1, V1 = d.leading_singular_pairs(None, shape)
    assert np.array_equal(U0, U1) and np.array_equal(S0, S1) and np.array_equal(V0, V1)
    for i in range(N):
        u, s, vt = np.linalg.svd(A[i], full_matrices=False)
        assert abs(S1[i] - s[0]) <= 1e-12 * s[0]
    with pytest.raises(_lib.PglError):
        d.leading_singular_pairs(None, shape)
    d.close()
