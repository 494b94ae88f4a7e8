"""
CPU tests that pin the oracle:
  * against golden vectors produced by the reference's own pyglm/utils/basis.py,
  * against the reference's invariants (rate consistency, one-bin causal shift),
  * against independent derivatives (torch float64 autograd, central differences),
  * C restatement (oracle/glm_oracle.c) against the numpy restatement.
"""
import numpy as np
import pytest

from oracle import glm_oracle as O
from tests import helpers as H


def test_bases_match_reference_golden(golden):
    std = {'type': 'cosine', 'n_eye': 0, 'n_cos': 5, 'a': 1.0 / 120, 'b': 0.5, 'orth': True, 'norm': False}
    assert np.allclose(O.create_basis(std), golden['std_imp_basis'], atol=1e-13)
    assert np.allclose(O.create_basis(dict(std, orth=False)), golden['cos5_raw_basis'], atol=1e-15)
    assert np.allclose(O.create_basis(dict(std, orth=False, norm=True)), golden['swm_imp_basis'], atol=1e-13)
    st = dict(std, n_cos=3, orth=False, norm=True)
    assert np.allclose(O.create_basis(st), golden['st_imp_basis'], atol=1e-13)
    assert np.allclose(O.create_basis(st), golden['st_temporal_basis'], atol=1e-13)
    assert np.array_equal(O.create_basis({'type': 'identity', 'n_eye': 3}), golden['st_spatial_basis'])
    assert np.allclose(O.create_basis(dict(std, n_cos=3)), golden['std_bkgd_basis'], atol=1e-13)


def test_convolve_with_basis_golden(golden):
    for ks, kf in (('conv_S', 'conv_fS'), ('conv_short_S', 'conv_short_fS'), ('conv_one_S', 'conv_one_fS')):
        f = O.convolve_with_basis(golden[ks], golden['conv_ibasis'])
        assert np.max(np.abs(f - golden[kf])) < 1e-12
        f2 = O.convolve_with_basis_fft(golden[ks], golden['conv_ibasis'])
        assert np.max(np.abs(f2 - golden[kf])) < 1e-12
    assert np.allclose(golden['conv_ibasis'], O.interp_basis_unit(golden['std_imp_basis'], 200))


def test_low_rank_2d_golden(golden):
    f = O.convolve_with_low_rank_2d_basis(golden['lr2d_stim'], golden['lr2d_ibasis_x'], golden['lr2d_ibasis_t'])
    assert np.max(np.abs(f - golden['lr2d_fstim'])) < 1e-12
    ibt = O.stim_temporal_ibasis(golden['st_temporal_basis'], 0.001, 0.3, True)
    assert np.allclose(ibt, golden['lr2d_ibasis_t'])


def test_one_bin_causal_shift(golden):
    """basis.py:438-451: filtering with a single unit tap delays the signal by one bin."""
    st, f = golden['shift_stim'], golden['shift_fstim']
    assert np.allclose(st[:-1, 0], f[1:, 0, 0]) and abs(f[0, 0, 0]) < 1e-15   # FFT round-off
    assert np.allclose(O.convolve_with_basis(st, np.array([[1.0]])), f)


def test_rate_consistency_invariant():
    """generate_synth_data.py:125-129: feature-path current == time-domain superposition
    of impulse responses (population.py:351-353)."""
    p = H.Problem(5, 1500, H.std_ibasis(), seed=40, weighted=True)
    n = 3
    w = p.theta[n, 1:].reshape(p.N, p.B)
    x_feat, _, _ = O.glm_currents(n, p.fS, w, p.Weff[:, n], p.theta[n, 0])
    x_dir = O.direct_currents(p.S.astype(float), O.impulse_responses(p.ibasis, w), p.Weff[:, n], p.theta[n, 0])
    assert np.max(np.abs(x_feat - x_dir)) < 1e-11
    assert np.allclose(O.nlin(x_feat, 'explinear'), O.nlin(x_dir, 'explinear'))


@pytest.mark.parametrize('kind,Dstim', [('explinear', 0), ('exp', 4)])
def test_gradient_vs_torch_autograd(kind, Dstim):
    torch = pytest.importorskip('torch')
    ib = H.std_ibasis() if kind == 'explinear' else H.st_ibasis()
    p = H.Problem(4, 900, ib, kind=kind, Dstim=Dstim, seed=41, weighted=True)
    n = 1
    ll0, g0 = p.oracle_ll_grad(n, n + 1)
    th = torch.tensor(p.theta[n], dtype=torch.float64, requires_grad=True)
    fS = torch.tensor(p.fS)
    Sn = torch.tensor(p.S[:, n].astype(float))
    w = th[1 + Dstim:].reshape(p.N, p.B)
    x = th[0] + ((fS * w[None]).sum(2) @ torch.tensor(p.Weff[:, n]))
    if Dstim:
        x = x + torch.tensor(p.fstim) @ th[1:1 + Dstim]
    if kind == 'exp':
        lam = torch.exp(x)
    else:
        lam = torch.nn.functional.softplus(x)
    ll = torch.sum(-p.dt * lam + torch.log(lam) * Sn)
    ll.backward()
    assert np.allclose(ll.item(), ll0[0], rtol=1e-12)
    assert np.max(np.abs(th.grad.numpy() - g0[0])) < 1e-9 * max(1.0, np.max(np.abs(g0)))


def test_gradient_vs_finite_differences():
    p = H.Problem(3, 600, H.std_ibasis(), seed=42)
    n = 0
    _, g0 = p.oracle_ll_grad(n, n + 1)
    eps = 1e-6
    for i in (0, 1, 7, 15):
        tp, tm = p.theta.copy(), p.theta.copy()
        tp[n, i] += eps
        tm[n, i] -= eps
        pp, pm = H.Problem(3, 600, H.std_ibasis(), seed=42), H.Problem(3, 600, H.std_ibasis(), seed=42)
        pp.theta, pm.theta = tp, tm
        fd = (pp.oracle_ll_grad(n, n + 1)[0][0] - pm.oracle_ll_grad(n, n + 1)[0][0]) / (2 * eps)
        assert abs(fd - g0[0, i]) < 1e-5 * max(1.0, abs(fd))


def test_chain_rules():
    rng = np.random.default_rng(3)
    # spatiotemporal: w_stim = vec(w_t (x) w_x)  (bkgd.py:214-220)
    w_t, w_x, g = rng.standard_normal(3), rng.standard_normal(4), rng.standard_normal(12)
    gt, gx = O.spatiotemporal_chain(g, w_t, w_x)
    f = lambda a, b: np.dot(g, O.spatiotemporal_w_stim(a, b))
    e = 1e-6
    assert np.allclose(gt, [(f(w_t + e * np.eye(3)[i], w_x) - f(w_t - e * np.eye(3)[i], w_x)) / (2 * e) for i in range(3)], atol=1e-8)
    assert np.allclose(gx, [(f(w_t, w_x + e * np.eye(4)[i]) - f(w_t, w_x - e * np.eye(4)[i])) / (2 * e) for i in range(4)], atol=1e-8)
    # Dirichlet: beta = |g|/sum|g| (impulse.py:286-291), including a negative entry
    gg = np.array([0.7, -1.3, 0.2, 2.1, 0.9])
    gb = rng.standard_normal(5)
    dg = O.dirichlet_chain(gb, gg)
    fb = lambda v: np.dot(gb, O.dirichlet_beta(v))
    assert np.allclose(dg, [(fb(gg + e * np.eye(5)[i]) - fb(gg - e * np.eye(5)[i])) / (2 * e) for i in range(5)], atol=1e-8)


def test_priors_closed_form():
    v = np.array([[3.0, 4.0], [0.0, -2.0]])
    assert np.isclose(O.group_lasso_log_p(v, 1.0, 0.0, 10.0), -(0.5 + 0.2))      # priors.py:202
    assert np.isclose(O.gaussian_log_p(v, 1.0, 2.0), -0.125 * np.sum((v - 1) ** 2))   # priors.py:139
    assert np.isclose(O.bias_log_p(20.3, 20, 0.1), -0.5 / 0.01 * 0.09)           # bias.py:33
    assert np.isclose(O.basis_stim_log_p(np.array([0.01, -0.02])), -0.5 * (1 + 4))
    A = np.array([[1, 0], [1, 1]])
    rho = np.array([[1.0, 0.3], [0.3, 1.0]])
    lp = O.erdos_renyi_log_p(A, rho)                                             # graph.py:68-71
    assert np.isclose(lp, 2 * np.log(1 - 1e-8) + np.log(0.7) + np.log(0.3))
    g = O.group_lasso_grad(np.zeros((1, 3)), 1.0, 0.0, 10.0)
    assert np.all(np.isnan(g))                                                    # 0/0 like T.grad


def test_mcmc_inner_ll_matches_full_ll():
    """gibbs.py:914: I_net = I_other + w*I_imp[:,n_pre] reproduces the full glm.ll when w is
    the current weight."""
    p = H.Problem(4, 800, H.std_ibasis(), seed=43, weighted=True)
    n_post, n_pre = 1, 2
    w = p.theta[n_post, 1:].reshape(p.N, p.B)
    I_imp = O.impulse_currents(p.fS, w)
    A = (p.Weff != 0).astype(float)
    I_other = O.other_current(I_imp, A, p.Weff, n_pre, n_post)
    ll = O.mcmc_inner_ll([p.Weff[n_pre, n_post]], p.theta[n_post, 0], 0.0, I_other, I_imp[:, n_pre],
                         p.S[:, n_post].astype(float), p.dt, p.kind)
    assert np.isclose(ll[0], p.oracle_ll_grad(n_post, n_post + 1)[0][0], rtol=1e-12)
    ws, om = O.gauss_hermite_nodes(0.0, 1.0)
    assert len(ws) == 10 and np.isclose(np.sum(om), np.sqrt(np.pi))


def test_c_oracle_matches_numpy_oracle():
    from oracle import c_oracle as CO
    p = H.Problem(6, 1200, H.std_ibasis(), seed=44, weighted=True)
    f = CO.features(p.S, p.ibasis)
    assert np.max(np.abs(f - p.fS)) < 1e-12
    ll, g = CO.ll_grad(p.S, f, p.theta, p.Weff, p.kind, p.dt, threads=2)
    ll0, g0 = p.oracle_ll_grad()
    assert np.allclose(ll, ll0, rtol=1e-12) and H.rel_err(g, g0) < 1e-12
    q = H.Problem(5, 700, H.st_ibasis(), kind='exp', Dstim=3, seed=45)
    f = CO.features(q.S, q.ibasis)
    ll, g = CO.ll_grad(q.S, f, q.theta, q.Weff, q.kind, q.dt, fstim=q.fstim)
    ll0, g0 = q.oracle_ll_grad()
    assert np.allclose(ll, ll0, rtol=1e-11) and H.rel_err(g, g0) < 1e-11


def test_packing_order():
    """packvec.py:17-45: sorted-key DFS -> [bias, bkgd.w_t, bkgd.w_x, imp.w_ir]."""
    d = {'imp': {'w_ir': np.arange(4.0)}, 'bias': {'bias': np.array([9.0])},
         'bkgd': {'w_x': np.array([7.0, 8.0]), 'w_t': np.array([5.0, 6.0])}, 'nlin': {}, 'n': {}}
    v, shapes = O.packdict(d)
    assert np.array_equal(v, [9, 5, 6, 7, 8, 0, 1, 2, 3])


def test_project_onto_basis_golden(golden):
    """oracle and host mirror of project_onto_basis against the reference's own output."""
    from theano_pyglm_amd.utils import basis as hb
    f, ibt = golden['proj_f'], golden['lr2d_ibasis_t']
    for lam, key in ((0, 'proj_beta'), (0.5, 'proj_beta_ridge')):
        assert np.allclose(O.project_onto_basis(f, ibt, lam), golden[key], rtol=1e-10, atol=1e-12)
        assert np.allclose(hb.project_onto_basis(f, ibt, lam), golden[key], rtol=1e-10, atol=1e-12)


def test_sta_oracle_definition():
    """the dense-lag-matrix STA (sta.py:56-80) equals the per-spike definition: the average of the
    stimulus windows preceding each spike; and STA-initialised weights recover a planted filter."""
    rng = np.random.RandomState(4)
    nT, D, L = 3000, 2, 40
    stim = rng.randn(nT, D)
    S = (rng.rand(nT, 3) < 0.05).astype(float)
    S[7, 0] = 3.0
    A = O.sta(stim, S, 0.001, 0.001, L, [0, 2])
    for i, n in enumerate([0, 2]):
        acc = np.zeros((L, D))
        for t in np.nonzero(S[:, n])[0]:
            for l in range(min(L, t + 1)):
                acc[l] += S[t, n] * stim[t - l]
        assert np.allclose(A[i], acc / S[:, n].sum(), rtol=1e-12, atol=1e-14)
    # silent neuron -> NaN (0/0) as in the reference
    S[:, 1] = 0
    assert np.all(np.isnan(O.sta(stim, S, 0.001, 0.001, L, [1])))
    # rank-1 planted STA is recovered by the SVD + projection (smart_init.py:66-84)
    ibt = np.linalg.qr(rng.randn(L, 3))[0]
    ibx = np.eye(D)
    wt, wx = np.array([1.0, -0.5, 0.25]), np.array([0.7, -1.2])
    sn = np.outer(ibt.dot(wt), ibx.dot(wx))
    w = O.sta_stim_weights(sn, 'spatiotemporal', ibt, ibx)
    assert np.allclose(np.outer(w['w_t'], w['w_x']), np.outer(wt, wx), atol=1e-12)
    wb = O.sta_stim_weights(sn, 'basis', ibt)
    assert np.allclose(wb['w_stim'].reshape(D, 3), np.outer(wx, wt), atol=1e-12)


def test_frame_rate_form_of_the_stimulus_filter_equals_the_reference_convolution():
    """The piecewise-linear identity the device's frame-rate stimulus kernels are built on (oracle.frame_rate_table)
    against the reference-pinned convolution: interpolate-then-filter (bkgd.py:303-340, basis.py:238-273; the golden
    `lr2d_*` vectors pin convolve_with_low_rank_2d_basis itself) for frames of 100, 50 and 150 bins, recordings that
    end before / behind the last frame, and the first Rt bins where early taps are dropped."""
    g = H.golden()
    ibt = g['lr2d_ibasis_t']
    rng = np.random.RandomState(3)
    for q, Tz, nT in ((100, 9, 1000), (50, 30, 1400), (150, 5, 700)):
        z = rng.randn(Tz, 4)
        s = O.interp_stim(z, q * 0.001, 0.001, nT)
        ref = O.convolve_with_basis(s, ibt)                 # (nT, 4, Bt)
        got = O.frame_rate_features(z, ibt, q, nT)
        assert np.max(np.abs(got - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))
    # the golden vectors of the reference's own low-rank convolution, through the same identity: its stimulus IS the
    # bin-rate series, i.e. frames of one bin
    stim, bx = g['lr2d_stim'], g['lr2d_ibasis_x']
    got = O.frame_rate_features(stim.dot(bx), ibt, 1, stim.shape[0])         # (T, Bx, Bt)
    assert got.shape == g['lr2d_fstim'].shape and np.allclose(got, g['lr2d_fstim'], atol=1e-12)


def test_blocked_c_baseline_matches_reference_dataflow():
    """oracle/glm_blocked.c (B2, all neurons per time tile) == oracle/glm_oracle.c (B1, per neuron)."""
    from oracle import c_oracle as CO
    from tests import helpers as H
    for kind, D, N in (('explinear', 0, 20), ('exp', 3, 7)):
        p = H.Problem(N, 1501, H.std_ibasis(), kind=kind, seed=1, weighted=True, Dstim=D)
        fS = CO.features(p.S, p.ibasis)
        ll0, g0 = CO.ll_grad(p.S, fS, p.theta, p.Weff, kind, p.dt, fstim=p.fstim)
        for th in (1, 3):
            ll, g = CO.ll_grad_blocked(p.S, fS, p.theta, p.Weff, kind, p.dt, fstim=p.fstim, threads=th)
            assert np.allclose(ll, ll0, rtol=1e-12) and H.rel_err(g, g0) < 1e-12
        ll, _ = CO.ll_grad_blocked(p.S, fS, p.theta, p.Weff, kind, p.dt, fstim=p.fstim, want_grad=False)
        assert np.allclose(ll, ll0, rtol=1e-12)
    assert CO.set_data_copy_seconds(fS, p.S.astype(float)) > 0


def test_softplus_tail_table_of_the_gibbs_kernel_matches_60_digit_values():
    """The {L0, s} rows compiled into csrc/pglm_gibbs.hip.h (PGL_SPT, read by k_gibbs_rate_cols for the band
    |x| < 12 of the softplus, glm.py:43-52) are the correctly rounded 60-digit values, and the table algorithm
    (tools/ubench/softplus_tail_table.py, the device code's prototype) reproduces log1p(exp(-a)) to 2e-16 absolute /
    5e-16 relative over the whole band."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('spt', os.path.join(root, 'tools', 'ubench', 'softplus_tail_table.py'))
    spt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(spt)
    src = open(os.path.join(root, 'theano_pyglm_amd', 'csrc', 'pglm_gibbs.hip.h')).read()
    body = src[src.index('PGL_SPT[97][2] = {'):]
    body = body[:body.index('};')]
    rows = re.findall(r'\{(0x[0-9a-f.]+p[+-]?\d+), (0x[0-9a-f.]+p[+-]?\d+)\}', body)
    assert len(rows) == 97
    T = np.array([[float.fromhex(a), float.fromhex(b)] for a, b in rows])
    assert np.array_equal(T, spt.table())
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(0, 12.0000005, 4000), np.arange(97) / 8.0, np.arange(96) / 8.0 + 0.0625])
    ref = spt.reference(a)
    err = np.abs(spt.tail(a, T) - ref)
    assert err.max() < 2e-16 and (err / ref).max() < 5e-16
