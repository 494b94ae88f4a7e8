"""
GPU parity tests: the HIP path (through the C ABI, theano_pyglm_amd._lib) against
the CPU oracle on the same seeded inputs.  Tolerances (north_star): ll rtol 1e-5,
gradient 1e-4 of max|g|; the f64 path is held to far tighter bounds here.
"""
import numpy as np
import pytest

from tests import helpers as H
from oracle import glm_oracle as O

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-10      # f64 path (north_star asks 1e-5)
G_RTOL = 1e-9        # relative to max|g| (north_star asks 1e-4)


def _check(prob, n_lo=0, n_hi=None, f32=False, nchunks=0, ll_rtol=LL_RTOL, g_rtol=G_RTOL):
    dev = prob.device(f32=f32, nchunks=nchunks)
    n_hi = prob.N if n_hi is None else n_hi
    ll, g = dev.ll_grad(prob.theta[n_lo:n_hi], prob.Weff, n_lo, n_hi)
    ll0, g0 = prob.oracle_ll_grad(n_lo, n_hi)
    assert np.all(np.isfinite(ll)) and np.all(np.isfinite(g))
    assert np.allclose(ll, ll0, rtol=ll_rtol, atol=0), (ll, ll0)
    assert H.rel_err(g, g0) < g_rtol, H.rel_err(g, g0)
    # ll-only call gives the same ll
    ll2, _ = dev.ll_grad(prob.theta[n_lo:n_hi], prob.Weff, n_lo, n_hi, want_grad=False)
    assert np.array_equal(ll, ll2)
    dev.close()
    return ll, g


def test_features_golden(golden):
    """pgl_features == reference convolve_with_basis (basis.py:201-236) golden vectors."""
    from theano_pyglm_amd import _lib
    for key_s, key_f in (('conv_S', 'conv_fS'), ('conv_short_S', 'conv_short_fS'),
                         ('conv_one_S', 'conv_one_fS')):
        S = golden[key_s]
        ib = golden['conv_ibasis']
        d = _lib.DeviceGlm(S.shape[1], S.shape[0], ib.shape[1], ib.shape[0], 'explinear', 0.001)
        d.set_spikes(S)
        d.set_basis(ib)
        f = d.features()
        assert np.max(np.abs(f - golden[key_f])) < 1e-12
        d.close()


def test_ll_grad_c1_like():
    """standard_glm shape N=4 (KT=2 padded), ragged nT."""
    _check(H.Problem(4, 3001, H.std_ibasis(), seed=1))


def test_ll_grad_n32():
    """C2 shape N=32 (KT=10), 2 post tiles."""
    _check(H.Problem(32, 4096, H.std_ibasis(), seed=2))


def test_ll_grad_ragged_n20_weighted():
    """N=20: second post tile is ragged; Weff = A*W (sparse_weighted-like)."""
    _check(H.Problem(20, 2500, H.std_ibasis(), seed=3, weighted=True))


def test_ll_grad_n128():
    """C3 shape N=128 (KT=40, 8 post tiles, 2 post blocks)."""
    _check(H.Problem(128, 1536, H.std_ibasis(), seed=4, w_scale=0.5))


def test_ll_grad_subrange():
    """Neuron shard [n_lo,n_hi) as used by the multi-GPU split."""
    p = H.Problem(40, 2000, H.std_ibasis(), seed=5)
    _check(p, 16, 32)
    _check(p, 7, 40)


def test_ll_grad_exp_stim():
    """spatiotemporal_glm shape: exp nonlinearity, B=3, R=300, 9 dense stimulus columns."""
    _check(H.Problem(8, 2200, H.st_ibasis(), kind='exp', Dstim=9, seed=6))


def test_ll_grad_n64_b3_stim_kt13():
    """C5 shape N=64, B=3 + 9 stimulus columns -> 201 feature columns (KT=13)."""
    _check(H.Problem(64, 1200, H.st_ibasis(), kind='exp', Dstim=9, seed=7, w_scale=0.02))


def test_chunking_invariance():
    """Different time-chunk decompositions agree to reduction round-off."""
    p = H.Problem(16, 5000, H.std_ibasis(), seed=8)
    a, ga = _check(p, nchunks=1)
    b, gb = _check(p, nchunks=7)
    c, gc = _check(p, nchunks=313)
    assert np.allclose(a, b, rtol=1e-12) and np.allclose(a, c, rtol=1e-12)
    assert H.rel_err(ga, gc) < 1e-12


def test_f32_feature_staging():
    """Optional f32 LDS staging of the feature tile: still far inside rtol 1e-5."""
    p = H.Problem(32, 4096, H.std_ibasis(), seed=9)
    _check(p, f32=True, ll_rtol=1e-8, g_rtol=1e-6)


def test_high_rate_overflows_staging():
    """> 16 events per neuron per window: the global-memory fallback of the event staging."""
    p = H.Problem(8, 1500, H.std_ibasis(), seed=10, rate_hz=400.0, bias_mu=3.0, w_scale=0.05)
    _check(p)


def test_empty_spikes():
    p = H.Problem(4, 500, H.std_ibasis(), seed=11)
    p.S[:] = 0
    p._fS = None
    _check(p)


def test_impulse_currents_and_state():
    p = H.Problem(6, 1800, H.std_ibasis(), seed=12, weighted=True)
    dev = p.device()
    n = 3
    w = p.theta[n, 1:].reshape(p.N, p.B)
    I = dev.impulse_currents(w)
    I0 = O.impulse_currents(p.fS, w)
    assert np.max(np.abs(I - I0)) < 1e-11
    lam, inet, istim = dev.state(n, p.theta[n], p.Weff[:, n])
    x0, inet0, _ = O.glm_currents(n, p.fS, w, p.Weff[:, n], p.theta[n, 0])
    assert np.max(np.abs(inet - inet0)) < 1e-10
    assert np.allclose(lam, O.nlin(x0, p.kind), rtol=1e-12)
    assert np.all(istim == 0)
    # reference invariant lam_true == lam_sim (generate_synth_data.py:125-129): the
    # feature path equals the simulator's time-domain superposition
    imps = O.impulse_responses(p.ibasis, w)
    xd = O.direct_currents(p.S.astype(float), imps, p.Weff[:, n], p.theta[n, 0])
    assert np.allclose(lam, O.nlin(xd, p.kind))
    dev.close()


def test_mcmc_inner_ll():
    """gibbs.py:910-937 at the 10 Gauss-Hermite nodes + w=0 (gibbs.py:1002-1032)."""
    p = H.Problem(5, 2600, H.std_ibasis(), seed=13, weighted=True)
    dev = p.device()
    n_post, n_pre = 2, 4
    w = p.theta[n_post, 1:].reshape(p.N, p.B)
    I_imp = O.impulse_currents(p.fS, w)
    A = (p.Weff != 0).astype(float)
    I_other = O.other_current(I_imp, A, p.Weff, n_pre, n_post)
    ws, _ = O.gauss_hermite_nodes(0.0, 1.0)
    ws = np.concatenate((ws, [0.0]))
    ref = O.mcmc_inner_ll(ws, p.theta[n_post, 0], 0.0, I_other, I_imp[:, n_pre],
                          p.S[:, n_post].astype(float), p.dt, p.kind)
    got = dev.ll_from_current(n_post, p.theta[n_post, 0], None, I_other, I_imp[:, n_pre], ws)
    assert np.allclose(got, ref, rtol=1e-11)
    # device-resident form with the rank-1 downdate
    dev.gibbs_prepare(n_post, p.theta[n_post], p.Weff[:, n_post])
    got2 = dev.gibbs_ll(n_pre, p.Weff[n_pre, n_post], ws)
    assert np.allclose(got2, ref, rtol=1e-10)
    # 23 candidate weights (> one launch of 16)
    ws3 = np.linspace(-2, 2, 23)
    ref3 = O.mcmc_inner_ll(ws3, p.theta[n_post, 0], 0.0, I_other, I_imp[:, n_pre],
                           p.S[:, n_post].astype(float), p.dt, p.kind)
    assert np.allclose(dev.gibbs_ll(n_pre, p.Weff[n_pre, n_post], ws3), ref3, rtol=1e-10)
    # update: set the pair's weight to 0.7 and look at another presynaptic neuron
    dev.gibbs_update(n_pre, 0.7 - p.Weff[n_pre, n_post])
    W2 = p.Weff.copy()
    W2[n_pre, n_post] = 0.7
    I_other2 = O.other_current(I_imp, (W2 != 0).astype(float), W2, 1, n_post)
    ref4 = O.mcmc_inner_ll(ws, p.theta[n_post, 0], 0.0, I_other2, I_imp[:, 1],
                           p.S[:, n_post].astype(float), p.dt, p.kind)
    assert np.allclose(dev.gibbs_ll(1, W2[1, n_post], ws), ref4, rtol=1e-10)
    dev.close()


def test_errors():
    from theano_pyglm_amd import _lib
    d = _lib.DeviceGlm(4, 100, 5, 200, 'explinear', 0.001)
    with pytest.raises(_lib.PglError):
        d.ll_grad(np.zeros((4, 21)), np.ones((4, 4)))      # no data yet
    with pytest.raises(_lib.PglError):
        d.set_spikes(np.full((100, 4), 0.5))               # non-integer counts
    d.close()
    with pytest.raises(_lib.PglError):
        _lib.DeviceGlm(4, 100, 9, 200, 'explinear', 0.001)  # B > 8


def test_time_range_partial_sums():
    """pgl_set_time_range: partial ll/grad over time shards add up to the full evaluation
    (the time-sharded multi-GPU split; shards see the spikes before their range)."""
    p = H.Problem(24, 5000, H.std_ibasis(), seed=14, weighted=True)
    dev = p.device()
    ll_full, g_full = dev.ll_grad(p.theta, p.Weff)
    cuts = [0, 1616, 1632, 3200, 5000]
    ll_sum, g_sum = 0.0, 0.0
    for a, b in zip(cuts[:-1], cuts[1:]):
        dev.set_time_range(a, b)
        ll, g = dev.ll_grad(p.theta, p.Weff)
        ll_sum, g_sum = ll_sum + ll, g_sum + g
    assert np.allclose(ll_sum, ll_full, rtol=1e-12)
    assert H.rel_err(g_sum, g_full) < 1e-12
    ll0, g0 = p.oracle_ll_grad()
    assert np.allclose(ll_sum, ll0, rtol=LL_RTOL) and H.rel_err(g_sum, g0) < G_RTOL
    from theano_pyglm_amd import _lib
    with pytest.raises(_lib.PglError):
        dev.set_time_range(8, 100)          # not a multiple of 16
    dev.close()


def test_full_size_properties_c3():
    """BASELINE config C3 at full size (N=128, T=600 s, nT=600 000): size-independent
    properties of the device path, plus an oracle spot check on a time sub-range.
      * ll of the ll-only call == ll of the ll+grad call; chunk decomposition invariance;
      * additivity over time shards (the multi-GPU split);
      * directional derivative of the device ll == device gradient . direction;
      * oracle parity on bins [0, 24000) (features only depend on the past)."""
    rng = np.random.default_rng(77)
    N, nT = 128, 600000
    p = H.Problem(N, nT, H.std_ibasis(), seed=1234 + 3, w_scale=0.5)
    dev = p.device()
    ll, g = dev.ll_grad(p.theta, p.Weff)
    assert np.all(np.isfinite(ll)) and np.all(np.isfinite(g))
    ll_only, _ = dev.ll_grad(p.theta, p.Weff, want_grad=False)
    assert np.array_equal(ll, ll_only)
    # chunk decomposition
    from theano_pyglm_amd import _lib
    dev.set_option(_lib.OPT_NCHUNKS, 61)
    ll_c, g_c = dev.ll_grad(p.theta, p.Weff)
    dev.set_option(_lib.OPT_NCHUNKS, 0)
    assert np.allclose(ll_c, ll, rtol=1e-12) and H.rel_err(g_c, g) < 1e-11
    # time shards (8 ranks)
    from theano_pyglm_amd import parallel as PL
    ll_s, g_s = 0.0, 0.0
    for r in range(8):
        lo, hi = PL.time_shard_bounds(nT, r, 8)
        dev.set_time_range(lo, hi)
        a, b = dev.ll_grad(p.theta, p.Weff)
        ll_s, g_s = ll_s + a, g_s + b
    dev.set_time_range(0, nT)
    assert np.allclose(ll_s, ll, rtol=1e-12) and H.rel_err(g_s, g) < 1e-11
    # directional derivative (per neuron), central differences on the device ll
    d = rng.standard_normal(p.theta.shape)
    eps = 1e-6
    lp, _ = dev.ll_grad(p.theta + eps * d, p.Weff, want_grad=False)
    lm, _ = dev.ll_grad(p.theta - eps * d, p.Weff, want_grad=False)
    fd = (lp - lm) / (2 * eps)
    an = np.sum(g * d, axis=1)
    assert np.max(np.abs(fd - an)) < 1e-5 * np.max(np.abs(an))
    # oracle spot check on the first 24000 bins, 3 neurons
    nsub = 24000
    dev.set_time_range(0, nsub)
    q = H.Problem(N, nsub, H.std_ibasis(), seed=0)
    q.S, q.theta, q.Weff, q._fS = p.S[:nsub], p.theta, p.Weff, None
    oracle = {}
    for n in (0, 77, 127):
        a, b = dev.ll_grad(p.theta[n:n + 1], p.Weff, n, n + 1)
        a0, b0 = q.oracle_ll_grad(n, n + 1)
        oracle[n] = (a0[0], b0[0])
        assert np.allclose(a, a0, rtol=LL_RTOL) and H.rel_err(b, b0) < G_RTOL
    # ... and the block call on the same full-size handle: all 128 neurons in ONE launch of the two-pass
    # resident-tile kernel (the bench workload's kernel), restricted to the oracle's bins
    assert dev.info()['kernel_version'] == 5
    a, b = dev.ll_grad(p.theta, p.Weff)
    for n, (a0, b0) in oracle.items():
        assert np.allclose(a[n], a0, rtol=LL_RTOL) and H.rel_err(b[n], b0) < G_RTOL
    # a second sub-range in the middle of the recording (tile-aligned start; features reach back across it)
    lo = 16 * 20000
    dev.set_time_range(lo, lo + 8000)
    a, b = dev.ll_grad(p.theta, p.Weff)
    from oracle import glm_oracle as O
    S2 = p.S[lo - 2000:lo + 8000].astype(float)              # 2000 bins of history >> R = 200 taps
    fS = O.convolve_with_basis_fft(S2, p.ibasis)[2000:]
    for n in (5, 100):
        th = p.theta[n]
        ll0, gb0, _, gw0 = O.glm_ll_grad(n, S2[2000:], fS, th[1:].reshape(N, p.B),
                                         p.Weff[:, n], th[0], p.dt, p.kind, None, None)
        assert np.allclose(a[n], ll0, rtol=LL_RTOL)
        assert H.rel_err(b[n], np.concatenate(([gb0], gw0.reshape(-1)))) < G_RTOL
    dev.close()


def test_tiny_shapes():
    """Degenerate sizes: one neuron; fewer bins than one 16-row tile; fewer bins than taps."""
    _check(H.Problem(1, 40, H.std_ibasis(), seed=15, rate_hz=100.0))
    _check(H.Problem(3, 7, H.std_ibasis(), seed=16, rate_hz=200.0))
    _check(H.Problem(2, 1, H.std_ibasis(), seed=17, rate_hz=500.0))
    _check(H.Problem(17, 100, H.st_ibasis(), kind='exp', seed=18, rate_hz=50.0, w_scale=0.01))


def test_sliced_path_large_populations():
    """More than 128 neurons / 640 feature columns: the 3-phase sliced path (forward launches
    accumulating the currents, one elementwise pass, backward launches per slice)."""
    _check(H.Problem(150, 900, H.std_ibasis(), seed=30, w_scale=0.3))                 # 128 + 22
    _check(H.Problem(260, 400, H.std_ibasis(), seed=31, w_scale=0.2, weighted=True))  # 3 slices
    # 640 impulse columns + 8 stimulus columns -> the stimulus gets a slice of its own
    _check(H.Problem(128, 500, H.std_ibasis(), seed=32, Dstim=8, w_scale=0.3))
    # dense stimulus wider than one slice (700 columns), few neurons
    _check(H.Problem(6, 600, H.st_ibasis(), kind='exp', Dstim=700, seed=33, w_scale=0.02), g_rtol=1e-8)
    # neuron shard + time range on the sliced path
    p = H.Problem(140, 1000, H.std_ibasis(), seed=34, w_scale=0.3)
    dev = p.device()
    ll_f, g_f = dev.ll_grad(p.theta[100:140], p.Weff, 100, 140)
    ll0, g0 = p.oracle_ll_grad(100, 140)
    assert np.allclose(ll_f, ll0, rtol=LL_RTOL) and H.rel_err(g_f, g0) < G_RTOL
    acc_ll, acc_g = 0.0, 0.0
    for a, b in ((0, 496), (496, 1000)):
        dev.set_time_range(a, b)
        ll, g = dev.ll_grad(p.theta[100:140], p.Weff, 100, 140)
        acc_ll, acc_g = acc_ll + ll, acc_g + g
    assert np.allclose(acc_ll, ll_f, rtol=1e-12) and H.rel_err(acc_g, g_f) < 1e-12
    dev.close()


def test_unsupported_shapes_fail_loudly():
    from theano_pyglm_amd import _lib
    q = H.Problem(128, 64, H.std_ibasis(), seed=20)
    dq = q.device()
    with pytest.raises((_lib.PglError, ValueError)):
        dq.ll_grad(q.theta[:3], q.Weff, 5, 4)                # empty / reversed neuron range
    with pytest.raises(_lib.PglError, match="range"):
        dq.ll_grad(q.theta[:3], q.Weff, 127, 130)            # beyond N
    dq.close()


def test_f32_feature_kernel_agrees():
    """PGL_OPT_FEATURE_F32 (f32 feature tile / basis taps in the K-split kernel) against the default
    f64 path, and the forced two-pass kernels on a population of 3 post tiles."""
    from theano_pyglm_amd import _lib
    p = H.Problem(48, 3000, H.std_ibasis(), seed=21, weighted=True)
    d0 = p.device()
    ll0, g0 = d0.ll_grad(p.theta, p.Weff)
    assert d0.info()['kernel_version'] == 7           # short feature rows, 3 post tiles: one wave per tile, resident tiles
    assert d0.info()['resident_feature_bytes'] > 0
    for kern in (2, 3, 4, 6):
        d1 = p.device()
        d1.set_option(_lib.OPT_KERNEL, kern)
        ll1, g1 = d1.ll_grad(p.theta, p.Weff)
        assert np.allclose(ll1, ll0, rtol=1e-12) and H.rel_err(g1, g0) < 1e-12
        d1.close()
    d2 = p.device(f32=True)
    ll2, g2 = d2.ll_grad(p.theta, p.Weff)
    assert d2.info()['kernel_version'] == 3
    assert np.allclose(ll2, ll0, rtol=1e-8) and H.rel_err(g2, g0) < 1e-6
    d0.close()
    d2.close()


def test_nonfinite_semantics_match_reference_expression():
    """lam underflows to 0 for x < -745: glm.py:52 gives log(0)*S = NaN (and a NaN gradient),
    which fit_glm maps to 1e16 / 0 (coord_descent.py:170-182).  exp overflow gives -inf.  Other
    neurons of the same launch stay finite."""
    p = H.Problem(5, 800, H.std_ibasis(), seed=60)
    p.theta[1, 0] = -900.0                       # neuron 1: lam == 0 everywhere
    dev = p.device()
    ll, g = dev.ll_grad(p.theta, p.Weff)
    ll0, g0 = p.oracle_ll_grad()
    assert np.isnan(ll[1]) and np.isnan(ll0[1])
    assert np.all(np.isnan(g[1])) 
    ok = [0, 2, 3, 4]
    assert np.allclose(ll[ok], ll0[ok], rtol=LL_RTOL) and H.rel_err(g[ok], g0[ok]) < G_RTOL
    dev.close()
    q = H.Problem(3, 400, H.st_ibasis(), kind='exp', seed=61, w_scale=0.01)
    q.theta[2, 0] = 800.0                        # exp(800) = inf
    dq = q.device()
    ll, g = dq.ll_grad(q.theta, q.Weff)
    with np.errstate(over='ignore', invalid='ignore'):
        ll0, g0 = q.oracle_ll_grad()
    assert ll[2] == -np.inf and ll0[2] == -np.inf
    assert np.allclose(ll[:2], ll0[:2], rtol=LL_RTOL)
    dq.close()


def test_external_stream_and_timing_window():
    """pgl_set_stream orders the handle's work on a caller-owned stream (torch's current stream, the
    one RCCL collectives are ordered against); every launch records its own event set and
    pgl_timing_summary averages them without a host sync between launches."""
    import torch
    p = H.Problem(16, 4000, H.std_ibasis(), seed=70)
    dev = p.device()
    ll0, g0 = dev.ll_grad(p.theta, p.Weff)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dev.set_stream(stream.cuda_stream)
        d_theta = torch.from_numpy(p.theta).cuda()
        d_W = torch.from_numpy(p.Weff).cuda()
        d_ll = torch.zeros(16, dtype=torch.float64, device='cuda')
        d_g = torch.zeros((16, p.theta.shape[1]), dtype=torch.float64, device='cuda')
        for _ in range(5):                                   # queued back to back, no host sync
            dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
        doubled = d_ll * 2.0                                 # torch op ordered after the evaluations
        stream.synchronize()
    n, fused, total = dev.timing_summary(reset=True)
    assert n == 5 and 0 < fused <= total
    assert dev.timing_summary()[0] == 0
    assert np.array_equal(d_ll.cpu().numpy(), ll0) and np.array_equal(d_g.cpu().numpy(), g0)
    assert np.array_equal(doubled.cpu().numpy(), 2.0 * ll0)
    dev.set_stream(None)                                     # back to the handle's own stream
    ll1, _ = dev.ll_grad(p.theta, p.Weff)
    assert np.array_equal(ll1, ll0)
    dev.close()


def test_two_pass_kernel_shapes_and_agreement():
    """The two-pass kernels (default from 5 post tiles = 65 neurons on: k_fused5 on resident feature
    tiles; k_fused3 with on-the-fly features) against the oracle and against the K-split kernel on
    shapes that stress the column split: odd k-tile counts, stimulus
    columns in the second half, ragged neuron counts, a time range off the tile grid, ll only."""
    from theano_pyglm_amd import _lib
    # N=80, B=5: 400 columns = 25 k-tiles -> halves of 13 (padded); N=96, B=3 + 9 stimulus columns
    for p in (H.Problem(80, 1800, H.std_ibasis(), seed=80, w_scale=0.5),
              H.Problem(96, 1500, H.st_ibasis(), kind='exp', Dstim=9, seed=81, w_scale=0.02),
              H.Problem(70, 1000, H.std_ibasis(), seed=82, weighted=True, w_scale=0.5),
              H.Problem(128, 700, H.std_ibasis(), seed=83, Dstim=0, w_scale=0.5)):
        ll, g = _check(p)                                  # default: resident feature tiles (k_fused5)
        assert p.device().info()['kernel_version'] == 5
        # K-split, two-pass on the fly (rows of more than 32 k-tiles -- N = 128 -- go to the K-split kernel: k_fused3 holds
        # at most 16 k-tiles per half in its registers), two-pass resident
        need = -(-(p.N * p.B + p.Dstim) // 16)
        for kern, ver in ((2, 2), (3, 4 if need <= 32 else 2), (4, 5)):
            dk = p.device()
            dk.set_option(_lib.OPT_KERNEL, kern)
            assert dk.info()['kernel_version'] == ver
            llk, gk = dk.ll_grad(p.theta, p.Weff)
            assert np.allclose(llk, ll, rtol=1e-12) and H.rel_err(gk, g) < 1e-12
            # new spikes / basis invalidate the resident tiles: same answer after a re-upload
            if kern == 4:
                dk.set_spikes(p.S)
                dk.set_basis(p.ibasis)
                if p.Dstim:
                    dk.set_stim_features(p.fstim)
                ll5, g5 = dk.ll_grad(p.theta, p.Weff)
                assert np.array_equal(ll5, llk) and np.array_equal(g5, gk)
                assert dk.info()['resident_feature_bytes'] > 0
            dk.close()
    # forced on a small population (2 post tiles, 6 idle waves) and with a restricted time range
    p = H.Problem(24, 3000, H.std_ibasis(), seed=84, weighted=True)
    d3 = p.device()
    d3.set_option(_lib.OPT_KERNEL, 3)
    ll3, g3 = d3.ll_grad(p.theta, p.Weff)
    ll0, g0 = p.oracle_ll_grad()
    assert np.allclose(ll3, ll0, rtol=LL_RTOL) and H.rel_err(g3, g0) < G_RTOL
    d3.set_time_range(1008, 2501)                          # t_hi off the 16-bin tile grid
    lla, ga = d3.ll_grad(p.theta, p.Weff)
    d2 = p.device()
    d2.set_option(_lib.OPT_KERNEL, 2)
    d2.set_time_range(1008, 2501)
    llb, gb = d2.ll_grad(p.theta, p.Weff)
    assert np.allclose(lla, llb, rtol=1e-12) and H.rel_err(ga, gb) < 1e-12
    d3.close()
    d2.close()


def test_two_pass_kernel_helper_waves():
    """k_fused5<.., HLP = 1>: in a block of five or six post tiles (N = 65 .. 96) the waves without a tile take over the
    second half of the forward k-steps (pass 1) and of the k-tiles (pass 2) of two tiles.  Against the oracle, and against
    the same launch without helpers (dev option 92): tiles without a helper agree bit for bit, helped tiles to rounding
    (their currents are the sum of two partial sums); list launches of 70 / 90 neurons (a lock-step sweep's shapes) and a
    time range off the chunk grid included."""
    from theano_pyglm_amd import _lib
    for N, seed, kw in ((80, 90, {}), (96, 91, {}), (72, 92, {'weighted': True}), (88, 93, {'Dstim': 7})):
        p = H.Problem(N, 4000, H.std_ibasis(), seed=seed, w_scale=0.5, **kw)
        ll0, g0 = p.oracle_ll_grad()
        d = p.device()
        assert d.info()['kernel_version'] == 5
        names = _lib.plan_kernels(N, B=p.B, R=p.R, Dstim=p.Dstim, nT=p.nT)
        assert any(n.startswith('k_fused5<') and n.endswith(', 1>') for n in names), names
        ll, g = d.ll_grad(p.theta, p.Weff)
        assert np.allclose(ll, ll0, rtol=LL_RTOL) and H.rel_err(g, g0) < G_RTOL
        d.set_option(92, 1)
        lln, gn = d.ll_grad(p.theta, p.Weff)
        d.set_option(92, 0)
        nb = -(-N // 16)
        helped = (0, 4) if nb == 5 else (0, 1)
        for t in range(nb):
            sl = slice(16 * t, min(16 * t + 16, N))
            if t in helped:
                assert np.allclose(ll[sl], lln[sl], rtol=1e-12) and H.rel_err(g[sl], gn[sl]) < 1e-12
                assert not np.array_equal(g[sl], gn[sl])       # (the helper really took part)
            else:
                assert np.array_equal(ll[sl], lln[sl]) and np.array_equal(g[sl], gn[sl])
        # ll only; a sub-range of the recording
        llo, _ = d.ll_grad(p.theta, p.Weff, want_grad=False)
        assert np.array_equal(llo, ll)
        d.set_time_range(1008, 3001)
        lla, ga = d.ll_grad(p.theta, p.Weff)
        d.set_option(92, 1)
        llb, gb = d.ll_grad(p.theta, p.Weff)
        assert np.allclose(lla, llb, rtol=1e-12) and H.rel_err(ga, gb) < 1e-12
        d.close()


def test_helper_waves_in_light_blocks():
    """Helpers also serve a block of three post tiles of a single slice (a 33 .. 48-neuron range or list against a row of more
    than 20 k-tiles) and the last block of one or two tiles of a wide population (every form of the kernel: forward-only
    column slices, pass 1 from the slab, pass 2 on the L and H parts).  Against the oracle and against the same launches
    without helpers (dev option 92): every tile of such a block is helped -- equal to rounding, not bit for bit."""
    from theano_pyglm_amd import _lib
    p = H.Problem(100, 3000, H.std_ibasis(), seed=77, w_scale=0.4)            # 32 k-tiles per row
    d = p.device()
    for lo, hi in ((10, 50), (60, 100), (3, 36)):
        names = _lib.plan_kernels(100, B=p.B, R=p.R, nT=p.nT, n_lo=lo, count=hi - lo)
        assert all(n.startswith('k_fused5<') and n.endswith(', 1>') for n in names), names
        ll, g = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
        llo, go = p.oracle_ll_grad(lo, hi)
        assert np.allclose(ll, llo, rtol=LL_RTOL) and H.rel_err(g, go) < G_RTOL
        d.set_option(92, 1)
        lln, gn = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
        d.set_option(92, 0)
        assert np.allclose(ll, lln, rtol=1e-12) and H.rel_err(g, gn) < 1e-12 and not np.array_equal(g, gn)
    d.close()
    for N, nT in ((130, 2500), (160, 6000)):                                  # wide: 8 + 1 and 8 + 2 post tiles
        p = H.Problem(N, nT, H.std_ibasis(), seed=N + 1, w_scale=0.4)
        names = _lib.plan_kernels(N, B=p.B, R=p.R, nT=nT)
        assert all(n.startswith('k_fused5<') and n.endswith(', 1>') for n in names), names
        d = p.device()
        ll, g = d.ll_grad(p.theta, p.Weff)
        llo, go = p.oracle_ll_grad(N - 20, N)
        assert np.allclose(ll[N - 20:], llo, rtol=LL_RTOL) and H.rel_err(g[N - 20:], go) < G_RTOL
        d.set_option(92, 1)
        lln, gn = d.ll_grad(p.theta, p.Weff)
        assert np.array_equal(ll[:128], lln[:128]) and np.array_equal(g[:128], gn[:128])         # the full block: no helpers
        assert np.allclose(ll, lln, rtol=1e-12) and H.rel_err(g, gn) < 1e-12 and not np.array_equal(g[128:], gn[128:])
        d.close()


def test_forced_kernels_on_tiny_shapes():
    """Forcing the two-pass kernels on tiny problems (one k-tile per half, a single time tile, one
    neuron) must either run correctly or fall back to the K-split kernel -- never misbehave."""
    from theano_pyglm_amd import _lib
    for p in (H.Problem(4, 300, H.std_ibasis(), seed=90),             # K = 20: halves of one k-tile
              H.Problem(1, 40, H.std_ibasis(), seed=91, rate_hz=100.0),    # K = 5: stays on the K-split kernel
              H.Problem(7, 16, H.std_ibasis(), seed=92, rate_hz=200.0),    # exactly one time tile
              H.Problem(33, 17, H.st_ibasis(), kind='exp', Dstim=2, seed=93, rate_hz=100.0, w_scale=0.01)):
        ll0, g0 = p.oracle_ll_grad()
        for kern in (2, 3, 4, 6, 7):
            d = p.device()
            d.set_option(_lib.OPT_KERNEL, kern)
            ll, g = d.ll_grad(p.theta, p.Weff)
            assert np.allclose(ll, ll0, rtol=LL_RTOL, atol=1e-12), (kern, p.N)
            assert H.rel_err(g, g0) < G_RTOL, (kern, p.N)
            ll2, _ = d.ll_grad(p.theta, p.Weff, want_grad=False)
            assert np.array_equal(ll, ll2)
            d.close()


def test_batched_gibbs_columns_exp_nonlinearity_and_narrow_launches():
    """pgl_gibbs_ll_cols with the exp nonlinearity (log lam = x), K = 1 ... 16 candidate weights, one
    column per launch (the ARS probes) up to all columns, against the oracle and the one-pair path."""
    N = 7
    p = H.Problem(N, 2100, H.st_ibasis(), kind='exp', seed=71, weighted=True, w_scale=0.02, rate_hz=30.0)
    d = p.device()
    d.gibbs_prepare_all(p.theta, p.Weff)
    A = (p.Weff != 0).astype(float)
    for K in (1, 5, 16, 23):
        for cols in (np.arange(N), np.array([4]), np.array([6, 1, 3])):
            pre = (cols * 3 + 2) % N
            ws = np.linspace(-1.5, 1.5, K)[None, :] * np.ones((len(cols), 1)) + 0.1 * cols[:, None]
            got = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
            assert got.shape == (len(cols), K)
            for i, c in enumerate(cols):
                w = p.theta[c, 1:].reshape(N, p.B)
                I_imp = O.impulse_currents(p.fS, w)
                I_other = O.other_current(I_imp, A, p.Weff, pre[i], c)
                ref = O.mcmc_inner_ll(ws[i], p.theta[c, 0], 0.0, I_other, I_imp[:, pre[i]], p.S[:, c].astype(float),
                                      p.dt, p.kind)
                assert np.allclose(got[i], ref, rtol=1e-10)
    # the currents kept on the device equal the oracle's total current (without the bias)
    xcur = d.gibbs_currents(2)
    w = p.theta[2, 1:].reshape(N, p.B)
    assert np.max(np.abs(xcur - O.impulse_currents(p.fS, w).dot(p.Weff[:, 2]))) < 1e-10
    d.close()


@pytest.mark.gpu
def test_batched_gibbs_regime_split_kernels_match_f64_kernel_and_oracle():
    """explinear pgl_gibbs_ll_cols: the regime-split kernels (f32 log1p term where |x| > 12, compacted f64
    band, spike terms from the event lists) against the all-f64 kernel (PGL_OPT_GIBBS_KERNEL = 1) and the
    oracle, with currents that put bins into every regime: x >> 12, x << -12, the band |x| <= 12, and
    x < -745 (lam underflows: NaN by glm.py:52).  Narrow (1-2 columns: time-split items), wide and ragged
    launches; a recording that is not a multiple of the 256-bin block; a time sub-range."""
    from theano_pyglm_amd import _lib
    N = 11
    p = H.Problem(N, 2100 + 77, H.std_ibasis(), kind='explinear', seed=93, weighted=True, w_scale=1.0,
                  rate_hz=25.0, bias_mu=6.0)
    p.theta[:4, 0] = [20.0, -3.0, 14.0, 0.5]
    d = p.device()
    A = (p.Weff != 0).astype(float)

    def oracle(cols, pre, ws, t_lo=0, t_hi=None):
        out = np.zeros((len(cols), ws.shape[1]))
        for i, c in enumerate(cols):
            w = p.theta[c, 1:].reshape(N, p.B)
            I_imp = O.impulse_currents(p.fS, w)
            I_other = O.other_current(I_imp, A, p.Weff, pre[i], c)
            sl = slice(t_lo, t_hi)
            out[i] = O.mcmc_inner_ll(ws[i], p.theta[c, 0], 0.0, I_other[sl], I_imp[sl, pre[i]],
                                     p.S[sl, c].astype(float), p.dt, p.kind)
        return out

    d.gibbs_prepare_all(p.theta, p.Weff)
    seen_nan = seen_finite = False
    for K in (1, 11, 16):
        for cols in (np.arange(N), np.array([2]), np.array([7, 0]), np.array([3, 9, 1]), np.arange(N)[::-1][:9]):
            pre = (cols * 5 + 1) % N
            # weights from tiny to huge: w*ic spans the band, the deep tails and the underflow
            base = np.concatenate(([0.0], np.geomspace(0.01, 300.0, 15)))[:K]
            ws = np.where(np.arange(K)[None, :] % 2 == 0, 1.0, -1.0) * base[None, :] * (1.0 + 0.05 * cols[:, None])
            d.set_option(_lib.OPT_GIBBS_KERNEL, 1)
            old = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
            d.set_option(_lib.OPT_GIBBS_KERNEL, 0)
            new = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
            again = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
            assert np.array_equal(new, again, equal_nan=True)            # fixed summation order
            # (the all-f64 kernel flushes lam in the last binade above the underflow, x in (-745.13, -744.4),
            #  to 0 -> NaN; the regime-split kernel and the oracle keep the denormal: compare where both are finite)
            both = np.isfinite(new) & np.isfinite(old)
            assert both.sum() >= 0.8 * np.isfinite(old).sum()
            assert np.allclose(new[both], old[both], rtol=1e-11, atol=0)
            with np.errstate(all='ignore'):
                ref = oracle(cols, pre, ws)
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(new), fin)
            assert np.allclose(new[fin], ref[fin], rtol=1e-10, atol=0)
            seen_nan |= bool((~fin).any())
            seen_finite |= bool(fin.any())
    assert seen_nan and seen_finite
    # a time sub-range (partial last block, first block not at bin 0)
    d.set_time_range(512, 1999)
    d.gibbs_prepare_all(p.theta, p.Weff)
    cols = np.array([0, 5, 10, 2])
    pre = np.array([3, 5, 0, 2])
    ws = np.linspace(-2.0, 2.0, 7)[None, :] * np.ones((4, 1))
    new = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
    with np.errstate(all='ignore'):
        ref = oracle(cols, pre, ws, 512, 1999)
    assert np.allclose(new, ref, rtol=1e-10, atol=0, equal_nan=True)
    d.close()


@pytest.mark.gpu
def test_batched_gibbs_shared_presynaptic_neuron_path():
    """A sweep step of the collapsed sampler lists the pair (j -> n) for every n: all columns share the presynaptic
    neuron, and k_gibbs_rate_cols then takes the pair currents from the filtered spike train of that one neuron
    (k_gibbs_pre_features, B multiply-adds per bin) instead of walking the events of the window per column.  Same
    numbers as the event-loop form (debug option bit 0x1000 forces it), the all-f64 kernel and the oracle; wide,
    narrow (time-split) and ragged launches, several sub-blocks per workgroup, a time sub-range."""
    from theano_pyglm_amd import _lib
    N = 13
    p = H.Problem(N, 256 * 6 + 41, H.std_ibasis(), kind='explinear', seed=77, weighted=True, w_scale=0.8,
                  rate_hz=30.0, bias_mu=9.0)
    p.theta[:3, 0] = [20.0, -2.0, 13.0]
    d = p.device()
    A = (p.Weff != 0).astype(float)

    def oracle(cols, pre, ws, t_lo=0, t_hi=None):
        out = np.zeros((len(cols), ws.shape[1]))
        for i, c in enumerate(cols):
            w = p.theta[c, 1:].reshape(N, p.B)
            I_imp = O.impulse_currents(p.fS, w)
            I_other = O.other_current(I_imp, A, p.Weff, pre[i], c)
            sl = slice(t_lo, t_hi)
            out[i] = O.mcmc_inner_ll(ws[i], p.theta[c, 0], 0.0, I_other[sl], I_imp[sl, pre[i]],
                                     p.S[sl, c].astype(float), p.dt, p.kind)
        return out

    d.gibbs_prepare_all(p.theta, p.Weff)
    for t_rng in (None, (304, 1500)):
        if t_rng:
            d.set_time_range(*t_rng)
            d.gibbs_prepare_all(p.theta, p.Weff)
        for n_pre in (0, 5, 12):
            for cols in (np.arange(N), np.array([4, 9]), np.arange(N)[::-1][:7]):
                pre = np.full(len(cols), n_pre)
                ws = np.linspace(-3.0, 3.0, 11)[None, :] * (1.0 + 0.1 * cols[:, None])
                for nloop in (0, 3):
                    d.set_option(99, nloop << 8)
                    shared = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
                    assert np.array_equal(shared, d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws), equal_nan=True)
                    d.set_option(99, (nloop << 8) | 0x1000)
                    events = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
                    d.set_option(99, 0)
                    assert np.array_equal(np.isfinite(shared), np.isfinite(events))
                    fin = np.isfinite(shared)
                    assert np.allclose(shared[fin], events[fin], rtol=1e-12, atol=0)
                d.set_option(_lib.OPT_GIBBS_KERNEL, 1)
                f64 = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
                d.set_option(_lib.OPT_GIBBS_KERNEL, 0)
                both = np.isfinite(f64) & fin
                assert np.allclose(shared[both], f64[both], rtol=1e-11, atol=0)
                with np.errstate(all='ignore'):
                    ref = oracle(cols, pre, ws, *(t_rng or (0, None)))
                rfin = np.isfinite(ref)
                assert np.array_equal(fin, rfin)
                assert np.allclose(shared[rfin], ref[rfin], rtol=1e-10, atol=0)
    d.close()


@pytest.mark.gpu
def test_batched_gibbs_sub_block_loop_and_event_bursts():
    """k_gibbs_rate_cols: several 256-bin sub-blocks per workgroup (forced through the debug option, as the
    full-size launches have them) with a ragged tail, and presynaptic bursts that overflow the LDS event
    staging (> 20 events per window -> the window tables are walked in global memory)."""
    from theano_pyglm_amd import _lib
    N = 9
    for rate, nT in ((20.0, 256 * 7 + 100), (160.0, 256 * 5 + 3)):
        p = H.Problem(N, nT, H.std_ibasis(), kind='explinear', seed=int(rate), weighted=True, w_scale=0.4,
                      rate_hz=rate, bias_mu=15.0)
        d = p.device()
        A = (p.Weff != 0).astype(float)
        d.gibbs_prepare_all(p.theta, p.Weff)
        cols = np.arange(N)
        pre = (cols * 2 + 3) % N
        ws = np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0]))[None, :] * np.ones((N, 1))
        ref = np.zeros((N, 11))
        for i, c in enumerate(cols):
            w = p.theta[c, 1:].reshape(N, p.B)
            I_imp = O.impulse_currents(p.fS, w)
            I_other = O.other_current(I_imp, A, p.Weff, pre[i], c)
            with np.errstate(all='ignore'):
                ref[i] = O.mcmc_inner_ll(ws[i], p.theta[c, 0], 0.0, I_other, I_imp[:, pre[i]],
                                         p.S[:, c].astype(float), p.dt, p.kind)
        fin = np.isfinite(ref)
        assert fin.mean() > 0.5
        base = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
        assert np.array_equal(np.isfinite(base), fin) and np.allclose(base[fin], ref[fin], rtol=1e-10, atol=0)
        for nloop in (2, 3, 8):
            d.set_option(99, nloop << 8)
            got = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
            assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], ref[fin], rtol=1e-10, atol=0)
            one = d.gibbs_ll_cols(cols[4:5], pre[4:5], p.Weff[pre[4:5], cols[4:5]], ws[4:5])
            assert np.allclose(one[0][fin[4]], ref[4][fin[4]], rtol=1e-10, atol=0)
        d.set_option(99, 0)
        d.close()


@pytest.mark.gpu
def test_batched_gibbs_random_shapes_against_f64_kernel():
    """Random population sizes (incl. N = 1 and N not a multiple of 8 or 16), recording lengths below and
    above one 256-bin block, 1..16 weights, arbitrary column lists with repeats: the regime-split kernels
    and the all-f64 kernel agree wherever both are finite, and the NaN pattern only differs inside the last
    binade above the underflow (see the regime-split test)."""
    from theano_pyglm_amd import _lib
    rng = np.random.RandomState(77)
    for N, nT in ((1, 37), (3, 300), (17, 1000), (40, 257), (9, 4099)):
        p = H.Problem(N, nT, H.std_ibasis(), kind='explinear', seed=N + nT, weighted=True, w_scale=0.7,
                      rate_hz=30.0, bias_mu=float(rng.uniform(-2.0, 22.0)))
        d = p.device()
        d.gibbs_prepare_all(p.theta, p.Weff)
        for _ in range(4):
            K = int(rng.randint(1, 17))
            nc = int(rng.randint(1, 2 * N + 2))
            cols = rng.randint(0, N, size=nc)
            pre = rng.randint(0, N, size=nc)
            ws = rng.standard_normal((nc, K)) * rng.choice([0.1, 1.0, 20.0])
            aw = p.Weff[pre, cols]
            d.set_option(_lib.OPT_GIBBS_KERNEL, 1)
            old = d.gibbs_ll_cols(cols, pre, aw, ws)
            d.set_option(_lib.OPT_GIBBS_KERNEL, 0)
            new = d.gibbs_ll_cols(cols, pre, aw, ws)
            both = np.isfinite(new) & np.isfinite(old)
            assert both.sum() >= 0.9 * np.isfinite(old).sum()
            assert np.allclose(new[both], old[both], rtol=1e-11, atol=0)
        d.close()


@pytest.mark.gpu
def test_denormal_rate_of_one_neuron_does_not_leak_into_its_post_tile():
    """A neuron whose rate is denormal at its spike bins (x ~ -730: 1/lam overflows) gets a non-finite
    gradient of its own -- but the compacted spike terms of pgl_rate4 must not carry 0 * inf into the other
    fifteen neurons of its post tile (every kernel family; regression: k_fused5 / k_fused7 did)."""
    from theano_pyglm_amd import _lib
    for N in (48, 128):
        p = H.Problem(N, 4000, H.std_ibasis(), seed=5, w_scale=0.5)
        base_ll = base_g = None
        for bias in (20.0, -730.0, 800.0, -900.0):
            th = p.theta.copy()
            th[5, 0] = bias
            for kern in (0, 2, 3, 4, 6, 7):
                d = p.device()
                d.set_option(_lib.OPT_KERNEL, kern)
                ll, g = d.ll_grad(th, p.Weff)
                ver = d.info()['kernel_version']
                d.close()
                others = np.arange(N) != 5
                assert np.all(np.isfinite(ll[others])) and np.all(np.isfinite(g[others])), (N, bias, kern, ver)
                if bias == 20.0 and kern == 0:
                    base_ll, base_g = ll, g
                assert np.allclose(ll[others], base_ll[others], rtol=1e-12, atol=0)
                assert np.allclose(g[others], base_g[others], rtol=1e-10, atol=1e-9 * np.abs(base_g).max())


@pytest.mark.gpu
def test_kernel_families_agree_on_wild_parameters():
    """Seeded short run of tools/fuzz_kernels.py: every kernel family against the K-split kernel with in-kernel
    features on random shapes with a few neurons at extreme biases / weights per trial (rates at the underflow
    and overflow edges): same finite pattern per neuron, same values where finite."""
    from theano_pyglm_amd import _lib
    rng = np.random.RandomState(2024)
    seen_nonfinite = 0
    for trial in range(14):
        N = int(rng.choice([17, 48, 64, 80, 128, 130]))
        kind = str(rng.choice(['explinear', 'exp']))
        nT = int(rng.choice([700, 2500]))
        Dstim = int(rng.choice([0, 0, 3]))
        p = H.Problem(N, nT, H.std_ibasis(), kind=kind, seed=int(rng.randint(1 << 30)), weighted=bool(rng.rand() < 0.5),
                      Dstim=Dstim, rate_hz=float(rng.choice([5.0, 20.0, 60.0])))
        th = p.theta.copy()
        for n in rng.choice(N, size=int(rng.randint(1, 5)), replace=False):
            mode = rng.randint(4)
            if mode == 0:
                th[n, 0] = rng.uniform(-800, 800)
            elif mode == 1:
                th[n, 1:] *= rng.choice([20.0, 200.0, 2000.0])
            elif mode == 2:
                th[n, 0] = rng.choice([-730.0, -745.0, -746.0, -709.0, 709.0, 745.0])
            else:
                th[n, 1 + Dstim:1 + Dstim + p.B] = rng.choice([1e3, -1e3, 1e5])
        if kind == 'exp':
            th[:, 0] = np.clip(th[:, 0], -800, 700)
        res = {}
        for kern in (2, 0, 3, 4, 6, 7):
            d = p.device()
            d.set_option(_lib.OPT_KERNEL, kern)
            ll, g = d.ll_grad(th, p.Weff)
            res[kern] = (ll, g, d.info()['kernel_version'])
            d.close()
        ll0, g0, _ = res[2]
        for kern in (0, 3, 4, 6, 7):
            ll, g, ver = res[kern]
            if ver == 2:
                continue
            f0, f1 = np.isfinite(ll0), np.isfinite(ll)
            gf0, gf1 = np.isfinite(g0).all(1), np.isfinite(g).all(1)
            assert np.array_equal(f0, f1) and np.array_equal(gf0, gf1), (trial, N, kind, kern, ver)
            assert np.allclose(ll[f0], ll0[f0], rtol=1e-9, atol=0), (trial, N, kind, kern, ver)
            if gf0.any():
                scale = np.abs(g0[gf0]).max()
                assert np.allclose(g[gf0], g0[gf0], rtol=1e-8, atol=1e-9 * scale), (trial, N, kind, kern, ver)
            seen_nonfinite += int((~gf1).sum())
    assert seen_nonfinite > 0


@pytest.mark.gpu
def test_narrow_post_blocks_of_a_wide_population_one_image_buffer():
    """16 / 32 post neurons of a 128-neuron population (K = 640: a neuron-sharded rank, or a late line-search launch
    of the lock-step MAP).  A 16-bin tile of 640 columns is 81 KB -- it does not fit the LDS twice -- so k_fused6 runs
    with ONE image buffer (8 waves: K split 8 ways for one post tile, 4 ways for two).  Against the oracle, against
    k_fused2 (option 95 = 2: in-kernel features), contiguous blocks, a neuron list, mixed regimes, a time sub-range
    that is not a whole number of tiles, ll-only == ll+grad."""
    N = 128
    p = H.Problem(N, 16 * 150 + 7, H.std_ibasis(), kind='explinear', seed=61, weighted=True, w_scale=0.6)
    p.theta[:40:3, 0] = np.linspace(-3.0, 9.0, 14)          # some neurons inside / below the |x| < 12 band
    d = p.device()
    W = np.ascontiguousarray(p.Weff)
    for lo, hi in ((0, 16), (16, 48), (100, 128), (5, 21)):
        d.set_option(95, 0)
        ll, g = d.ll_grad(p.theta[lo:hi], W, lo, hi)
        assert d.info(lo, hi)['kernel_version'] == 6
        ll0, _ = d.ll_grad(p.theta[lo:hi], W, lo, hi, want_grad=False)
        assert np.array_equal(ll, ll0)
        lr, gr = p.oracle_ll_grad(lo, hi)
        assert np.allclose(ll, lr, rtol=1e-10, atol=0) and np.allclose(g, gr, rtol=1e-9, atol=1e-9 * np.abs(gr).max())
        d.set_option(95, 2)
        ll2, g2 = d.ll_grad(p.theta[lo:hi], W, lo, hi)
        assert d.info(lo, hi)['kernel_version'] == 2
        assert np.allclose(ll, ll2, rtol=1e-12, atol=0) and np.allclose(g, g2, rtol=1e-11, atol=1e-11 * np.abs(g2).max())
    d.set_option(95, 0)
    import torch
    idx = np.array([3, 77, 12, 120, 64, 9, 31], dtype=np.int32)
    t_idx, t_th, t_W = torch.from_numpy(idx).cuda(), torch.from_numpy(p.theta[idx].copy()).cuda(), torch.from_numpy(W).cuda()
    t_ll = torch.zeros(len(idx), dtype=torch.float64, device='cuda')
    t_g = torch.zeros((len(idx), p.P), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    d.ll_grad_list_dev(t_idx.data_ptr(), len(idx), t_th.data_ptr(), t_W.data_ptr(), t_ll.data_ptr(), t_g.data_ptr())
    d.sync()
    lll, gl = t_ll.cpu().numpy(), t_g.cpu().numpy()
    for j, n in enumerate(idx):
        lr, gr = p.oracle_ll_grad(int(n), int(n) + 1)
        assert np.allclose(lll[j], lr[0], rtol=1e-10, atol=0)
        assert np.allclose(gl[j], gr[0], rtol=1e-9, atol=1e-9 * np.abs(gr).max())
    d.set_time_range(160, 16 * 100 + 5)
    ll, g = d.ll_grad(p.theta[:16], W, 0, 16)
    assert d.info(0, 16)['kernel_version'] == 6
    d.set_option(95, 2)
    ll2, g2 = d.ll_grad(p.theta[:16], W, 0, 16)
    assert np.allclose(ll, ll2, rtol=1e-12, atol=0) and np.allclose(g, g2, rtol=1e-11, atol=1e-11 * np.abs(g2).max())
    d.close()


def test_evaluations_are_reproducible_bit_for_bit():
    """No atomics, fixed summation orders: the same evaluation repeated on one handle, and on a second handle built from the
    same data, returns the same bits -- for every kernel family the dispatcher picks (K-split and no-K-split resident
    tiles, the two-pass kernel with and without helper waves, a narrow shard's block rings, column slices of a wide
    population on the post-block-major grid, whose workgroups the dispatcher hands out in an order that varies)."""
    from theano_pyglm_amd import _lib
    for N, nT, lo, hi in ((32, 12000, 0, 32), (48, 6000, 0, 48), (80, 9000, 0, 80), (128, 9000, 0, 128), (128, 9000, 32, 48),
                          (160, 9000, 0, 160)):
        p = H.Problem(N, nT, H.std_ibasis(), seed=7 * N + hi, w_scale=0.5)
        d = p.device()
        ll, g = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
        for rep in range(3):
            ll2, g2 = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
            assert np.array_equal(ll, ll2) and np.array_equal(g, g2), (N, lo, hi, rep)
        d2 = p.device()
        ll3, g3 = d2.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
        assert np.array_equal(ll, ll3) and np.array_equal(g, g3), (N, lo, hi)
        ll4, _ = d2.ll_grad(p.theta[lo:hi], p.Weff, lo, hi, want_grad=False)
        assert np.array_equal(ll, ll4)
        d.close()
        d2.close()


def test_wide_population_on_resident_tiles():
    """Populations of more than 128 neurons (more than 640 feature columns) on the resident-tile two-pass kernels: equal-width
    column slices, forward-only passes of the first slices adding their currents in the slab (k_fused5<.., 1, 2 / 3>), pass 1 of
    the last slice from the slab, pass 2 on the H and -- for the earlier slices -- the L column parts (k_fused5<.., 2, 0, 1>).
    Against the in-kernel-feature K-split path (PGL_OPT_KERNEL = 2) to 1e-12, the oracle, for whole populations, a neuron
    range, a neuron list (pgl_ll_grad_list_dev), a time range off the tile grid, ll only; dense stimulus columns ride in
    the last slice.  With one feature per neuron (B = 1 rows are too short for the two-pass kernel) the K-split path stays."""
    import torch
    from theano_pyglm_amd import _lib
    for N, nT, Ds in ((130, 3000, 0), (160, 9000, 0), (200, 2000, 7), (256, 2000, 0), (300, 1500, 0)):
        p = H.Problem(N, nT, H.std_ibasis(), seed=N, Dstim=Ds, w_scale=0.5)
        d = p.device()
        assert d.info()['kernel_version'] == 5 and d.info()['resident_feature_bytes'] > 0
        names = _lib.plan_kernels(N, B=p.B, R=p.ibasis.shape[0], Dstim=Ds, nT=nT)
        assert all(n.startswith('k_fused5<') for n in names) and any(n.endswith(('2, 0, 1, 0>', '2, 0, 1, 1>')) for n in names), names
        ll, g = d.ll_grad(p.theta, p.Weff)
        d2 = p.device()
        d2.set_option(_lib.OPT_KERNEL, 2)
        assert d2.info()['kernel_version'] == 2
        ll2, g2 = d2.ll_grad(p.theta, p.Weff)
        assert np.allclose(ll, ll2, rtol=1e-12) and H.rel_err(g, g2) < 1e-11
        llo, go = p.oracle_ll_grad(N - 3, N)
        assert np.allclose(ll[N - 3:], llo, rtol=LL_RTOL) and H.rel_err(g[N - 3:], go) < G_RTOL
        ll_only, _ = d.ll_grad(p.theta, p.Weff, want_grad=False)
        assert np.array_equal(ll_only, ll)
        lo, hi = 17, N - 9
        llr, gr = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
        assert np.allclose(llr, ll[lo:hi], rtol=1e-12) and H.rel_err(gr, g[lo:hi]) < 1e-11
        if N in (130, 160, 300):
            # a last post block of one to four tiles: the grid is post-block-major with one chunk per CU and post block (every
            # CU runs a full block, then a light one); the chunk-major grid (dev option 91) must give the same sums
            d3 = p.device()
            d3.set_option(91, 1)
            assert d3.info()['blocks'] != d.info()['blocks'] or nT < 16 * 256
            ll3, g3 = d3.ll_grad(p.theta, p.Weff)
            assert np.allclose(ll, ll3, rtol=1e-12) and H.rel_err(g, g3) < 1e-11
            d3.close()
        # a neuron list (what the lock-step optimizer launches once neurons have converged)
        idx = np.sort(np.random.RandomState(N).permutation(N)[:37]).astype(np.int32)
        d_idx = torch.from_numpy(idx).cuda()
        d_th = torch.from_numpy(np.ascontiguousarray(p.theta[idx])).cuda()
        d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
        d_ll = torch.zeros(len(idx), dtype=torch.float64, device='cuda')
        d_g = torch.zeros((len(idx), p.P), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        d.ll_grad_list_dev(d_idx.data_ptr(), len(idx), d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
        d.sync()
        assert np.allclose(d_ll.cpu().numpy(), ll[idx], rtol=1e-12) and H.rel_err(d_g.cpu().numpy(), g[idx]) < 1e-11
        d.set_time_range(160, nT - 37)
        d2.set_time_range(160, nT - 37)
        llt, gt = d.ll_grad(p.theta, p.Weff)
        llt2, gt2 = d2.ll_grad(p.theta, p.Weff)
        assert np.allclose(llt, llt2, rtol=1e-12) and H.rel_err(gt, gt2) < 1e-11
        d.close()
        d2.close()
    p = H.Problem(160, 1000, H.std_ibasis()[:, :1], seed=5, w_scale=0.5)        # B = 1: 80-column slices
    d = p.device()
    assert d.info()['kernel_version'] == 2
    ll, g = d.ll_grad(p.theta, p.Weff)
    llo, go = p.oracle_ll_grad(0, 2)
    assert np.allclose(ll[:2], llo, rtol=LL_RTOL) and H.rel_err(g[:2], go) < G_RTOL
    d.close()


def test_narrow_shard_block_ring_kernel():
    """One post tile (a shard of <= 16 neurons: north star's neuron split of C3 at 8 GPUs) against a feature row of 25 .. 40
    k-tiles: k_fused8 -- K split over eight waves, every wave streaming its own slice of block-form images through a private
    LDS ring.  Against the in-kernel-feature K-split path (PGL_OPT_KERNEL = 2) to 1e-12 and the oracle: neuron ranges of 16,
    9 and 1 neurons (padding lanes), a neuron list, a time range off the tile grid (the masked epilogue, short chunks), ll
    only, an exp nonlinearity, low rates (the general regime of the epilogue) and rows of 25, 32 and 40 k-tiles."""
    import torch
    from theano_pyglm_amd import _lib
    for N, nT, nlin, bias in ((128, 5000, 'explinear', 20.0), (100, 3000, 'exp', -3.0), (80, 2000, 'explinear', -1.0),
                              (128, 333, 'explinear', 5.0)):
        p = H.Problem(N, nT, H.std_ibasis(), kind=nlin, seed=N + nT, w_scale=0.3 if nlin == 'explinear' else 0.02,
                      bias_mu=bias, weighted=(N == 100))
        d = p.device()
        d2 = p.device()
        d2.set_option(_lib.OPT_KERNEL, 2)
        for lo, hi in ((32, 48), (N - 9, N), (5, 6)):
            names = _lib.plan_kernels(N, B=p.B, R=p.ibasis.shape[0], nT=nT, n_lo=lo, count=hi - lo)
            assert names == ['k_fused8<5, 8, 0>'], names
            assert d.info(lo, hi)['kernel_version'] == 6 and d2.info(lo, hi)['kernel_version'] == 2
            ll, g = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
            ll2, g2 = d2.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
            assert np.allclose(ll, ll2, rtol=1e-12) and H.rel_err(g, g2) < 1e-11, (N, lo, hi)
            llo, go = p.oracle_ll_grad(lo, min(hi, lo + 2))
            assert np.allclose(ll[:len(llo)], llo, rtol=LL_RTOL) and H.rel_err(g[:len(llo)], go) < G_RTOL
            ll_only, _ = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi, want_grad=False)
            assert np.array_equal(ll_only, ll)
        # a list of 16 neurons in arbitrary order
        idx = np.random.RandomState(N).permutation(N)[:16].astype(np.int32)
        d_idx = torch.from_numpy(idx).cuda()
        d_th = torch.from_numpy(np.ascontiguousarray(p.theta[idx])).cuda()
        d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
        d_ll = torch.zeros(len(idx), dtype=torch.float64, device='cuda')
        d_g = torch.zeros((len(idx), p.P), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        d.ll_grad_list_dev(d_idx.data_ptr(), len(idx), d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
        d.sync()
        llf, gf = d2.ll_grad(p.theta, p.Weff)
        assert np.allclose(d_ll.cpu().numpy(), llf[idx], rtol=1e-12) and H.rel_err(d_g.cpu().numpy(), gf[idx]) < 1e-11
        # a time range off the tile grid
        t0, t1 = 32, nT - 37
        d.set_time_range(t0, t1)
        d2.set_time_range(t0, t1)
        llt, gt = d.ll_grad(p.theta[32:48], p.Weff, 32, 48)
        llt2, gt2 = d2.ll_grad(p.theta[32:48], p.Weff, 32, 48)
        assert np.allclose(llt, llt2, rtol=1e-12) and H.rel_err(gt, gt2) < 1e-11
        d.close()
        d2.close()


def test_narrow_shard_f32_resident_blocks():
    """PGL_OPT_FEATURE_F32 = 2 (opt-in reduced-traffic mode, SURVEY §7 "Precision": f32 storage of features, f64
    accumulation): the one-post-tile kernel streams f32 blocks (half the HBM bytes) and widens them to f64 on the way into
    LDS -- k_fused8<5, 8, 1>.  Every arithmetic operation is f64; only the stored feature is rounded (2^-24 relative), so
    the result sits ~1e-8 from the f64 path -- far inside north star's rtol 1e-5, far outside the 1e-10 of the default path,
    which is why it is never the default.  Ranges, padding lanes, a neuron list, a time range off the tile grid, ll only,
    both nonlinearities; other shapes ignore the option."""
    import torch
    from theano_pyglm_amd import _lib
    worst_ll = worst_g = 0.0
    for N, nT, nlin, bias in ((128, 5000, 'explinear', 20.0), (100, 3000, 'exp', -3.0), (80, 2100, 'explinear', -1.0)):
        p = H.Problem(N, nT, H.std_ibasis(), kind=nlin, seed=N + nT, w_scale=0.3 if nlin == 'explinear' else 0.02,
                      bias_mu=bias, weighted=(N == 100))
        d = p.device()
        d.set_option(_lib.OPT_FEATURE_F32, 2)
        d0 = p.device()
        for lo, hi in ((32, 48), (N - 9, N), (5, 6)):
            names = _lib.plan_kernels(N, B=p.B, R=p.ibasis.shape[0], nT=nT, n_lo=lo, count=hi - lo, opt_f32=2)
            assert names == ['k_fused8<5, 8, 1>'], names
            i32, i64 = d.info(lo, hi), d0.info(lo, hi)
            assert i32['kernel_version'] == 6 and abs(i32['resident_feature_bytes'] * 2 - i64['resident_feature_bytes']) < 1
            ll, g = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
            ll0, g0 = d0.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
            assert not np.array_equal(g, g0)                       # (another path)
            worst_ll = max(worst_ll, float(np.max(np.abs(ll - ll0) / np.abs(ll0))))
            worst_g = max(worst_g, H.rel_err(g, g0))
            assert np.allclose(ll, ll0, rtol=1e-6) and H.rel_err(g, g0) < 1e-6, (N, lo, hi)
            ll_only, _ = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi, want_grad=False)
            assert np.array_equal(ll_only, ll)
        # the whole population (eight post tiles: the two-pass kernel) ignores the option: f64 images, same bits
        llw, gw = d.ll_grad(p.theta, p.Weff)
        llw0, gw0 = d0.ll_grad(p.theta, p.Weff)
        assert np.array_equal(llw, llw0) and np.array_equal(gw, gw0)
        # a list of 16 neurons in arbitrary order; a time range off the tile grid
        idx = np.random.RandomState(N).permutation(N)[:16].astype(np.int32)
        d_idx = torch.from_numpy(idx).cuda()
        d_th = torch.from_numpy(np.ascontiguousarray(p.theta[idx])).cuda()
        d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
        d_ll = torch.zeros(len(idx), dtype=torch.float64, device='cuda')
        d_g = torch.zeros((len(idx), p.P), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        d.ll_grad_list_dev(d_idx.data_ptr(), len(idx), d_th.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
        d.sync()
        assert np.allclose(d_ll.cpu().numpy(), llw0[idx], rtol=1e-6) and H.rel_err(d_g.cpu().numpy(), gw0[idx]) < 1e-6
        d.set_time_range(32, nT - 37)
        d0.set_time_range(32, nT - 37)
        llt, gt = d.ll_grad(p.theta[32:48], p.Weff, 32, 48)
        llt0, gt0 = d0.ll_grad(p.theta[32:48], p.Weff, 32, 48)
        assert np.allclose(llt, llt0, rtol=1e-6) and H.rel_err(gt, gt0) < 1e-6
        d.close()
        d0.close()
    print("f32 resident blocks: worst relative deviation from the f64 path  ll %.1e  grad %.1e" % (worst_ll, worst_g))
    assert worst_ll < 1e-6 and worst_g < 1e-6
