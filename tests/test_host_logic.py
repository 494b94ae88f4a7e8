"""CPU tests of the host-side mirror (no compute on the device)."""
import json
import os

import numpy as np
import pytest

from oracle import glm_oracle as O
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity, check_stability
from theano_pyglm_amd.models import templates as T
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.utils import basis as B
from theano_pyglm_amd.utils import packvec as PV
from theano_pyglm_amd.utils.syms import flatten

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_templates_equal_reference():
    with open(os.path.join(ROOT, 'tests', 'golden', 'templates_golden.json')) as f:
        g = json.load(f)
    assert T.standard_glm() == g['standard_glm']
    assert T.sparse_weighted_model() == g['sparse_weighted_model']
    assert T.spatiotemporal_glm() == g['spatiotemporal_glm']


def test_make_model_and_sparsity():
    m = make_model('standard_glm', N=7, dt=0.001)
    assert m['N'] == 7 and m['dt'] == 0.001 and T.standard_glm()['N'] == 2
    with pytest.raises(Exception):
        make_model('no_such_model')
    s = make_model('sparse_weighted_model', N=128, dt=0.001)
    stabilize_sparsity(s)
    assert abs(s['network']['graph']['rho'] - min(1.0, (0.7 + 0.2) ** 2 / 128.0)) < 1e-15   # model_factory.py:87-102
    assert make_model('network_glm', N=3)['impulse']['type'] == 'dirichlet'
    with pytest.raises(KeyError):
        Population(make_model('standard_glm', N=2))          # no dt -> glm.py:16 KeyError


def test_basis_module_matches_golden(golden):
    std = T.standard_glm()['impulse']['basis']
    assert np.allclose(B.create_basis(std), golden['std_imp_basis'], atol=1e-13)
    assert np.allclose(B.create_basis(T.sparse_weighted_model()['impulse']['basis']), golden['swm_imp_basis'], atol=1e-13)
    assert np.allclose(B.convolve_with_basis(golden['conv_S'], golden['conv_ibasis']), golden['conv_fS'], atol=1e-12)
    assert np.allclose(B.convolve_with_low_rank_2d_basis(golden['lr2d_stim'], golden['lr2d_ibasis_x'],
                                                         golden['lr2d_ibasis_t']), golden['lr2d_fstim'], atol=1e-12)
    f = np.sin(np.linspace(0, 3, 200))
    beta = B.project_onto_basis(f, golden['conv_ibasis'])
    assert beta.shape == (5, 1)
    assert np.allclose(golden['conv_ibasis'].T.dot(golden['conv_ibasis'].dot(beta)[:, 0] - f), 0, atol=1e-9)


def test_component_bases_match_oracle(golden):
    p = Population(make_model('standard_glm', N=3, dt=0.001))
    assert np.allclose(p.glm.imp_model.ibasis, O.linear_impulse_ibasis(golden['std_imp_basis'], 0.001, 0.2, False))
    q = Population(make_model('sparse_weighted_model', N=3, dt=0.001))
    assert np.allclose(q.glm.imp_model.ibasis, O.dirichlet_impulse_ibasis(golden['swm_imp_basis'], 0.001, 0.2, True))
    r = Population(make_model('spatiotemporal_glm', N=3, dt=0.001))
    assert np.allclose(r.glm.imp_model.ibasis, O.linear_impulse_ibasis(golden['st_imp_basis'], 0.001, 0.3, True))
    assert np.allclose(r.glm.bkgd_model.ibasis_t, O.stim_temporal_ibasis(golden['st_temporal_basis'], 0.001, 0.3, True))
    assert np.allclose(r.glm.bkgd_model.ibasis_x, np.eye(3))


def test_state_dict_schema_and_packing():
    """population.py:149-162 / SURVEY Appendix A; packed layout SURVEY §8a A7."""
    rng = np.random.RandomState(0)
    p = Population(make_model('standard_glm', N=4, dt=0.001))
    x = p.sample(rng)
    assert sorted(x.keys()) == ['glms', 'latent', 'net'] and x['net'] == {'graph': {}, 'weights': {}}
    assert [g['n'] for g in x['glms']] == [0, 1, 2, 3]
    assert x['glms'][0]['imp']['w_ir'].shape == (20,) and x['glms'][0]['bias']['bias'].shape == (1,)
    syms = p.glm_syms()
    v, shapes = PV.packdict(PV.get_vars(syms, x['glms'][1]))
    assert v.size == 21 and v[0] == x['glms'][1]['bias']['bias'][0]
    assert np.array_equal(v[1:], x['glms'][1]['imp']['w_ir'])
    back = PV.unpackdict(v, shapes)
    assert np.array_equal(back['imp']['w_ir'], x['glms'][1]['imp']['w_ir'])
    assert [s.name for s in flatten(syms)] == ['bias', 'w_ir']
    nv = p.extract_vars(x, 2)
    assert sorted(nv.keys()) == ['glm', 'latent', 'net'] and nv['glm']['n'] == 2
    # missing key -> the seval binding error (theano_func_wrapper.py:92-93)
    bad = p.sample(rng)
    del bad['glms'][0]['imp']['w_ir']
    with pytest.raises(Exception, match="not found"):
        p.compute_log_prior(bad)
    # sparse model: A int8, W flat, Dirichlet g_n
    q = Population(stabilize_sparsity(make_model('sparse_weighted_model', N=3, dt=0.001)))
    y = q.sample(rng)
    assert y['net']['graph']['A'].dtype == np.int8 and y['net']['weights']['W'].shape == (9,)
    assert sorted(y['glms'][0]['imp'].keys()) == ['g_0', 'g_1', 'g_2']
    assert check_stability(q.model, y, 3) in (True, False)
    st = Population(make_model('spatiotemporal_glm', N=2, dt=0.001))
    z = st.sample(rng)
    v, _ = PV.packdict(PV.get_vars(st.glm_syms(), z['glms'][0]))
    assert v.size == 1 + 3 + 3 + 6
    assert np.array_equal(v[1:4], z['glms'][0]['bkgd']['w_t']) and np.array_equal(v[4:7], z['glms'][0]['bkgd']['w_x'])


def test_log_prior_matches_oracle():
    rng = np.random.RandomState(1)
    p = Population(make_model('standard_glm', N=3, dt=0.001))
    x = p.sample(rng)
    ref = sum(O.bias_log_p(g['bias']['bias'][0], 20, 0.1) +
              O.group_lasso_log_p(g['imp']['w_ir'].reshape(3, 5), 1.0, 0.0, 10.0) for g in x['glms'])
    assert np.isclose(p.compute_log_prior(x), ref)
    q = Population(stabilize_sparsity(make_model('sparse_weighted_model', N=3, dt=0.001)))
    y = q.sample(rng)
    W = y['net']['weights']['W'].reshape(3, 3)
    ref = O.erdos_renyi_log_p(y['net']['graph']['A'].astype(float), q.network.graph.rho) + \
        O.gaussian_weight_log_p(W, 0.0, 1.0, -0.2, 0.5)
    for g in y['glms']:
        ref += O.bias_log_p(g['bias']['bias'][0], 20.0, 0.25) + \
            O.dirichlet_log_p([g['imp']['g_%d' % k] for k in range(3)], 1)
    assert np.isclose(q.compute_log_prior(y), ref)


def test_population_fast_paths_equal_the_per_neuron_forms():
    """compute_log_prior evaluates every component over all neurons at once, theta_matrix fills the rows by column blocks,
    the binding check walks precompiled key paths, shape_vars replaces the sample the reference draws for its packing
    shapes (coord_descent.py:24-26, 93-95): same values as the per-neuron forms, same error for an unbound variable."""
    import copy
    from theano_pyglm_amd.inference.coord_descent import get_vars
    from theano_pyglm_amd.utils.syms import differentiable
    for name, N in (('standard_glm', 9), ('spatiotemporal_glm', 5), ('sparse_weighted_model', 6)):
        popn = Population(make_model(name, N=N, dt=0.001))
        x = popn.sample(np.random.RandomState(N))
        ref = popn.latent.log_p(x.get('latent', {})) + popn.network.log_p(x['net'])
        for n in range(N):
            ref += popn.glm.log_prior(x['glms'][n])
        assert np.isclose(popn.compute_log_prior(x), ref, rtol=1e-13, atol=0)
        rows = np.array([popn.glm.theta_row(x['glms'][n]) for n in range(N)])
        assert np.array_equal(popn.theta_matrix(x, 0, N), rows) and np.array_equal(popn.theta_matrix(x, 2, 4), rows[2:4])
        z = popn.shape_vars()
        gs, ns = popn.glm_syms(), differentiable(popn.get_variables()['net'])
        assert PV.packdict(get_vars(gs, popn.extract_vars(z, 0)['glm']))[1] == PV.packdict(get_vars(gs, popn.extract_vars(x, 0)['glm']))[1]
        assert PV.packdict(get_vars(ns, z['net']))[1] == PV.packdict(get_vars(ns, x['net']))[1]
        popn.compute_log_prior(z)                                  # a fully bound dictionary
        for path in (('glms', N - 1, 'bias', 'bias'), ('glms', 0, 'imp'), ('net',), ('glms', 1, 'nlin')):
            y = copy.deepcopy(x)
            v = y
            for k in path[:-1]:
                v = v[k]
            del v[path[-1]]
            with pytest.raises(Exception, match="Key %s not found in either vals or defaults!" % path[-1]):
                popn.compute_log_prior(y)
        y = copy.deepcopy(x)
        y['glms'][2]['imp'] = None
        with pytest.raises(Exception, match="Key imp not found"):
            popn.compute_log_prior(y)


def test_flat_weights_and_chain_rules():
    rng = np.random.RandomState(2)
    q = Population(stabilize_sparsity(make_model('sparse_weighted_model', N=3, dt=0.001)))
    y = q.sample(rng)
    xn = y['glms'][1]
    xn['imp']['g_1'] = xn['imp']['g_1'] * np.array([1, -1, 1, 1, -1])
    th = q.glm.theta_row(xn)
    assert th.size == 16 and np.allclose(th[1:].reshape(3, 5).sum(1), 1.0)
    gth = rng.randn(16)
    gd = q.glm.chain_grad(xn, gth)
    for k in range(3):
        assert np.allclose(gd['imp']['g_%d' % k], O.dirichlet_chain(gth[1 + 5 * k:6 + 5 * k], xn['imp']['g_%d' % k]))
    st = Population(make_model('spatiotemporal_glm', N=2, dt=0.001))
    z = st.sample(rng)['glms'][0]
    th = st.glm.theta_row(z)
    assert np.allclose(th[1:10], O.spatiotemporal_w_stim(z['bkgd']['w_t'], z['bkgd']['w_x']))
    gth = rng.randn(th.size)
    gd = st.glm.chain_grad(z, gth)
    gt, gx = O.spatiotemporal_chain(gth[1:10], z['bkgd']['w_t'], z['bkgd']['w_x'])
    assert np.allclose(gd['bkgd']['w_t'], gt) and np.allclose(gd['bkgd']['w_x'], gx)
    # W_eff orientation [n_pre, n_post] = A*W (glm.py:31-37)
    assert np.allclose(q.W_eff(y), y['net']['graph']['A'] * y['net']['weights']['W'].reshape(3, 3))


def test_stimulus_host_twin_matches_oracle():
    """The numpy twin of the device stimulus-feature build (used by the host simulator)."""
    rng = np.random.RandomState(3)
    st = Population(make_model('spatiotemporal_glm', N=2, dt=0.001))
    stim = rng.randn(20, 3)
    data = {'S': np.zeros((2000, 2)), 'stim': stim, 'dt_stim': 0.1, 'T': 2.0, 'dt': 0.001, 'N': 2}
    st.preprocess_data(data)
    f = st.glm.bkgd_model.host_features(data, 2000)
    ref = O.spatiotemporal_stim_features(stim, 0.1, 0.001, 2000, st.glm.bkgd_model.ibasis_x,
                                         st.glm.bkgd_model.ibasis_t)
    assert f.shape == (2000, 9) and np.allclose(f, ref, atol=1e-12)
    m = make_model('standard_glm', N=2, dt=0.001)
    m['bkgd']['type'] = 'basis'
    bs = Population(m)
    d2 = {'S': np.zeros((2000, 2)), 'stim': stim[:, :1], 'dt_stim': 0.1, 'T': 2.0, 'dt': 0.001, 'N': 2}
    bs.preprocess_data(d2)
    ref = O.basis_stim_features(stim[:, :1], 0.1, 0.001, 2000, bs.glm.bkgd_model.ibasis)
    assert np.allclose(bs.glm.bkgd_model.host_features(d2, 2000), ref, atol=1e-12)
    with pytest.raises(Exception, match="Stim dimension"):
        bs.preprocess_data({'S': np.zeros((2000, 2)), 'stim': stim, 'dt_stim': 0.1, 'T': 2.0})


def test_native_simulate_matches_python_loop():
    """pgl_simulate (C++) consumes the uniform stream in the reference's draw order: same spikes
    as the line-by-line Python restatement of population.py:291-364."""
    from theano_pyglm_amd import _lib
    for name, N in (('standard_glm', 3), ('sparse_weighted_model', 4)):
        m = stabilize_sparsity(make_model(name, N=N, dt=0.001))
        p = Population(m)
        x = p.sample(np.random.RandomState(5))
        S_py, X_py = p.simulate(x, (0, 1.2), 0.001, None, 0.1, rng=np.random.RandomState(6), native=False)
        # same stream for the native loop
        nT = 1200
        X0 = np.zeros((nT, N))
        for n in range(N):
            X0[:, n] = x['glms'][n]['bias']['bias'][0]
        imps = np.array([p.glm.imp_model.impulse(x['glms'][n]['imp']) for n in range(N)])
        AW = p.W_eff(x)[:, :, None] * np.transpose(imps, axes=[1, 0, 2])      # (pre, post, R)
        u = np.random.RandomState(6).random_sample(200000)
        S_c, X_c, nexc = _lib.simulate(X0, np.ascontiguousarray(np.transpose(AW, (0, 2, 1))),
                                       p.glm.nlin_model.kind, 0.001, uniforms=u, seed=1)
        assert S_py.sum() > 10
        assert np.array_equal(S_py, S_c)
        assert np.allclose(X_py, X_c, rtol=1e-10, atol=1e-10)
        # and through Population.simulate(native=True): the reference's consistency invariant
        S_n, X_n = p.simulate(x, (0, 1.2), 0.001, None, 0.1, rng=np.random.RandomState(7))
        for n in range(N):
            xd = O.direct_currents(S_n, p.glm.imp_model.impulse(x['glms'][n]['imp']), p.W_eff(x)[:, n],
                                   x['glms'][n]['bias']['bias'][0])
            assert np.allclose(X_n[:, n], xd)


def test_simulate_consistency_host():
    """Population.simulate reproduces the superposition the oracle restates
    (population.py:351-353): X equals direct_currents of the spikes it emitted."""
    rng = np.random.RandomState(4)
    p = Population(make_model('standard_glm', N=3, dt=0.001))
    x = p.sample(rng)
    S, X = p.simulate(x, (0, 1.5), 0.001, None, 0.1, rng=rng, native=False)
    assert S.shape == (1500, 3) and S.max() <= 10
    for n in range(3):
        imps = p.glm.imp_model.impulse(x['glms'][n]['imp'])
        xd = O.direct_currents(S, imps, np.ones(3), x['glms'][n]['bias']['bias'][0])
        assert np.allclose(X[:, n], xd)


def test_data_io_roundtrip_and_segment(tmp_path):
    """utils/io.py: the data-dict schema survives .pkl and .mat round trips (io.py:82-124) and
    segment_data cuts spikes on the dt grid and the stimulus on the dt_stim grid (io.py:126-149)."""
    from theano_pyglm_amd.utils import io
    rng = np.random.RandomState(0)
    data = {'S': (rng.rand(4000, 3) < 0.02).astype(float), 'N': 3, 'dt': 0.001, 'T': 4.0,
            'stim': rng.randn(40, 2), 'dt_stim': 0.1, 'vars': {'glms': [{'bias': {'bias': np.array([1.0])}}]},
            'preprocessed': True, '_device_handle': object()}
    for ext in ('pkl', 'mat'):
        p = str(tmp_path / ('d.' + ext))
        io.save_data(data, p)
        d2 = io.load_data(p)
        assert isinstance(d2['N'], int) and d2['N'] == 3 and d2['T'] == 4.0 and d2['dt'] == 0.001
        assert np.array_equal(d2['S'], data['S']) and np.array_equal(d2['stim'], data['stim'])
        assert '_device_handle' not in d2 and 'preprocessed' not in d2
    assert io.load_data(str(tmp_path / 'd.pkl'))['vars']['glms'][0]['bias']['bias'][0] == 1.0
    seg = io.segment_data(data, (1.0, 2.5))
    assert seg['T'] == 1.5 and abs(seg['S'].shape[0] - 1500) <= 1 and abs(seg['stim'].shape[0] - 15) <= 1
    # the reference indexes with float floor division (io.py:141-147): 1.0 // 0.001 == 999.0
    i0, i1, j0, j1 = int(1.0 // 0.001), int(2.5 // 0.001), int(1.0 // 0.1), int(2.5 // 0.1)
    assert np.array_equal(seg['S'], data['S'][i0:i1]) and np.array_equal(seg['stim'], data['stim'][j0:j1])
    assert 'preprocessed' not in seg and data['S'].shape == (4000, 3)
    with pytest.raises(AssertionError):
        io.segment_data(data, (3.0, 5.0))
    with pytest.raises(Exception):
        io.load_data(str(tmp_path / 'd.txt'))


def test_log_sum_exp_sample_matches_reference_golden():
    """inference/log_sum_exp.log_sum_exp_sample against choices drawn by the reference's own
    pyglm/inference/log_sum_exp.py (tests/golden/make_golden_lse.py): same seed -> same uniform -> same
    category, including -inf entries and log probabilities far below the exp range."""
    import os
    from theano_pyglm_amd.inference.log_sum_exp import log_sum_exp_sample
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lse_golden.npz'))
    assert len(g['choice']) == 60
    for lnp, n, seed, choice in zip(g['lnp'], g['n'], g['seed'], g['choice']):
        got = log_sum_exp_sample(lnp[:n], rng=np.random.RandomState(int(seed)))
        assert got == int(choice)
    with pytest.raises(Exception):
        log_sum_exp_sample(np.array([-np.inf, -np.inf]))


def test_model_factory_matches_reference_golden():
    """make_model + stabilize_sparsity, check_stability and convert_model (basis -> dirichlet) against
    outputs of the reference's own pyglm/models/model_factory.py (tests/golden/make_golden_models.py)."""
    import copy
    import os
    from theano_pyglm_amd.models.model_factory import convert_model
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    with open(os.path.join(here, 'models_golden.json')) as f:
        gj = json.load(f)
    g = np.load(os.path.join(here, 'models_golden.npz'))

    def close(a, b):
        if isinstance(a, dict):
            assert set(a) == set(b), (sorted(a), sorted(b))
            for k in a:
                close(a[k], b[k])
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                close(x, y)
        elif isinstance(a, float) or isinstance(b, float):
            assert abs(float(a) - float(b)) <= 1e-15 * max(1.0, abs(float(b))), (a, b)
        else:
            assert a == b, (a, b)

    for key, ref_model in gj['models'].items():
        name, N = key.split('/')
        m = stabilize_sparsity(make_model(name, N=int(N), dt=0.001))
        close(json.loads(json.dumps(m)), ref_model)
    model = make_model('sparse_weighted_model', N=5, dt=0.001)
    for A, W, ok in zip(g['stab_A'], g['stab_W'], gj['stability']):
        assert check_stability(model, {'net': {'graph': {'A': A}, 'weights': {'W': W}}}, 5) == ok
    # convert_model: same inputs (impulse responses of a sampled standard_glm state, target basis)
    N = 3
    from_model = make_model('standard_glm', N=N, dt=0.001)
    to_model = make_model('sparse_weighted_model', N=N, dt=0.001)
    from_popn, to_popn = Population(from_model), Population(to_model)
    from_vars = from_popn.sample(np.random.RandomState(7))
    to_vars = to_popn.sample(np.random.RandomState(8))
    assert np.array_equal(np.array([from_vars['glms'][n]['imp']['w_ir'] for n in range(N)]), g['from_w_ir'])
    assert np.allclose(to_popn.glm.imp_model.ibasis, g['conv_basis'], rtol=0, atol=0)
    imp = np.array([from_popn.glm.imp_model.impulse(from_vars['glms'][n]['imp']) for n in range(N)])
    assert np.array_equal(imp, g['conv_impulses'])
    tm = copy.deepcopy(to_model)
    tm['network']['graph'].pop('rho', None)
    conv = convert_model(from_popn, from_model, from_vars, to_popn, tm, to_vars)
    gg = np.array([[conv['glms'][n2]['imp']['g_%d' % n1] for n1 in range(N)] for n2 in range(N)])
    assert np.allclose(gg, g['conv_g'], rtol=1e-12, atol=0)
    assert np.allclose(conv['net']['weights']['W'], g['conv_W'], rtol=1e-12, atol=0)
    assert np.array_equal(conv['net']['graph']['A'], g['conv_A'])
    assert np.array_equal(np.array([conv['glms'][n]['bias']['bias'] for n in range(N)]), g['conv_bias'])


def test_packed_layout_matches_reference_golden():
    from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars
    """The packed per-neuron parameter layout (packdict / unpackdict / get_vars / set_vars; SURVEY §8a
    A7) against the reference's own pyglm/utils/packvec.py (tests/golden/make_golden_packvec.py):
    entry i of the packed vector belongs to the same variable as in the reference."""
    import copy
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'packvec_golden.json')) as f:
        g = json.load(f)

    def lists(d):
        return dict((k, lists(v) if isinstance(v, dict) else np.asarray(v).tolist()) for k, v in d.items())

    for name, case in g.items():
        popn = Population(make_model(name, N=case['N'], dt=0.001))
        x = popn.sample(np.random.RandomState(1))
        syms = popn.glm_syms()
        xv = get_vars(syms, x['glms'][1])
        v, shapes = packdict(xv)
        assert v.size == case['size']
        assert lists(shapes) == case['shapes']
        un = unpackdict(np.arange(float(v.size)), shapes)
        assert lists(un) == case['unpacked']
        # packdict is the inverse of the reference's unpackdict: packing the unpacked arange returns arange
        assert np.array_equal(packdict(un)[0], np.arange(float(v.size)))
        target = copy.deepcopy(x['glms'][1])
        set_vars(syms, target, un)
        assert lists(dict((k, t) for k, t in target.items() if isinstance(t, dict))) == case['after_set_vars']
    # the documented layouts
    assert list(g['standard_glm']['unpacked']) == ['bias', 'imp'] or set(g['standard_glm']['unpacked']) >= {'bias', 'imp'}
    assert g['standard_glm']['unpacked']['bias']['bias'] == [0.0]
    st = g['spatiotemporal_glm']['unpacked']
    assert st['bkgd']['w_t'] == [1.0, 2.0, 3.0] and st['bkgd']['w_x'] == [4.0, 5.0, 6.0]      # 'w_t' < 'w_x'


def test_block_image_swizzle_is_conflict_free_for_both_mfma_operand_patterns():
    """The block-form feature images of k_fused8 (csrc/pglm_fused_resident.hip.h: pgl_blk_off) store element (time row t, column c)
    of a 16 x 16 f64 block at t * 16 + (c ^ (2 * (t >> 1))).  A ds_read_b64 serves 32 lanes per pass over 64 banks of
    4 bytes, i.e. 32 slots of 8 bytes: both MFMA operand patterns -- forward A (lane = time row i, k-step columns 4 ks +
    grp) and backward A = F^T (lane = column i, time rows 4 q + grp) -- must put the 32 lanes of each pass into 32
    different slots, without the row padding the other kernels' images use (k_build_fimg writes the same permutation)."""
    def off(t, c):
        return t * 16 + (c ^ (2 * (t >> 1)))

    # a permutation of the block
    assert sorted(off(t, c) for t in range(16) for c in range(16)) == list(range(256))
    for half in (0, 1):                                   # lanes 0-31 / 32-63: grp in {0, 1} / {2, 3}
        for s in range(4):                                # k-step (forward) resp. time quarter (backward)
            fwd = {off(i, 4 * s + grp) % 32 for i in range(16) for grp in (2 * half, 2 * half + 1)}
            bwd = {off(4 * s + grp, i) % 32 for i in range(16) for grp in (2 * half, 2 * half + 1)}
            assert len(fwd) == 32 and len(bwd) == 32, (half, s)
