"""Where a batched collapsed-Gibbs sweep at the C4 shape spends its time (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.inference import gibbs as G
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nT = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
model = make_model('sparse_weighted_model', N=N, dt=0.001)
stabilize_sparsity(model)
popn = Population(model)
rng = np.random.default_rng(1238)
S = np.minimum(rng.poisson(20.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
popn.add_data({'S': S, 'N': N, 'dt': 0.001, 'T': nT * 0.001, 'stim': None, 'dt_stim': 0.1})
x = popn.sample(np.random.RandomState(4))
x['net']['weights']['W'] = 0.2 * np.asarray(x['net']['weights']['W'])
upd = G.CollapsedGibbsNetworkColumnUpdate(rng=np.random.RandomState(2))
upd.preprocess(popn)
h = popn._handle(popn._current)
A = np.asarray(x['net']['graph']['A']).reshape(N, N); W = np.asarray(x['net']['weights']['W']).reshape(N, N)
t0 = time.time(); th = popn.theta_matrix(x); t_theta = time.time() - t0
t0 = time.time(); h.gibbs_prepare_all(th, A * W); h.sync(); t_prep = time.time() - t0
t0 = time.time(); h.gibbs_prepare_all(th, A * W); h.sync(); t_prep2 = time.time() - t0
cols = np.arange(N); pre = (cols * 37 + 11) % N
ws = np.tile(np.linspace(-4, 4, 11), (N, 1)); aw = (A * W)[pre, cols]
h.gibbs_ll_cols(cols, pre, aw, ws)
t0 = time.time()
for _ in range(20): h.gibbs_ll_cols(cols, pre, aw, ws)
t_ll = (time.time() - t0) / 20
for nm, wz in (('all weights 0 (series regime everywhere)', np.zeros((N, 11))), ('positive nodes only', np.abs(ws)),
               ('K=1', ws[:, :1])):
    h.gibbs_ll_cols(cols, pre, aw, wz)
    t0 = time.time()
    for _ in range(10): h.gibbs_ll_cols(cols, pre, aw, wz)
    print("  ll_cols, %s: %.3f ms" % (nm, (time.time() - t0) / 10 * 1e3))
t0 = time.time()
for _ in range(50): h.gibbs_ll_cols(cols[:1], pre[:1], aw[:1], ws[:1, :1])
t_one = (time.time() - t0) / 50
t0 = time.time()
for _ in range(50): h.gibbs_ll_cols(cols[:1], pre[:1], aw[:1], np.linspace(-0.2, 0.2, 14)[None, :])
print("  ll_cols(1 col x 14): %.3f ms" % ((time.time() - t0) / 50 * 1e3))
t0 = time.time()
for _ in range(20): h.gibbs_update_cols(cols[:3], pre[:3], np.array([0.1, -0.1, 0.2]))
h.sync(); t_upd = (time.time() - t0) / 20
print("theta_matrix %.1f ms | prepare_all first %.1f ms, again %.1f ms | ll_cols(%d cols x 11) %.3f ms | "
      "ll_cols(1 col x 1) %.3f ms | update_cols(3) %.3f ms" % (t_theta * 1e3, t_prep * 1e3, t_prep2 * 1e3, N, t_ll * 1e3, t_one * 1e3, t_upd * 1e3))
acc = {'wide': 0.0, 'narrow': 0.0, 'update': 0.0, 'n_narrow': 0}
_ll, _up = h.gibbs_ll_cols, h.gibbs_update_cols


def ll_timed(c, p_, a, w):
    t = time.perf_counter()
    r = _ll(c, p_, a, w)
    k = 'wide' if len(c) > 4 else 'narrow'
    acc[k] += time.perf_counter() - t
    acc['n_narrow'] += k == 'narrow'
    return r


def up_timed(*a):
    t = time.perf_counter()
    r = _up(*a)
    acc['update'] += time.perf_counter() - t
    return r


h.gibbs_ll_cols, h.gibbs_update_cols = ll_timed, up_timed
for rep in range(2):
    upd.n_ars_evals = 0
    for k in acc: acc[k] = 0
    t0 = time.time(); upd.update_all(x); t_sw = time.time() - t0
    print("sweep %d: %.3f s (wide launches %.3f s, %d narrow launches %.3f s, current updates %.3f s, host rest %.3f s), "
          "ARS evals %d, edges %d" % (rep, t_sw, acc['wide'], acc['n_narrow'], acc['narrow'], acc['update'],
                                      t_sw - acc['wide'] - acc['narrow'] - acc['update'], upd.n_ars_evals,
                                      int(np.asarray(x['net']['graph']['A']).sum())))
if '--cprofile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); upd.update_all(x); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)
