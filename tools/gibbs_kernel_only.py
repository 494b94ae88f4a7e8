"""A few launches of the batched Gibbs inner-ll kernels at the C4 shape (for rocprofv3; dev tool)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
N, nT = 128, 600000
model = make_model('sparse_weighted_model', N=N, dt=0.001)
stabilize_sparsity(model)
popn = Population(model)
rng = np.random.default_rng(1238)
S = np.minimum(rng.poisson(20.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
popn.add_data({'S': S, 'N': N, 'dt': 0.001, 'T': nT * 0.001, 'stim': None, 'dt_stim': 0.1})
x = popn.sample(np.random.RandomState(4))
x['net']['weights']['W'] = 0.2 * np.asarray(x['net']['weights']['W'])
dev = popn._handle(popn._current)
A = np.asarray(x['net']['graph']['A']).reshape(N, N)
W = np.asarray(x['net']['weights']['W']).reshape(N, N)
dev.gibbs_prepare_all(popn.theta_matrix(x), A * W)
cols = np.arange(N)
pre = (cols * 37 + 11) % N
ws = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
aw = (A * W)[pre, cols]
import time
from theano_pyglm_amd import _lib
res = {}
for opt in (1, 0, 1, 0):
    dev.set_option(_lib.OPT_GIBBS_KERNEL, opt)
    for _ in range(3):
        ll = dev.gibbs_ll_cols(cols, pre, aw, ws)
    t0 = time.perf_counter()
    for _ in range(20):
        ll = dev.gibbs_ll_cols(cols, pre, aw, ws)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    res[opt] = ll
    print("PGL_OPT_GIBBS_KERNEL=%d: %.3f ms per launch of %d pairs x %d weights, finite %.3f" % (opt, ms, N, ws.shape[1], np.isfinite(ll).mean()))
pre1 = np.full(N, 11)
aw1 = (A * W)[pre1, cols]
for _ in range(3):
    ll = dev.gibbs_ll_cols(cols, pre1, aw1, ws)
t0 = time.perf_counter()
for _ in range(20):
    ll = dev.gibbs_ll_cols(cols, pre1, aw1, ws)
print("one presynaptic neuron for all columns (a sweep step): %.3f ms per launch" % ((time.perf_counter() - t0) / 20 * 1e3))
dev.set_option(99, 0x1000)
for _ in range(3):
    ll2 = dev.gibbs_ll_cols(cols, pre1, aw1, ws)
t0 = time.perf_counter()
for _ in range(20):
    ll2 = dev.gibbs_ll_cols(cols, pre1, aw1, ws)
print("   the same launch through the event loop: %.3f ms; max rel diff %.2e" % ((time.perf_counter() - t0) / 20 * 1e3, np.nanmax(np.abs(ll - ll2) / np.abs(ll2))))
dev.set_option(99, 0)
fin = np.isfinite(res[0]) & np.isfinite(res[1])
print("max rel diff regime-split vs all-f64: %.2e" % np.max(np.abs(res[0][fin] - res[1][fin]) / np.abs(res[1][fin])))
