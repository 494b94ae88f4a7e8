"""Launch time of the batched Gibbs inner-ll kernels at the C4 shape against the forced sub-block loop length (dev tool)."""
import sys
import time
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
ws = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
for name, pre in (('pairs', (cols * 37 + 11) % N), ('sweep step', np.full(N, 11))):
    aw = (A * W)[pre, cols]
    for nl in (0, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14, 15):
        dev.set_option(99, nl << 8)
        for _ in range(3):
            dev.gibbs_ll_cols(cols, pre, aw, ws)
        t0 = time.perf_counter()
        for _ in range(20):
            dev.gibbs_ll_cols(cols, pre, aw, ws)
        print("%-10s nloop %2d: %.3f ms per launch" % (name, nl, (time.perf_counter() - t0) / 20 * 1e3))
    dev.set_option(99, 0)
