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
for _ in range(6):
    ll = dev.gibbs_ll_cols(cols, pre, aw, ws)
print(np.isfinite(ll).mean())
