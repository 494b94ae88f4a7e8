"""Ablation of the regime-split batched Gibbs kernel at the C4 shape (dev tool): option 99 bits 1 = no pair-current
loop, 2 = no evaluation, 4 = no event staging, 8 = no current loads."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.inference import gibbs as G
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
A = np.asarray(x['net']['graph']['A']).reshape(N, N); W = np.asarray(x['net']['weights']['W']).reshape(N, N)
dev.gibbs_prepare_all(popn.theta_matrix(x), A * W)
print('R', popn.glm.imp_model.ibasis.shape)
cols = np.arange(N); pre = (cols * 37 + 11) % N
ws11 = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
aw = (A * W)[pre, cols]
for dbg in (0, 1, 2, 4, 8, 12, 15, 16, 32, 64, 16 + 32 + 64, 1 + 16 + 32 + 64):
    dev.set_option(99, dbg)
    for K in (11,):
        ws = ws11[:, :K].copy()
        for _ in range(3): dev.gibbs_ll_cols(cols, pre, aw, ws)
        t0 = time.perf_counter()
        for _ in range(10): dev.gibbs_ll_cols(cols, pre, aw, ws)
        print("dbg %2d K %2d: %.3f ms per call" % (dbg, K, (time.perf_counter() - t0) * 100))
