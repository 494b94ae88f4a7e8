"""Distribution of the currents x_k(t) = x0(t) + w_k ic(t) of the collapsed-Gibbs inner ll at the C4 shape over the
10 Gauss-Hermite nodes + w = 0 (which regime do the evaluations of k_gibbs_rate_cols fall into?).  Dev tool."""
import sys
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
from oracle import glm_oracle as O
N, nT = 128, 120000
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
nodes = np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0]))
edges = np.array([0, 2, 4, 6, 8, 10, 12, 20, 700, np.inf])
hist = np.zeros(len(edges) - 1)
neg = 0
tot = 0
imp = popn.glm.imp_model
for c in (0, 17, 55, 101):
    xcur = dev.gibbs_currents(c) + x['glms'][c]['bias']['bias'][0]
    for pre in ((c * 37 + 11) % N, (c + 1) % N):
        beta = imp.flat_weights(x['glms'][c]['imp']).reshape(N, -1)[pre]
        h = imp.ibasis.dot(beta)
        ic = O.convolve_with_basis_fft(S[:, pre:pre + 1].astype(float), h[:, None])[:, 0, 0]
        x0 = xcur - (A * W)[pre, c] * ic
        for w in nodes:
            xk = x0 + w * ic
            hist += np.histogram(np.abs(xk), bins=edges)[0]
            neg += np.sum(xk <= -12)
            tot += xk.size
print("fraction of evaluations by |x|:")
for a, b, h_ in zip(edges[:-1], edges[1:], hist):
    print("  [%g, %g): %.4f" % (a, b, h_ / tot))
print("x <= -12: %.4f" % (neg / tot))
