"""Run time of the batched Gibbs inner-ll launch by column group and by single column at the C4 shape: how uneven is
the work of the workgroups / waves of k_gibbs_rate_cols?  (dev tool)"""
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
theta = popn.theta_matrix(x)
dev.gibbs_prepare_all(theta, A * W)
ws = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))


def t_launch(cols, pre, reps=10):
    aw = (A * W)[pre, cols]
    for _ in range(2):
        dev.gibbs_ll_cols(cols, pre, aw, ws[:len(cols)])
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.gibbs_ll_cols(cols, pre, aw, ws[:len(cols)])
    return (time.perf_counter() - t0) / reps * 1e3


allc = np.arange(N)
print("all 128 columns: %.3f ms" % t_launch(allc, np.full(N, 11)))
tg = []
for g in range(16):
    cols = np.arange(8 * g, 8 * g + 8)
    tg.append(t_launch(cols, np.full(8, 11)))
print("by group of 8 (ms):", " ".join("%.3f" % t for t in tg), "| sum %.3f max/mean %.2f" % (sum(tg), max(tg) / np.mean(tg)))
t1 = np.array([t_launch(np.array([c]), np.array([11]), reps=5) for c in range(N)])
print("single columns (ms): min %.3f median %.3f max %.3f; by group max/mean of the 8: %s" % (
    t1.min(), np.median(t1), t1.max(), " ".join("%.2f" % (t1[8 * g:8 * g + 8].max() / t1[8 * g:8 * g + 8].mean()) for g in range(16))))
print("bias of the neurons (theta[:, 0]): min %.2f median %.2f max %.2f; corr(time, -|bias|) %.2f" % (
    theta[:, 0].min(), np.median(theta[:, 0]), theta[:, 0].max(), np.corrcoef(t1, -np.abs(theta[:, 0]))[0, 1]))
order = np.argsort(-t1)
# heavy and light columns paired in every group of 8 / in every wave's two items
perm = np.empty(N, dtype=int)
half = N // 2
inter = np.empty(N, dtype=int)
inter[0::2] = order[:half]
inter[1::2] = order[::-1][:half]
print("columns regrouped (heavy next to light): %.3f ms" % t_launch(inter, np.full(N, 11)))
print("columns sorted by time (heavy groups first): %.3f ms" % t_launch(order, np.full(N, 11)))
print("columns sorted, light first: %.3f ms" % t_launch(order[::-1].copy(), np.full(N, 11)))
heavy = np.argsort(-t1)[:8]
ll = dev.gibbs_ll_cols(allc, np.full(N, 11), (A * W)[np.full(N, 11), allc], ws)
print("heaviest columns:", " ".join("%d (%.3f ms, %d non-finite)" % (c, t1[c], int((~np.isfinite(ll[c])).sum())) for c in heavy))
print("columns with a non-finite value:", np.nonzero((~np.isfinite(ll)).any(axis=1))[0], "their times", t1[(~np.isfinite(ll)).any(axis=1)])
