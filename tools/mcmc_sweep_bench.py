"""Wall-clock of full Gibbs sweeps at the C4 shape (sparse_weighted_model, N=128, T=600 s) through
the host mirror: HMC bias block (11 batched ll+grad evals), Dirichlet impulse rounds (3 evals per
in-degree round), collapsed network column updates (N^2 inner-ll batches + ARS).  Dev tool.

    python tools/mcmc_sweep_bench.py [N] [T] [sweeps]
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from theano_pyglm_amd.inference import gibbs as G
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nT = int(round(T / 0.001))
rng = np.random.RandomState(1238)
model = make_model('sparse_weighted_model', N=N, dt=0.001)
stabilize_sparsity(model)
popn = Population(model)
S = np.minimum(rng.poisson(0.02, size=(nT, N)), 10).astype(np.uint8)         # 20 Hz
data = {'S': S, 'N': N, 'dt': 0.001, 'T': T, 'stim': None, 'dt_stim': 0.1}
t0 = time.time()
popn.add_data(data)
print("upload + event lists: %.2f s" % (time.time() - t0))
x = popn.sample(rng)
x['net']['weights']['W'] = 0.05 * np.asarray(x['net']['weights']['W'])
print("A density %.4f (rho %.4f)" % (np.mean(x['net']['graph']['A']), model['network']['graph']['rho']))
serial, par = G.initialize_updates(popn, rng)
for s in range(sweeps):
    t_sweep = time.time()
    lp = popn.compute_log_p(x)
    line = []
    for upd in par:
        t0 = time.time()
        upd.update_all(x)
        line.append("%s %.3f s" % (type(upd).__name__, time.time() - t0))
    print("sweep %d: log p %.1f | %s | total %.2f s" % (s, lp, " | ".join(line), time.time() - t_sweep))
net = par[-1]
print("ARS extra inner-ll evaluations: %d; HMC batched evals: bias %d, impulse %d"
      % (net.n_ars_evals, par[0].n_evals, par[2].n_evals))
