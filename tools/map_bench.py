"""MAP wall-clock (secondary metric): one coord_descent(maxiter=1) sweep = all N
per-neuron BFGS fits (<= 225 iterations each), excluding data generation.  Dev tool."""
import sys, time, copy
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
mode = sys.argv[3] if len(sys.argv) > 3 else 'torch'
dt = 0.001
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 3)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
model = make_model('standard_glm', N=N, dt=dt)
popn = Population(model)
data = {'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': None, 'dt_stim': 0.1}
t0 = time.time(); popn.add_data(data); print("add_data %.2fs" % (time.time() - t0))
x0 = popn.sample(np.random.RandomState(0))
lp0 = popn.compute_log_p(x0)
for rep in range(3):
    t0 = time.time()
    x = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1,
                         batched={'torch': 'torch', 'numpy': True, 'seq': False, 'default': None}[mode], verbose=False)
    wall = time.time() - t0
    print("MAP sweep %d mode=%s N=%d T=%gs: wall %.3f s  %s" % (rep, mode, N, T, wall, getattr(popn, 'last_fit_stats', None)))
lp1 = popn.compute_log_p(x)
print("log p %.3f -> %.3f" % (lp0, lp1))
