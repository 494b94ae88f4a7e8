"""Where the FIRST MAP sweep of a process spends its time (later sweeps: tools/map_bench.py).  Dev tool."""
import sys, time, copy, cProfile, pstats, io
import numpy as np
sys.path.insert(0, '.')
t00 = time.time()
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
N, T, dt = 128, 600.0, 0.001
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 3)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
popn = Population(make_model('standard_glm', N=N, dt=dt))
t0 = time.time(); popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': None, 'dt_stim': 0.1}); print("add_data %.2f s" % (time.time() - t0))
x0 = popn.sample(np.random.RandomState(0))
t0 = time.time(); lp0 = popn.compute_log_p(x0); print("first compute_log_p %.2f s" % (time.time() - t0))
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
x = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, verbose=False)
print("first sweep %.2f s" % (time.time() - t0))
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
