"""Where the MAP sweep's wall-clock goes (dev tool): cProfile of coord_descent(batched='torch')."""
import sys, time, copy, cProfile, pstats
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd

N, T, dt = 128, 600.0, 0.001
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 3)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
popn = Population(make_model('standard_glm', N=N, dt=dt))
popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': None, 'dt_stim': 0.1})
x0 = popn.sample(np.random.RandomState(0))
popn.compute_log_p(x0)
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    x = cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    pr.disable()
    print("rep %d wall %.3f s" % (rep, time.time() - t0))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
