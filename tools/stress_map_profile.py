"""cProfile of coord_descent(maxiter=1) on the C5 stress model (dev tool)."""
import copy, cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
N, T, D, dt, dt_stim = 64, 300.0, 1024, 0.001, 0.1
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 5)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
stim = rng.standard_normal((int(round(T / dt_stim)), D))
tmpl = templates.spatiotemporal_glm()
tmpl['bkgd']['D_stim'] = D
tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
popn = Population(make_model(tmpl, N=N, dt=dt))
popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': stim, 'dt_stim': dt_stim})
x0 = popn.sample(np.random.RandomState(0))
for g in x0['glms']:
    g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(D))
cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
cd.coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1)
pr.disable()
print("wall %.3f s" % (time.time() - t0))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
