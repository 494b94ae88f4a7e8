"""Trajectory of one neuron of the C5 stress model: sequential scipy fit (callback values) against the lock-step fit of the
one-neuron shard and against the same neuron inside the 64-neuron sweep, at several maxiter (dev tool)."""
import copy, sys
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
from theano_pyglm_amd.inference.smart_init import initialize_with_data
import scipy.optimize as opt
from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars

n = int(sys.argv[1]) if len(sys.argv) > 1 else 63
N, T, D, dt = 64, 300.0, 1024, 0.001
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 5)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
stim = rng.standard_normal((nT // 100, D))
tmpl = templates.spatiotemporal_glm()
tmpl['bkgd']['D_stim'] = D
tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
popn = Population(make_model(tmpl, N=N, dt=dt))
popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': stim, 'dt_stim': 0.1})
x0 = popn.sample(np.random.RandomState(0))
for g in x0['glms']:
    g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(D))
initialize_with_data(popn, popn.data_sequences[-1], x0)
glm_syms, nlp, grad_nlp = cd.prep_first_order_glm_inference(popn)
nv = popn.extract_vars(copy.deepcopy(x0), n)
v0, shapes = packdict(get_vars(glm_syms, nv['glm']))
traj = []
res = opt.minimize(lambda v: nlp(v, nv), v0, jac=lambda v: grad_nlp(v, nv), method='bfgs', options={'maxiter': 225},
                   callback=lambda xk: traj.append(nlp(xk, nv)))
print("scipy nit %d nfev %d fun %.9f" % (res.nit, res.nfev, res.fun))
for k in (5, 20, 50, 100, 150, 200, 225):
    xa = copy.deepcopy(x0)
    f1, _, _ = fit_glms_batched_torch(popn, xa, maxiter=k, n_lo=n, n_hi=n + 1)
    s1 = popn.last_fit_stats['per_neuron']
    xb = copy.deepcopy(x0)
    f64, _, _ = fit_glms_batched_torch(popn, xb, maxiter=k)
    s64 = popn.last_fit_stats['per_neuron']
    print("iter %3d: scipy %.9f | shard-of-one %.9f (rel %.1e, ls steps %d) | in the 64-sweep %.9f (rel %.1e, ls steps %d)"
          % (k, traj[k - 1], f1[0], (f1[0] - traj[k - 1]) / abs(traj[k - 1]), s1['line_search_steps'][0], f64[n],
             (f64[n] - traj[k - 1]) / abs(traj[k - 1]), s64['line_search_steps'][n]), flush=True)
