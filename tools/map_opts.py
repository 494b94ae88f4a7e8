"""Lock-step MAP fit of a named configuration under several optimizer options (dev tool):
    python tools/map_opts.py standard_glm 128 600 '{}' '{"init_scaling": true}'
    python tools/map_opts.py spatiotemporal_glm 64 300 '{}' '{"init_scaling": true}'       (D_stim = 3)
    python tools/map_opts.py stress 64 300 ...                                             (D_stim = 1024)"""
import copy, json, sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
from theano_pyglm_amd.inference.smart_init import initialize_with_data

name, N, T = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
optsets = [json.loads(a) for a in sys.argv[4:]] or [{}]
dt = 0.001
nT = int(round(T / dt))
cfg = {'standard_glm': 3, 'spatiotemporal_glm': 5, 'stress': 5}[name]
rng = np.random.default_rng(1234 + cfg)
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
stim = None
if name == 'standard_glm':
    model = make_model('standard_glm', N=N, dt=dt)
else:
    D = 1024 if name == 'stress' else 3
    stim = rng.standard_normal((nT // 100, D))
    tmpl = templates.spatiotemporal_glm()
    tmpl['bkgd']['D_stim'] = D
    tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
    model = make_model(tmpl, N=N, dt=dt)
popn = Population(model)
popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': stim, 'dt_stim': 0.1})
x0 = popn.sample(np.random.RandomState(0))
if name == 'stress':
    for g in x0['glms']:
        g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(1024))
initialize_with_data(popn, popn.data_sequences[-1], x0)
res = []
for opts in optsets:
    for rep in range(2):
        xb = copy.deepcopy(x0)
        t0 = time.perf_counter()
        nlp, it, ev = fit_glms_batched_torch(popn, xb, **opts)
        wall = time.perf_counter() - t0
    st = dict(popn.last_fit_stats)
    st.pop('per_neuron', None)
    res.append(nlp)
    print("%s: %.3f s  sum nlp %.4f  %s" % (json.dumps(opts), wall, nlp.sum(), st), flush=True)
for i in range(1, len(res)):
    d = res[i] - res[0]
    print("option set %d - set 0, per neuron: min %.3f median %.3f max %.3f; better in %d of %d" % (i, d.min(), np.median(d), d.max(), int((d < 0).sum()), len(d)))
