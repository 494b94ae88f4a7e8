"""C5 stress variant end to end (dev tool): data simulated from a spatiotemporal_glm draw with a 32x32-pixel stimulus
(D_stim = 1024, identity spatial basis), default MAP sweep (lock-step BFGS on the separable device path), timing."""
import sys, time, copy
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.harness.generate_synth_data import make_dataset
from theano_pyglm_amd.harness import synth_map
from theano_pyglm_amd.inference.coord_descent import coord_descent

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
tmpl = templates.spatiotemporal_glm()
tmpl['bkgd']['D_stim'] = D
tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
t0 = time.time()
def tame(x):
    for g in x['glms']:
        g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(D))
        g['bkgd']['w_t'] = np.asarray(g['bkgd']['w_t']) * 0.5
        g['bias']['bias'] = 2.5 + 0.3 * np.asarray(g['bias']['bias'])
        g['imp']['w_ir'] = np.asarray(g['imp']['w_ir']) * 0.1
model, popn_true, data = make_dataset(tmpl, N, T, seed=1234 + 5, adjust=tame)
print("data: %.1f s; rates %.1f..%.1f Hz" % (time.time() - t0, data['S'].sum(0).min() / T, data['S'].sum(0).max() / T), flush=True)
x_true = data['vars']
clean = dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed'))
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.models.model_factory import make_model
popn = Population(make_model(tmpl, N=N, dt=0.001))
popn.add_data(dict(clean))
print("stim path:", popn._handle(popn._current).info()['stim_path'], "separable:", popn.glm.bkgd_model.separable, flush=True)
x0 = popn.sample(np.random.RandomState(3))
lp0 = popn.compute_log_p(x0)
import cProfile, pstats, os
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    if os.environ.get('CPROF'): pr.enable()
    x_inf = coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1)
    if os.environ.get('CPROF'): pr.disable()
    print("MAP sweep %.2f s" % (time.time() - t0), popn.last_fit_stats, flush=True)
if os.environ.get('CPROF'): pstats.Stats(pr).sort_stats('tottime').print_stats(14)
lp1 = popn.compute_log_p(x_inf)
ll_true = popn_true.compute_ll(x_true)
print("log p: initial %.1f -> MAP %.1f; ll MAP %.1f vs ll true %.1f" % (lp0, lp1, popn.compute_ll(x_inf), ll_true))
wx_t = np.array([g['bkgd']['w_x'] for g in x_true['glms']]); wx_i = np.array([g['bkgd']['w_x'] for g in x_inf['glms']])
c = [abs(np.corrcoef(wx_t[n], wx_i[n])[0, 1]) for n in range(N)]
print("|corr| of recovered spatial filters: median %.3f min %.3f" % (np.median(c), np.min(c)))
