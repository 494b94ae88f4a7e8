"""C5 (spatiotemporal_glm) MAP: the GPU lock-step optimizer against sequential scipy fits (dev tool).
    python tools/c5_map_check.py [N] [T] [poisson|model]"""
import sys, time, copy
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
src = sys.argv[3] if len(sys.argv) > 3 else 'poisson'
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else None          # override of the impulse prior sigma (template: 0.001)
nT = int(round(T / 0.001))
if src == 'model':
    from theano_pyglm_amd.harness.generate_synth_data import make_dataset
    t0 = time.time()
    model, popn, data = make_dataset('spatiotemporal_glm', N, T, seed=1234 + 5, check=False)
    print("simulated %.1fs, rates (Hz) min %.1f median %.1f max %.1f" % (time.time() - t0, (data['S'].sum(0) / T).min(),
          np.median(data['S'].sum(0) / T), (data['S'].sum(0) / T).max()))
    popn.add_data(data)
else:
    rng = np.random.default_rng(1234 + 5)
    S = np.minimum(rng.poisson(20.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
    stim = np.random.RandomState(1234 + 5).randn(nT // 100, 3)
    model = make_model('spatiotemporal_glm', N=N, dt=0.001)
    if sigma is not None:
        model['impulse']['sigma'] = sigma
    popn = Population(model)
    popn.add_data({'S': S, 'N': N, 'dt': 0.001, 'T': T, 'stim': stim, 'dt_stim': 0.1})
x0 = popn.sample(np.random.RandomState(7))
res_b = {}
for scaling in (True, False):
    xb = copy.deepcopy(x0)
    t0 = time.time()
    nlp_b, iters, evals = fit_glms_batched_torch(popn, xb, init_scaling=scaling)
    print("lock-step init_scaling=%s: %.2f s  %s" % (scaling, time.time() - t0, popn.last_fit_stats))
    res_b[scaling] = (nlp_b.copy(), copy.deepcopy(xb))
print("sum nlp: scaling %.6f  no scaling %.6f" % (res_b[True][0].sum(), res_b[False][0].sum()))
nlp_b, xb = res_b[False]
prms = cd.prep_first_order_glm_inference(popn)
for n in (0, N // 3, 2 * N // 3, N - 1):
    for mi in (225, 1000):
        xs = copy.deepcopy(x0)
        nv = popn.extract_vars(xs, n)
        t0 = time.time()
        res = cd.fit_glm(nv, n, prms, maxiter=mi)
        print("neuron %d scipy maxiter %d: nit %d nfev %d fun %.9f (%s) %.1fs | lock-step %.9f  rel diff %.2e  |g|max %.2e"
              % (n, mi, res.nit, res.nfev, res.fun, res.message[:30], time.time() - t0, nlp_b[n],
                 (nlp_b[n] - res.fun) / abs(res.fun), np.max(np.abs(popn.compute_grad(xb, n)))))
