"""C5 stress variant (spatiotemporal_glm N=64, T=300 s, D_stim=1024, identity spatial basis; P = 1220 per neuron):
one MAP sweep through the lock-step optimizer, optionally against sequential scipy fits of a few neurons (dev tool).

    python tools/stress_map.py [--maxiter 225] [--scipy 0,21,42,63] [--reps 2] [--no-sta] [--N 64] [--D 1024] [--merge doubles]
"""
import argparse, copy, json, sys, time
import numpy as np
sys.path.insert(0, '.')

ap = argparse.ArgumentParser()
ap.add_argument('--maxiter', type=int, default=225)
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--scipy', default='')
ap.add_argument('--N', type=int, default=64)
ap.add_argument('--T', type=float, default=300.0)
ap.add_argument('--D', type=int, default=1024)
ap.add_argument('--no-sta', action='store_true')
ap.add_argument('--merge', type=int, default=-1, help='PGL_OPT_BFGS_MERGE: doubles of update history up to which an iteration is one kernel')
ap.add_argument('--opts', default='{}', help='JSON dict of keyword options for fit_glms_batched_torch')
args = ap.parse_args()

import __graft_entry__ as ge
ge.build_hip()
import torch                              # (as in bench.py: imported and initialised before the sweep is timed)
torch.zeros(8, device='cuda').sum().item()
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.models.model_factory import make_model
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
from theano_pyglm_amd.inference.smart_init import initialize_with_data

N, T, D, dt, dt_stim = args.N, args.T, args.D, 0.001, 0.1
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
if not args.no_sta:
    initialize_with_data(popn, popn.data_sequences[-1], x0)
lp0 = popn.compute_log_p(x0)
if args.merge >= 0:
    for d in popn.data_sequences:
        popn._handle(d).set_option(7, args.merge)
opts = json.loads(args.opts)
for rep in range(args.reps):
    xb = copy.deepcopy(x0)
    t0 = time.perf_counter()
    nlp_b, iters, evals = fit_glms_batched_torch(popn, xb, maxiter=args.maxiter, **opts)
    wall = time.perf_counter() - t0
    lp1 = popn.compute_log_p(xb)
    print("rep %d: %.3f s  log p %.4f -> %.4f  %s" % (rep, wall, lp0, lp1, popn.last_fit_stats), flush=True)
if args.scipy:
    prms = cd.prep_first_order_glm_inference(popn)
    for n in [int(v) for v in args.scipy.split(',')]:
        nv = popn.extract_vars(copy.deepcopy(x0), n)
        t0 = time.perf_counter()
        res = cd.fit_glm(nv, n, prms, maxiter=args.maxiter)
        print("neuron %d scipy: nit %d nfev %d fun %.9f (%s) %.1f s | lock-step %.9f  (lock-step - scipy)/|scipy| %.2e  |g|max %.2e"
              % (n, res.nit, res.nfev, res.fun, res.message[:34], time.perf_counter() - t0, nlp_b[n],
                 (nlp_b[n] - res.fun) / abs(res.fun), np.max(np.abs(popn.compute_grad(xb, n)))), flush=True)
