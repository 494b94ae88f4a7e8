"""Randomised check of the lock-step optimizer against sequential scipy fits (dev tool, GPU):
    python tools/fuzz_map.py [seed ...]
Every seed draws a model family (standard_glm; spatiotemporal_glm with its low-rank spatial basis or with an identity basis
over many pixels: the frame-rate kernels; sparse_weighted_model: Dirichlet impulses), a population size, a recording
length, a firing rate and a starting point, runs fit_glms_batched_torch with both forms of the inverse Hessian, and
fits up to three neurons with fit_glm (scipy BFGS: coord_descent.py:161-204) from the same start.  Reported per seed:
the largest relative difference of the final objective (implicit vs dense, lock-step vs scipy) and of the iteration
counts; a line starting with DISCREPANCY when lock-step ends more than 1e-6 (relative) ABOVE scipy on a neuron both
call converged, or when the two forms differ by more than 1e-8."""
import copy, sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch, supported

seeds = [int(a) for a in sys.argv[1:]] or list(range(8))
bad = 0
for seed in seeds:
    rng = np.random.default_rng(9000 + seed)
    family = ['standard_glm', 'spatiotemporal_lowrank', 'spatiotemporal_identity', 'sparse_weighted_model'][seed % 4]
    N = int(rng.choice([3, 8, 16, 23, 40, 64]))
    T = float(rng.choice([8.0, 20.0, 45.0]))
    rate = float(rng.choice([5.0, 20.0, 40.0]))
    dt, dt_stim = 0.001, 0.1
    nT = int(round(T / dt))
    S = np.minimum(rng.poisson(rate * dt, size=(nT, N)), 10).astype(np.uint8)
    stim = None
    if family == 'standard_glm':
        model = make_model('standard_glm', N=N, dt=dt)
    elif family == 'sparse_weighted_model':
        model = make_model('sparse_weighted_model', N=N, dt=dt)
        stabilize_sparsity(model)
    else:
        D = int(rng.choice([2, 5])) if family.endswith('lowrank') else int(rng.choice([64, 200, 513]))
        stim = rng.standard_normal((int(round(T / dt_stim)), D))
        tmpl = templates.spatiotemporal_glm()
        tmpl['bkgd']['D_stim'] = D
        if family.endswith('identity'):
            tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
        model = make_model(tmpl, N=N, dt=dt)
    popn = Population(model)
    popn.add_data({'S': S, 'N': N, 'dt': dt, 'T': T, 'stim': stim, 'dt_stim': dt_stim})
    x0 = popn.sample(np.random.RandomState(seed))
    if family.endswith('identity'):
        for g in x0['glms']:
            g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(D))
    if not supported(popn):
        print("seed %d %s N=%d: lock-step path not supported for this model, skipped" % (seed, family, N))
        popn.release_data()
        continue
    maxiter = int(rng.choice([60, 225]))
    t0 = time.perf_counter()
    xd, xi = copy.deepcopy(x0), copy.deepcopy(x0)
    fd, _, _ = fit_glms_batched_torch(popn, xd, maxiter=maxiter, hessian='dense')
    sd = popn.last_fit_stats
    fi, _, _ = fit_glms_batched_torch(popn, xi, maxiter=maxiter, hessian='implicit')
    si = popn.last_fit_stats
    forms = float(np.max(np.abs(fd - fi) / np.maximum(1.0, np.abs(fd))))
    dit = int(np.max(np.abs(np.array(sd['per_neuron']['iterations']) - np.array(si['per_neuron']['iterations']))))
    prms = cd.prep_first_order_glm_inference(popn)
    worst, wn, nits = 0.0, -1, []
    with np.errstate(all='ignore'):
        for n in rng.choice(N, size=min(3, N), replace=False):
            n = int(n)
            xs = copy.deepcopy(x0)
            res = cd.fit_glm(popn.extract_vars(xs, n), n, prms, maxiter=maxiter)
            rel = (fi[n] - res.fun) / max(1.0, abs(res.fun))        # > 0: lock-step ended above scipy
            reld = (fd[n] - res.fun) / max(1.0, abs(res.fun))
            nits.append((n, int(res.nit), int(si['per_neuron']['iterations'][n]), bool(res.success),
                         "%.1e" % rel, "%.1e" % reld))
            if res.success and rel > worst:
                worst, wn = float(rel), n
    flag = (worst > 1e-6) or (forms > 1e-8 and maxiter == 225 and sd['converged_gtol'] == N)
    bad += int(flag)
    print("%sseed %d %s N=%d T=%g rate=%g P=%d maxiter=%d: dense vs implicit %.1e (iterations differ by <= %d), "
          "converged %d/%d | vs scipy (neuron, nit scipy, nit lock-step, scipy success, (implicit - scipy) / |scipy|, (dense - scipy) / |scipy|): %s, worst excess over a "
          "converged scipy fit %.1e | %.1f s"
          % ("DISCREPANCY " if flag else "", seed, family, N, T, rate, popn.glm.P, maxiter,
             forms, dit, si['converged_gtol'], N, nits, worst, time.perf_counter() - t0), flush=True)
    popn.release_data()
print("%d seeds, %d discrepancies" % (len(seeds), bad))
