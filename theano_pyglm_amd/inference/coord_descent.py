"""
MAP estimation by coordinate descent over neurons -- counterpart of
pyglm/inference/coord_descent.py.

`prep_first_order_glm_inference`, `fit_glm` and `coord_descent` keep the reference's
signatures and semantics (per-neuron BFGS, maxiter 225, NaN -> 1e16 / NaN gradient -> 0,
convergence on |delta log p| < atol).  The gradient the reference gets from
T.grad(glm.ll)/T.grad(glm.log_prior) (coord_descent.py:27-30) comes from the fused HIP
ll+grad kernel plus closed-form prior gradients.

For standard_glm-like models (constant weights, complete graph) the N per-neuron problems
are independent, so `fit_glms_batched` advances all of them in lock-step:
one fused device pass per iteration evaluates ll and gradient of every neuron at its own
trial point.  `coord_descent(..., batched=True)` uses it (optimizer state in numpy),
`batched='torch'` keeps the optimizer state on the GPU (inference/batched_bfgs.py) and is what
the default (batched=None) selects whenever the model's packing is served; batched=False
reproduces the reference's sequential sweep of scipy fits.
"""
import copy

import numpy as np
import scipy.optimize as opt

from theano_pyglm_amd.inference.smart_init import initialize_with_data
from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars
from theano_pyglm_amd.components.network import CompleteGraphModel


def _neuron_lp_grad(population, x, want_grad):
    """log_prior + sum_data ll (and packed gradient) for the neuron described by
    x = extract_vars(state, n)."""
    glm = population.glm
    syms = population.glm_syms()
    xn = x['glm']
    n = int(xn['n'])
    lp = glm.log_prior(xn)
    g = None
    if want_grad:
        g, _ = packdict(get_vars(syms, glm.grad_log_prior(xn)))
    theta = glm.theta_row(xn)[None, :]
    Weff = population.network.W_eff(x['net'])
    for data in population.data_sequences:
        population.set_data(data)
        h = population._handle(data)
        ll, gt = h.ll_grad(theta, Weff, n, n + 1, want_grad=want_grad)
        lp += ll[0]
        if want_grad:
            gv, _ = packdict(get_vars(syms, glm.chain_grad(xn, gt[0])))
            g = g + gv
    return lp, g


def prep_first_order_glm_inference(population):
    """coord_descent.py:15-82: returns (glm_syms, nlp, grad_nlp)."""
    glm_syms = population.glm_syms()
    x0 = population.shape_vars()          # (the reference draws a sample here, for the shapes only: coord_descent.py:24-26)
    nvars = population.extract_vars(x0, 0)
    _, glm_shapes = packdict(get_vars(glm_syms, nvars['glm']))

    def nlp(x_glm_vec, x):
        x_glm = unpackdict(x_glm_vec, glm_shapes)
        set_vars(glm_syms, x['glm'], x_glm)
        lp, _ = _neuron_lp_grad(population, x, False)
        return -1.0 * lp

    def grad_nlp(x_glm_vec, x):
        x_glm = unpackdict(x_glm_vec, glm_shapes)
        set_vars(glm_syms, x['glm'], x_glm)
        _, g = _neuron_lp_grad(population, x, True)
        return -1.0 * g

    return glm_syms, nlp, grad_nlp


def prep_first_order_network_inference(population):
    """coord_descent.py:84-132.  Only float network variables are optimised; for constant
    weights / complete graphs there are none and fit_network is a no-op (:141)."""
    from theano_pyglm_amd.utils.syms import differentiable
    network = population.network
    net_syms = differentiable(population.get_variables()['net'])
    x0 = population.shape_vars()          # (shapes only: coord_descent.py:93-95)
    _, shapes = packdict(get_vars(net_syms, x0['net']))

    def nlp(x_vec, x):
        set_vars(net_syms, x['net'], unpackdict(x_vec, shapes))
        return -1.0 * network.log_p(x['net'])          # the reference's network.log_prior typo: log_p

    def grad_nlp(x_vec, x):
        set_vars(net_syms, x['net'], unpackdict(x_vec, shapes))
        g = {'graph': {}, 'weights': network.weights.grad_log_p(x['net']['weights'])}
        gv, _ = packdict(get_vars(net_syms, g))
        return -1.0 * gv

    return net_syms, nlp, grad_nlp


def fit_network(x, net_inf_prms):
    """coord_descent.py:134-159."""
    net_syms, net_nll, g_net_nll = net_inf_prms
    x_net_0, shapes = packdict(get_vars(net_syms, x['net']))
    if x_net_0.size > 0:
        res = opt.minimize(lambda v: net_nll(v, x), x_net_0, jac=lambda v: g_net_nll(v, x),
                           method='Newton-CG')
        set_vars(net_syms, x['net'], unpackdict(res.x, shapes))


def fit_glm(xn, n, glm_inf_prms, verbose=False, maxiter=225):
    """coord_descent.py:161-204."""
    glm_syms, glm_nll, g_glm_nll = glm_inf_prms
    x_glm_0, shapes = packdict(get_vars(glm_syms, xn['glm']))

    def nll(v):
        y = glm_nll(v, xn)
        if np.isnan(y):
            y = 1e16
        return y

    def grad_nll(v):
        g = g_glm_nll(v, xn)
        if np.any(np.isnan(g)):
            g = np.zeros_like(g)
        return g

    it = [0]

    def cbk(x_curr):
        if verbose:
            print("Newton iter %d.\tNeuron %d. LL: %.1f" % (it[0], n, -1.0 * nll(x_curr)))
        it[0] += 1

    res = opt.minimize(nll, x_glm_0, method="bfgs", jac=grad_nll,
                       options={'disp': verbose, 'maxiter': maxiter}, callback=cbk)
    set_vars(glm_syms, xn['glm'], unpackdict(res.x, shapes))
    return res


def fit_glms_batched(population, x, maxiter=225, gtol=1e-5, n_lo=0, n_hi=None, verbose=False):
    """Lock-step BFGS for neurons [n_lo,n_hi): every iteration costs one fused device pass
    (ll + gradient of all those neurons at their own trial points).  Backtracking Armijo
    line search, BFGS update skipped when s.y <= 0; NaN handling as fit_glm.  Updates x in
    place and returns the per-neuron negative log posteriors."""
    N = population.N
    n_hi = N if n_hi is None else n_hi
    M = n_hi - n_lo
    syms = population.glm_syms()
    vecs, shapes = [], None
    for n in range(n_lo, n_hi):
        v, shapes = packdict(get_vars(syms, x['glms'][n]))
        vecs.append(v)
    X = np.array(vecs)
    P = X.shape[1]

    def evaluate(Xt):
        for i, n in enumerate(range(n_lo, n_hi)):
            set_vars(syms, x['glms'][n], unpackdict(Xt[i], shapes))
        lp, g = population.compute_lp_grad_packed(x, n_lo, n_hi)
        f = -lp
        g = -g
        f = np.where(np.isnan(f), 1e16, f)
        g[np.any(np.isnan(g), axis=1)] = 0.0
        return f, g

    f, g = evaluate(X)
    H = np.tile(np.eye(P)[None], (M, 1, 1))
    active = np.linalg.norm(g, ord=np.inf, axis=1) > gtol
    it = 0
    n_evals = 1
    while it < maxiter and np.any(active):
        it += 1
        p = -np.einsum('mij,mj->mi', H, g)
        slope = np.einsum('mi,mi->m', p, g)
        bad = slope >= 0                                   # not a descent direction: reset
        if np.any(bad):
            H[bad] = np.eye(P)
            p[bad] = -g[bad]
            slope[bad] = -np.einsum('mi,mi->m', g[bad], g[bad])
        alpha = np.ones(M)
        if it == 1:
            alpha = np.minimum(1.0, 1.0 / np.maximum(np.linalg.norm(g, axis=1), 1e-300))
        done = ~active
        Xn, fn, gn = X.copy(), f.copy(), g.copy()
        for _ls in range(30):
            Xt = np.where(done[:, None], Xn, X + alpha[:, None] * p)
            ft, gt = evaluate(Xt)
            n_evals += 1
            ok = (~done) & (ft <= f + 1e-4 * alpha * slope)
            Xn[ok], fn[ok], gn[ok] = Xt[ok], ft[ok], gt[ok]
            done = done | ok
            if np.all(done):
                break
            alpha = np.where(done, alpha, alpha * 0.5)
        stalled = ~done                                    # line search failed: freeze neuron
        s = Xn - X
        y = gn - g
        sy = np.einsum('mi,mi->m', s, y)
        upd = active & (~stalled) & (sy > 1e-12)
        for m in np.nonzero(upd)[0]:
            rho = 1.0 / sy[m]
            Hy = H[m].dot(y[m])
            H[m] += (1.0 + rho * y[m].dot(Hy)) * rho * np.outer(s[m], s[m]) \
                - rho * (np.outer(Hy, s[m]) + np.outer(s[m], Hy))
        X, f, g = Xn, fn, gn
        active = active & (~stalled) & (np.linalg.norm(g, ord=np.inf, axis=1) > gtol)
        if verbose:
            print("batched BFGS iter %d: active %d, mean nlp %.3f, evals %d"
                  % (it, int(active.sum()), float(np.mean(f)), n_evals))
    for i, n in enumerate(range(n_lo, n_hi)):
        set_vars(syms, x['glms'][n], unpackdict(X[i], shapes))
    return f, it, n_evals


def resolve_batched(population, batched):
    """batched=None (the default of coord_descent and the harness): the GPU lock-step optimizer whenever the
    model's packing is served (batched_bfgs.supported), else the reference's sequential per-neuron scipy fits;
    False / True / 'torch' force the sequential / numpy lock-step / GPU lock-step sweep."""
    if batched is None:
        from theano_pyglm_amd.inference.batched_bfgs import supported
        if not supported(population):
            return False
        # the GPU optimizer needs torch with a visible device and a library that exports the row kernels; without them
        # the default stays what it always was, the sequential sweep (which raises PglError itself if there is no GPU)
        try:
            import torch
            from theano_pyglm_amd import _lib
            if not torch.cuda.is_available() or not hasattr(_lib.load(), 'pgl_bfgs_update_dev'):
                return False
        except Exception:
            return False
        return 'torch'
    return batched


def coord_descent(population, x0=None, maxiter=50, atol=1e-5, batched=None, verbose=False):
    """coord_descent.py:206-266.  `batched`: see resolve_batched (None = automatic; batched=False is the
    reference's sweep of N sequential scipy BFGS fits, kept for comparison)."""
    N = population.model['N']
    batched = resolve_batched(population, batched)
    network = population.network
    if not isinstance(network.graph, CompleteGraphModel):
        print(" WARNING: MAP inference via coordinate descent can only be performed "
              "with the complete graph model.")
    if x0 is None:
        x0 = population.sample()
    initialize_with_data(population, population.data_sequences[-1], x0)
    lp = population.compute_log_p(x0)
    if verbose:
        print("Initial LP=%.2f." % lp)
    net_inf_prms = prep_first_order_network_inference(population)
    glm_inf_prms = prep_first_order_glm_inference(population)
    x = x0
    lp_prev = lp                                      # x is x0: the value just computed
    converged = False
    it = 0
    while not converged and it < maxiter:
        it += 1
        if batched == 'torch':
            from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
            fit_glms_batched_torch(population, x, verbose=verbose)
        elif batched:
            fit_glms_batched(population, x, verbose=verbose)
        else:
            for n in np.arange(N):
                nvars = population.extract_vars(x, n)
                fit_glm(nvars, n, glm_inf_prms, verbose=verbose)
                x['glms'][n] = nvars['glm']
        fit_network(x, net_inf_prms)
        lp = population.compute_log_p(x)
        if verbose:
            print("Iteration %d: LP=%.2f. Change in LP: %.2f" % (it, lp, lp - lp_prev))
        converged = np.abs(lp - lp_prev) < atol
        lp_prev = lp
    return x
