"""
Lock-step BFGS for all neurons with the optimizer state resident on the GPU.

For standard_glm-like models (LinearBasisImpulses + No/Basis stimulus) the packed
per-neuron vector of coord_descent ([bias, w_stim, w_ir]) *is* the flat
feature-weight row the device kernel consumes, and the N per-neuron MAP problems are
independent (constant weights / complete graph).  So the whole sweep of
coord_descent.fit_glm over n = 0..N-1 (coord_descent.py:161-204, 243-247) runs as one
batched BFGS: per iteration one fused ll+grad launch on device pointers, the priors and the
dense inverse-Hessian updates (M x P x P) as torch tensor ops on the same GPU -- nothing but
a few scalars crosses PCIe.  PyTorch is plumbing here (device memory + batched BLAS); the
likelihood and its gradient come from the HIP kernels.

NaN semantics of fit_glm are kept: nll NaN -> 1e16, NaN gradient -> 0 (coord_descent.py:170-182).
"""
import numpy as np

from theano_pyglm_amd.components.bkgd import NoStimulus, BasisStimulus
from theano_pyglm_amd.components.impulse import LinearBasisImpulses
from theano_pyglm_amd.components.priors import Gaussian, GroupLasso


def supported(population):
    glm = population.glm
    return isinstance(glm.imp_model, LinearBasisImpulses) and \
        isinstance(glm.bkgd_model, (NoStimulus, BasisStimulus))


def _prior_terms(population, torch, X):
    """log prior (M,) and its gradient (M,P) of the rows [bias, w_stim, w_ir] in torch."""
    glm = population.glm
    N, B, D = population.N, glm.imp_model.B, glm.Dstim
    b = X[:, 0]
    mu_b, sg_b = float(glm.bias_model.mu_bias), float(glm.bias_model.sig_bias)
    lp = -0.5 / sg_b ** 2 * (b - mu_b) ** 2                                   # bias.py:33
    G = torch.zeros_like(X)
    G[:, 0] = -(b - mu_b) / sg_b ** 2
    if D > 0:                                                                 # bkgd.py:76
        ws = X[:, 1:1 + D]
        lp = lp - 0.5 / (0.01 ** 2) * (ws ** 2).sum(1)
        G[:, 1:1 + D] = -ws / (0.01 ** 2)
    w = X[:, 1 + D:].reshape(-1, N, B)
    pr = glm.imp_model.prior
    if isinstance(pr, GroupLasso):                                            # priors.py:202
        z = (w - float(pr.mu)) / float(pr.sigma)
        nrm = torch.sqrt((z ** 2).sum(2, keepdim=True))
        lp = lp - float(pr.lam) * nrm.sum((1, 2))
        G[:, 1 + D:] = (-float(pr.lam) * z / nrm / float(pr.sigma)).reshape(X.shape[0], -1)   # 0/0 -> NaN
    elif isinstance(pr, Gaussian):                                            # priors.py:139
        lp = lp - 0.5 / float(pr.sigma) ** 2 * ((w - float(pr.mu)) ** 2).sum((1, 2))
        G[:, 1 + D:] = (-(w - float(pr.mu)) / float(pr.sigma) ** 2).reshape(X.shape[0], -1)
    else:
        raise Exception("unsupported impulse prior for the batched GPU optimizer")
    return lp, G


def fit_glms_batched_torch(population, x, maxiter=225, gtol=1e-5, n_lo=0, n_hi=None, verbose=False,
                           reduce=None):
    """In-place MAP fit of x['glms'][n_lo:n_hi]; returns (nlp (M,), iterations, evaluations).

    Everything runs on one dedicated torch stream: the device handles are switched to it
    (pgl_set_stream), so prior terms, the fused ll+grad launches, the line-search bookkeeping and the
    inverse-Hessian updates are ordered by the stream -- no host synchronisation per evaluation; the
    host only reads the handful of scalars that steer the loop.  Neurons whose line search has
    already succeeded are masked out of the launch (pgl_ll_grad_list_dev evaluates an arbitrary
    list of neurons), so late, poorly conditioned neurons do not pay for the whole population.
    A neuron whose backtracking fails restarts once from steepest descent before it is frozen
    (scipy's BFGS stops there with "precision loss", coord_descent.py:194-199).

    `reduce`: optional callable applied in place to the packed device tensor [ll | grad] of every
    evaluation before the priors are added -- the all-reduce of a time-sharded multi-GPU fit (every rank
    evaluates its own bins of all neurons and runs the identical optimizer on the reduced values)."""
    import torch
    if not supported(population):
        raise Exception("batched GPU BFGS needs LinearBasisImpulses and No/Basis stimulus")
    N = population.N
    n_hi = N if n_hi is None else n_hi
    M = n_hi - n_lo
    dev = torch.device('cuda', population.device)
    handles = []
    for data in population.data_sequences:
        population.set_data(data)
        handles.append(population._handle(data))
    stream = torch.cuda.Stream(dev)
    for h in handles:
        h.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            out = _lockstep_bfgs(population, torch, dev, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose, reduce)
            stream.synchronize()
    finally:
        for h in handles:
            h.set_stream(None)
    return out


def _lockstep_bfgs(population, torch, dev, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose, reduce=None):
    X = torch.tensor(population.theta_matrix(x, n_lo, n_hi), dtype=torch.float64, device=dev)
    P = X.shape[1]
    Weff = torch.tensor(population.W_eff(x), dtype=torch.float64, device=dev)
    n_evals = [0]
    neuron_evals = [0]
    rows_all = torch.arange(M, device=dev)

    def evaluate(Xt, rows=None):
        """nlp and its gradient for the rows `rows` (default all) of the shard, at Xt (len(rows), P)."""
        Xt = Xt.contiguous()
        cnt = Xt.shape[0]
        lp, G = _prior_terms(population, torch, Xt)
        idx = None if rows is None or cnt == M else (rows + n_lo).to(torch.int32).contiguous()
        for h in handles:
            buf = torch.empty(cnt * (1 + P), dtype=torch.float64, device=dev)     # [ll | grad]: one all-reduce
            ll, gr = buf[:cnt], buf[cnt:].view(cnt, P)
            if idx is None:
                h.ll_grad_dev(Xt.data_ptr(), Weff.data_ptr(), ll.data_ptr(), gr.data_ptr(), n_lo, n_hi)
            else:
                h.ll_grad_list_dev(idx.data_ptr(), cnt, Xt.data_ptr(), Weff.data_ptr(), ll.data_ptr(),
                                   gr.data_ptr())
            if reduce is not None:
                reduce(buf)
            lp = lp + ll
            G = G + gr
        n_evals[0] += 1
        neuron_evals[0] += cnt
        f, g = -lp, -G
        f = torch.where(torch.isnan(f), torch.full_like(f, 1e16), f)
        bad = torch.isnan(g).any(1)
        g = torch.where(bad[:, None], torch.zeros_like(g), g)
        return f, g

    f, g = evaluate(X)
    eye = torch.eye(P, dtype=torch.float64, device=dev)
    H = eye.repeat(M, 1, 1)
    Hg = g.clone()                                              # H g, carried along: one pass over H per launch
    active = g.abs().amax(1) > gtol
    frozen = torch.zeros(M, dtype=torch.bool, device=dev)       # line search failed twice in a row
    restarts = torch.zeros(M, dtype=torch.int64, device=dev)
    iters = torch.zeros(M, dtype=torch.int64, device=dev)       # BFGS iterations of every neuron
    nhalf = torch.zeros(M, dtype=torch.int64, device=dev)       # step halvings of the current line search

    def first_step(gg):
        return torch.clamp(1.0 / gg.norm(dim=1).clamp_min(1e-300), max=1.0)

    # Every neuron runs its own BFGS state machine -- "line search at step alpha along p" -- and every launch
    # evaluates the pending trial point of ALL active neurons, whether that is the first trial of a new
    # iteration or a backtracking trial: a neuron whose trial succeeds moves on to its next iteration without
    # waiting for the neurons that still backtrack.  The iterates of a neuron are those of the iteration-
    # synchronous loop (its trial points do not depend on the others); the number of launches is the largest
    # number of trials any neuron needs instead of the sum over iterations of the per-iteration maximum, and
    # launches with a handful of backtracking neurons (1.2 ms for <= 16 neurons) disappear.
    p = -Hg
    slope = (p * g).sum(1)
    alpha = first_step(g)
    it = 0
    max_launches = maxiter * 31 + 2
    while bool(active.any()) and n_evals[0] < max_launches:
        rows = rows_all[active]
        Xt = X[rows] + alpha[rows, None] * p[rows]
        ft, gt = evaluate(Xt, rows)
        ok = ft <= f[rows] + 1e-4 * alpha[rows] * slope[rows]
        acc = torch.zeros(M, dtype=torch.bool, device=dev)
        acc[rows[ok]] = True
        fail = active & ~acc
        # ---- failed trials: halve the step; after 30 halvings restart once from steepest descent, then freeze
        alpha = torch.where(fail, alpha * 0.5, alpha)
        nhalf = torch.where(fail, nhalf + 1, nhalf)
        stalled = fail & (nhalf >= 30)
        # ---- accepted trials: BFGS update of H with (s, y), new direction
        Xn, fn, gn = X.clone(), f.clone(), g.clone()
        Xn[rows[ok]], fn[rows[ok]], gn[rows[ok]] = Xt[ok], ft[ok], gt[ok]
        s = Xn - X
        y = gn - g
        sy = (s * y).sum(1)
        upd = acc & (sy > 1e-12)
        # H g_new: the one full read of H (M x P x P) per launch; H y = H g_new - H g follows from it
        t = torch.bmm(H, gn[:, :, None])[:, :, 0]
        if bool(upd.any()):
            rho = torch.where(upd, 1.0 / sy.clamp_min(1e-300), torch.zeros_like(sy))
            Hy = t - Hg
            yHy = (y * Hy).sum(1)
            c = (1.0 + rho * yHy) * rho
            # H += c s s^T - rho (Hy s^T + s Hy^T) as ONE rank-3 update: a single read-modify-write of H
            U = torch.stack((c[:, None] * s, -rho[:, None] * Hy, -rho[:, None] * s), dim=2)       # (M, P, 3)
            V = torch.stack((s, s, Hy), dim=2)                                                    # (M, P, 3)
            H.baddbmm_(U, V.transpose(1, 2))
            t = t + torch.bmm(U, torch.bmm(V.transpose(1, 2), gn[:, :, None]))[:, :, 0]          # H_new g_new
        X, f, g, Hg = Xn, fn, gn, t
        iters = torch.where(acc, iters + 1, iters)
        restarts = torch.where(acc, torch.zeros_like(restarts), restarts)
        again = stalled & (restarts == 0)
        frozen = frozen | (stalled & ~again)
        restarts = torch.where(again, restarts + 1, restarts)
        # new line search for the accepted and the restarted neurons
        newls = acc | again
        pn = -Hg
        sl = (pn * g).sum(1)
        reset = (newls & (sl >= 0)) | again                         # not a descent direction / restart: H = I
        if bool(reset.any()):
            H[reset] = eye
            Hg = torch.where(reset[:, None], g, Hg)
            pn = torch.where(reset[:, None], -g, pn)
            sl = (pn * g).sum(1)
        p = torch.where(newls[:, None], pn, p)
        slope = torch.where(newls, sl, slope)
        alpha = torch.where(newls, torch.where(again, first_step(g), torch.ones_like(alpha)), alpha)
        nhalf = torch.where(newls, torch.zeros_like(nhalf), nhalf)
        active = active & (~frozen) & (g.abs().amax(1) > gtol) & (iters < maxiter)
        it = int(iters.max())
        if verbose:
            print("batched BFGS launch %d: active %d, mean nlp %.3f, max iteration %d"
                  % (n_evals[0], int(active.sum()), float(f.mean()), it))
    gmax = g.abs().amax(1)
    n_conv = int((gmax <= gtol).sum())
    n_frozen = int((frozen & (gmax > gtol)).sum())
    Xh = X.cpu().numpy()
    D = population.glm.Dstim
    for i, n in enumerate(range(n_lo, n_hi)):
        xn = x['glms'][n]
        xn['bias']['bias'] = Xh[i, 0:1].copy()
        if D > 0:
            xn['bkgd']['w_stim'] = Xh[i, 1:1 + D].copy()
        xn['imp']['w_ir'] = Xh[i, 1 + D:].copy()
    population.last_fit_stats = {'iterations': it, 'evaluations': n_evals[0],
                                 'neuron_evaluations': neuron_evals[0],
                                 'converged_gtol': n_conv, 'stalled': n_frozen,
                                 'maxiter': M - n_conv - n_frozen}
    return f.cpu().numpy(), it, n_evals[0]
