"""
Lock-step BFGS for all neurons with the optimizer state resident on the GPU.

The N per-neuron MAP problems of coord_descent (coord_descent.py:161-204, 243-247) are independent given
the network (neuron n's parameters only enter ll_n), so the whole sweep over n = 0..N-1
runs as one batched BFGS: per trial point one fused ll+grad launch on device pointers; the priors, the
chain rules between the model's own variables and the flat feature weights the device consumes, and the
dense inverse-Hessian updates (M x P x P) are torch tensor ops on the same GPU -- nothing but a few scalars
crosses PCIe.  PyTorch is plumbing here (device memory + batched BLAS); the likelihood and its gradient
come from the HIP kernels.

Every packing of coord_descent's per-neuron vector that the scoped models produce is served
(`_Packing`): [bias, w_stim, w_ir] (standard_glm: the vector IS the device's theta row),
[bias, w_t, w_x, w_ir] (spatiotemporal_glm, bkgd.py:214-227: theta carries vec(w_t (x) w_x), or -- wide
stimuli -- [w_t, w_x] themselves on the separable device path), and Dirichlet impulses
[bias, .., g_0 .. g_{N-1}] (impulse.py:286-291: theta carries |g| / sum|g|).

NaN semantics of fit_glm are kept: nll NaN -> 1e16, NaN gradient -> 0 (coord_descent.py:170-182).
"""
import numpy as np

from theano_pyglm_amd.components.bkgd import NoStimulus, BasisStimulus, SpatiotemporalStimulus
from theano_pyglm_amd.components.impulse import LinearBasisImpulses, DirichletImpulses
from theano_pyglm_amd.components.priors import Gaussian, GroupLasso


def supported(population):
    glm = population.glm
    if not isinstance(glm.bkgd_model, (NoStimulus, BasisStimulus, SpatiotemporalStimulus)):
        return False
    if isinstance(glm.imp_model, LinearBasisImpulses):
        return isinstance(glm.imp_model.prior, (Gaussian, GroupLasso))
    return isinstance(glm.imp_model, DirichletImpulses)


class _Packing(object):
    """The per-neuron optimisation vector (the differentiable GLM variables of coord_descent.py:24-30:
    bias, bkgd, imp blocks) as rows of a torch matrix, its map to the device's flat feature weights
    (Glm.theta_row), the chain rule back (Glm.chain_grad) and the log prior with its gradient
    (Glm.log_prior) -- the torch twins of the numpy host components, for M neurons at once.
    BFGS started from H = I is equivariant under permutations of the coordinates, so the order of the
    blocks inside a row (here: bias, bkgd, imp in natural neuron order) does not change the iterates."""

    def __init__(self, population, torch):
        self.torch = torch
        glm = population.glm
        self.glm = glm
        self.N, self.B = population.N, glm.imp_model.B
        bk = glm.bkgd_model
        if isinstance(bk, SpatiotemporalStimulus):
            self.bk = 'st_sep' if bk.separable else 'st'
            self.Bt, self.Bx = bk.Bt, bk.Bx
            self.nbk = bk.Bt + bk.Bx
        elif isinstance(bk, BasisStimulus):
            self.bk, self.nbk = 'basis', bk.n_vars
        else:
            self.bk, self.nbk = 'none', 0
        self.dirichlet = isinstance(glm.imp_model, DirichletImpulses)
        self.Pp = 1 + self.nbk + self.N * self.B
        # pgl_ll_grad_list_dev with a separable stimulus: only when the device evaluates it at the frame rate
        # (pgl_info 'stim_path' == 2; the tap-rate kernels take neuron ranges only)
        sepf = False
        if self.bk == 'st_sep' and getattr(population, '_current', None) is not None:
            try:
                sepf = population._handle(population._current).info()['stim_path'] == 2
            except Exception:
                sepf = False
        self.list_launch = self.bk != 'st_sep' or sepf
        # the row IS the device's theta row and its prior is one of the forms the row kernels know
        # (pgl_bfgs_objective_dev): the whole state machine then runs as HIP row kernels.  Separable stimulus: the block
        # [w_t, w_x] under N(0, sigma) (bkgd.py:223-224 with the template's mu = 0)
        self.identity = (self.bk in ('none', 'basis') or (self.bk == 'st_sep' and sepf and float(bk.mu) == 0.0)) and \
            not self.dirichlet and isinstance(glm.imp_model.prior, (Gaussian, GroupLasso))
        self.stim_sigma = float(bk.sigma) if self.bk == 'st_sep' else 0.01

    def prior_params(self):
        """(kind, mu_b, sg_b, stim_sigma, mu, sigma, lam) for pgl_bfgs_objective_dev."""
        glm = self.glm
        pr = glm.imp_model.prior
        kind = 1 if isinstance(pr, GroupLasso) else 0
        return (kind, float(glm.bias_model.mu_bias), float(glm.bias_model.sig_bias), self.stim_sigma, float(pr.mu),
                float(pr.sigma), float(getattr(pr, 'lam', 0.0)))

    # -- state dict <-> rows -------------------------------------------------------------------
    def pack(self, x, n_lo, n_hi):
        rows = []
        for n in range(n_lo, n_hi):
            xn = x['glms'][n]
            parts = [np.asarray(xn['bias']['bias'], float).reshape(-1)]
            if self.bk == 'basis':
                parts.append(np.asarray(xn['bkgd']['w_stim'], float).reshape(-1))
            elif self.bk in ('st', 'st_sep'):
                parts += [np.asarray(xn['bkgd']['w_t'], float).reshape(-1),
                          np.asarray(xn['bkgd']['w_x'], float).reshape(-1)]
            if self.dirichlet:
                parts += [np.asarray(xn['imp']['g_%d' % m], float).reshape(-1) for m in range(self.N)]
            else:
                parts.append(np.asarray(xn['imp']['w_ir'], float).reshape(-1))
            rows.append(np.concatenate(parts))
        return np.array(rows)

    def unpack(self, x, Xh, n_lo, n_hi):
        o = 1 + self.nbk
        for i, n in enumerate(range(n_lo, n_hi)):
            xn = x['glms'][n]
            xn['bias']['bias'] = Xh[i, 0:1].copy()
            if self.bk == 'basis':
                xn['bkgd']['w_stim'] = Xh[i, 1:o].copy()
            elif self.bk in ('st', 'st_sep'):
                xn['bkgd']['w_t'] = Xh[i, 1:1 + self.Bt].copy()
                xn['bkgd']['w_x'] = Xh[i, 1 + self.Bt:o].copy()
            if self.dirichlet:
                g = Xh[i, o:].reshape(self.N, self.B)
                for m in range(self.N):
                    xn['imp']['g_%d' % m] = g[m].copy()
            else:
                xn['imp']['w_ir'] = Xh[i, o:].copy()

    # -- rows -> theta rows, and the chain rule back ------------------------------------------------
    def theta(self, X):
        torch = self.torch
        o = 1 + self.nbk
        parts = [X[:, 0:1]]
        if self.bk == 'st':                                        # vec(w_t (x) w_x), index bt*Bx+bx
            wt, wx = X[:, 1:1 + self.Bt], X[:, 1 + self.Bt:o]
            parts.append((wt[:, :, None] * wx[:, None, :]).reshape(X.shape[0], -1))
        elif self.nbk:
            parts.append(X[:, 1:o])
        if self.dirichlet:                                         # |g| / sum|g|  (impulse.py:286-291)
            ga = X[:, o:].reshape(-1, self.N, self.B).abs()
            parts.append((ga / ga.sum(2, keepdim=True)).reshape(X.shape[0], -1))
        else:
            parts.append(X[:, o:])
        return torch.cat(parts, dim=1).contiguous()

    def chain(self, X, Gth):
        """d/d(rows) from d/d(theta rows)."""
        torch = self.torch
        o = 1 + self.nbk
        M = X.shape[0]
        parts = [Gth[:, 0:1]]
        D = self.glm.Dstim
        if self.bk == 'st':
            wt, wx = X[:, 1:1 + self.Bt], X[:, 1 + self.Bt:o]
            Gs = Gth[:, 1:1 + D].reshape(M, self.Bt, self.Bx)
            parts += [(Gs * wx[:, None, :]).sum(2), (Gs * wt[:, :, None]).sum(1)]
        elif self.nbk:
            parts.append(Gth[:, 1:1 + D])
        Gw = Gth[:, 1 + D:]
        if self.dirichlet:      # d beta_b / d g_c = sign(g_c) (delta_bc s - |g_b|) / s^2, s = sum|g|
            g = X[:, o:].reshape(M, self.N, self.B)
            ga = g.abs()
            sm = ga.sum(2, keepdim=True)
            gb = Gw.reshape(M, self.N, self.B)
            inner = (gb * ga).sum(2, keepdim=True)
            parts.append((torch.sign(g) * (gb * sm - inner) / sm ** 2).reshape(M, -1))
        else:
            parts.append(Gw)
        return torch.cat(parts, dim=1)

    # -- log prior (M,) and its gradient (M, Pp) ---------------------------------------------------
    def prior(self, X):
        torch = self.torch
        glm = self.glm
        o = 1 + self.nbk
        M = X.shape[0]
        b = X[:, 0]
        mu_b, sg_b = float(glm.bias_model.mu_bias), float(glm.bias_model.sig_bias)
        lp = -0.5 / sg_b ** 2 * (b - mu_b) ** 2                                   # bias.py:33
        G = torch.zeros_like(X)
        G[:, 0] = -(b - mu_b) / sg_b ** 2
        if self.bk == 'basis':                                                    # bkgd.py:76
            ws = X[:, 1:o]
            lp = lp - 0.5 / (0.01 ** 2) * (ws ** 2).sum(1)
            G[:, 1:o] = -ws / (0.01 ** 2)
        elif self.bk in ('st', 'st_sep'):                                         # bkgd.py:223-224
            mu, sg = float(glm.bkgd_model.mu), float(glm.bkgd_model.sigma)
            ws = X[:, 1:o]
            lp = lp - 0.5 / sg ** 2 * ((ws - mu) ** 2).sum(1)
            G[:, 1:o] = -(ws - mu) / sg ** 2
        w = X[:, o:].reshape(M, self.N, self.B)
        if self.dirichlet:                                                        # impulse.py:320-322
            al = float(glm.imp_model.alpha)
            lp = lp + ((al - 1.0) * torch.log(w.abs()).sum(2) - w.abs().sum(2)).sum(1)
            G[:, o:] = ((al - 1.0) / w - torch.sign(w)).reshape(M, -1)
            return lp, G
        pr = glm.imp_model.prior
        if isinstance(pr, GroupLasso):                                            # priors.py:202
            z = (w - float(pr.mu)) / float(pr.sigma)
            nrm = torch.sqrt((z ** 2).sum(2, keepdim=True))
            lp = lp - float(pr.lam) * nrm.sum((1, 2))
            G[:, o:] = (-float(pr.lam) * z / nrm / float(pr.sigma)).reshape(M, -1)   # 0/0 -> NaN
        elif isinstance(pr, Gaussian):                                            # priors.py:139
            lp = lp - 0.5 / float(pr.sigma) ** 2 * ((w - float(pr.mu)) ** 2).sum((1, 2))
            G[:, o:] = (-(w - float(pr.mu)) / float(pr.sigma) ** 2).reshape(M, -1)
        else:
            raise Exception("unsupported impulse prior for the batched GPU optimizer")
        return lp, G


def fit_glms_batched_torch(population, x, maxiter=225, gtol=1e-5, n_lo=0, n_hi=None, verbose=False,
                           reduce=None, lag=None, init_scaling=False, row_kernels=True):
    """In-place MAP fit of x['glms'][n_lo:n_hi]; returns (nlp (M,), iterations, evaluations).

    Everything runs on one dedicated torch stream that the device handles are switched to
    (pgl_set_stream): prior terms, the fused ll+grad launches, the line-search bookkeeping and the
    inverse-Hessian updates are ordered by the stream, and the host never waits for the launch it has
    just queued.  What steers the loop -- which neurons are still active -- reaches the host `lag`
    launches late (default: 1 with the HIP row kernels, 2 with framework tensor ops) through pinned memory: the active set only ever shrinks, so a launch over the stale
    (larger) list evaluates a few rows whose results are masked out on the device, and the host keeps
    queueing torch ops while the GPU is busy with the previous evaluations.  Neurons that have converged
    drop out of the launch list (pgl_ll_grad_list_dev evaluates an arbitrary list of neurons), so late,
    poorly conditioned neurons do not pay for the whole population.  A neuron whose backtracking fails
    restarts once from steepest descent before it is frozen (scipy's BFGS stops there with "precision
    loss", coord_descent.py:194-199).

    `row_kernels`: for rows that are the device's own theta rows (standard_glm-like models) the line-search and
    update bookkeeping runs as HIP row kernels (_lockstep_bfgs_rows); False keeps it in framework tensor ops
    (same iterates; the path every other packing takes).

    `init_scaling`: scale the identity of a (re)started inverse Hessian by s.y / y.y before the first update
    (Nocedal & Wright 6.20).  Off by default, like scipy's BFGS: measured on the named configurations it cuts the
    evaluations of the badly scaled spatiotemporal_glm (impulse prior precision 1e6 next to O(1e3) curvatures) from
    4 700 to 250 per 225 iterations, but costs standard_glm 5x the iterations (C3: 22 -> 121).

    `reduce`: optional callable applied in place to the packed device tensor [ll | grad] of every
    evaluation before the priors are added -- the all-reduce of a time-sharded multi-GPU fit (every rank
    evaluates its own bins of all neurons and runs the identical optimizer on the reduced values; the
    launch lists are functions of the reduced values and of the fixed lag, hence identical on all ranks)."""
    import torch
    if not supported(population):
        raise Exception("batched GPU BFGS: unsupported stimulus / impulse model (see batched_bfgs.supported)")
    N = population.N
    n_hi = N if n_hi is None else n_hi
    M = n_hi - n_lo
    dev = torch.device('cuda', population.device)
    handles = []
    for data in population.data_sequences:
        population.set_data(data)
        handles.append(population._handle(data))
    stream = torch.cuda.Stream(dev)
    stream.wait_stream(torch.cuda.current_stream(dev))        # whatever the caller queued comes first
    for h in handles:
        h.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            out = _lockstep_bfgs(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M,
                                 verbose, reduce, max(0, int(lag)) if lag is not None else None, init_scaling,
                                 row_kernels)
            stream.synchronize()
    finally:
        # also on the error path: kernels still queued on `stream` read the optimizer state and the handles' scratch;
        # let them finish before the caching allocator reuses the tensors and the handles go back to their own streams
        try:
            stream.synchronize()
        except Exception:
            pass
        for h in handles:
            h.set_stream(None)
    return out


def _lockstep_bfgs_rows(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose, reduce,
                        lag, pk):
    """The same state machines as _lockstep_bfgs with all row-wise bookkeeping in HIP row kernels
    (pgl_bfgs_trial / objective / accept / update_dev, one workgroup per neuron row): per launch of trial points
    ~10 kernels -- trial, fused ll+grad, objective, accept, batched GEMV, update, batched rank-3 update, identity
    rows, the copy of the active flags -- instead of ~110 framework kernels of 4 us each.  For rows that are the
    device's own theta rows [bias, w_stim, w_ir] (standard_glm-like models)."""
    h0 = handles[0]
    P = pk.Pp
    nst = h0.bfgs_state_doubles(M, P)
    st = torch.zeros(nst, dtype=torch.float64, device=dev)
    MP = M * P

    def vec(i):
        return st[i * MP:(i + 1) * MP].view(M, P)
    X, g, p, Hg, t_ = vec(0), vec(1), vec(2), vec(3), vec(6)
    U, V = st[7 * MP:10 * MP].view(M, P, 3), st[10 * MP:13 * MP].view(M, P, 3)
    sc = st[13 * MP:].view(12, M)
    f, alpha, slope, scale, iters, active, frozen = sc[0], sc[1], sc[2], sc[4], sc[5], sc[8], sc[9]
    X.copy_(torch.tensor(pk.pack(x, n_lo, n_hi), dtype=torch.float64, device=dev))
    Weff = torch.tensor(population.W_eff(x), dtype=torch.float64, device=dev)
    prm = pk.prior_params()
    n_evals, neuron_evals = [0], [0]

    def evaluate(Xt, idx32, L):
        """f and g (fit_glm's NaN rules applied) of the L rows Xt for the neurons idx32 (None: the whole shard);
        returned as views of one block [f | g]."""
        tot = None
        for h in handles:
            buf = torch.empty(L * (1 + P), dtype=torch.float64, device=dev)      # [ll | grad]: one all-reduce
            if idx32 is None:
                h.ll_grad_dev(Xt.data_ptr(), Weff.data_ptr(), buf.data_ptr(), buf[L:].data_ptr(), n_lo, n_hi)
            else:
                h.ll_grad_list_dev(idx32.data_ptr(), L, Xt.data_ptr(), Weff.data_ptr(), buf.data_ptr(),
                                   buf[L:].data_ptr())
            if reduce is not None:
                reduce(buf)
            tot = buf if tot is None else tot.add_(buf)
        h0.bfgs_objective_dev(L, P, Xt.data_ptr(), tot.data_ptr(), tot[L:].data_ptr(), *prm)
        n_evals[0] += 1
        neuron_evals[0] += L
        return tot[:L], tot[L:].view(L, P)

    f0, g0 = evaluate(X, None, M)
    f.copy_(f0)
    g.copy_(g0)
    Hg.copy_(g0)
    p.copy_(-g0)
    slope.copy_(-(g0 * g0).sum(1))
    alpha.copy_(torch.clamp(1.0 / g0.norm(dim=1).clamp_min(1e-300), max=1.0))
    active.copy_((g0.abs().amax(1) > gtol).to(torch.float64))
    H = torch.eye(P, dtype=torch.float64, device=dev).repeat(M, 1, 1)
    max_launches = maxiter * 31 + 2
    ring = [torch.empty(M, dtype=torch.float64).pin_memory() for _ in range(lag + 1)]
    pending = []

    def publish(k):
        hb = ring[k % (lag + 1)]
        hb.copy_(active, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        pending.append((hb, ev))

    publish(0)
    rows32, idx32, L = None, None, M
    launches = 0
    while launches < max_launches:
        if len(pending) > lag:
            hb, ev = pending.pop(0)
            ev.synchronize()                                    # a launch `lag` back: normally long done
            act_h = hb.numpy() != 0.0
            n_act = int(act_h.sum())
            if n_act == 0:
                break
            if n_act < L:
                idx_h = np.nonzero(act_h)[0].astype(np.int32)
                L = n_act
                rows32 = torch.from_numpy(idx_h).to(dev, non_blocking=True)
                idx32 = rows32 + n_lo
            if verbose:
                print("batched BFGS launch %d: %d neurons active %d launches ago, list of %d"
                      % (launches, n_act, lag, L))
        launches += 1
        Xt = torch.empty((L, P), dtype=torch.float64, device=dev)
        rp = rows32.data_ptr() if rows32 is not None else 0
        h0.bfgs_trial_dev(st.data_ptr(), M, P, rp, L, Xt.data_ptr())
        ft, gt = evaluate(Xt, idx32, L)
        h0.bfgs_accept_dev(st.data_ptr(), M, P, rp, L, Xt.data_ptr(), ft.data_ptr(), gt.data_ptr())
        torch.bmm(H, g.unsqueeze(2), out=t_.unsqueeze(2))       # t = H g_new: the one full read of H per launch
        h0.bfgs_update_dev(st.data_ptr(), M, P, gtol, maxiter)
        H.baddbmm_(U, V.transpose(1, 2))                        # H += U V^T (zero factors for rows without an update)
        h0.reset_identity_dev(H.data_ptr(), scale.data_ptr(), M, P)
        publish(launches)
    it = int(iters.max())
    gmax = g.abs().amax(1)
    n_conv = int((gmax <= gtol).sum())
    n_frozen = int(((frozen != 0) & (gmax > gtol)).sum())
    pk.unpack(x, X.cpu().numpy(), n_lo, n_hi)
    population.last_fit_stats = {'iterations': it, 'evaluations': n_evals[0],
                                 'neuron_evaluations': neuron_evals[0],
                                 'converged_gtol': n_conv, 'stalled': n_frozen,
                                 'maxiter': M - n_conv - n_frozen, 'bookkeeping': 'hip row kernels',
                                 'lag': lag, 'init_scaling': False}
    return f.cpu().numpy(), it, n_evals[0]


def _lockstep_bfgs(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose,
                   reduce=None, lag=2, init_scaling=False, row_kernels=True):
    pk = _Packing(population, torch)
    if row_kernels and pk.identity and not init_scaling:
        # (queueing a launch takes ~10 library calls here: one launch of run-ahead keeps the GPU fed)
        return _lockstep_bfgs_rows(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose,
                                   reduce, 1 if lag is None else lag, pk)
    lag = 2 if lag is None else lag
    X = torch.tensor(pk.pack(x, n_lo, n_hi), dtype=torch.float64, device=dev)
    Pp = X.shape[1]
    P = population.glm.P
    Weff = torch.tensor(population.W_eff(x), dtype=torch.float64, device=dev)
    n_evals = [0]
    neuron_evals = [0]
    rows_all = torch.arange(M, device=dev)

    def evaluate(Xt, rows, Xfull=None):
        """nlp and its gradient at the rows Xt (cnt, Pp) of the neurons `rows` (None: the whole shard)."""
        cnt = Xt.shape[0]
        lp, G = pk.prior(Xt)
        if rows is not None and not pk.list_launch:
            # no neuron lists on this device path: evaluate the whole shard with the trial rows scattered
            # into the current point
            Xe = Xfull.index_copy(0, rows, Xt)
            th, idx, ecnt = pk.theta(Xe), None, M
        else:
            th = pk.theta(Xt)
            idx = None if rows is None else (rows + n_lo).to(torch.int32).contiguous()
            ecnt = cnt
        Gth = torch.zeros((ecnt, P), dtype=torch.float64, device=dev) if len(handles) > 1 else None
        llt = None
        for h in handles:
            buf = torch.empty(ecnt * (1 + P), dtype=torch.float64, device=dev)     # [ll | grad]: one all-reduce
            ll, gr = buf[:ecnt], buf[ecnt:].view(ecnt, P)
            if idx is None:
                h.ll_grad_dev(th.data_ptr(), Weff.data_ptr(), ll.data_ptr(), gr.data_ptr(), n_lo, n_hi)
            else:
                h.ll_grad_list_dev(idx.data_ptr(), ecnt, th.data_ptr(), Weff.data_ptr(), ll.data_ptr(),
                                   gr.data_ptr())
            if reduce is not None:
                reduce(buf)
            llt = ll if llt is None else llt + ll
            Gth = gr if Gth is None else Gth.add_(gr)
        if ecnt != cnt:
            llt, Gth = llt[rows], Gth[rows]
        lp = lp + llt
        G = G + pk.chain(Xt, Gth)
        n_evals[0] += 1
        neuron_evals[0] += ecnt
        f, g = -lp, -G
        f = torch.where(torch.isnan(f), torch.full_like(f, 1e16), f)
        bad = torch.isnan(g).any(1)
        g = torch.where(bad[:, None], torch.zeros_like(g), g)
        return f, g

    f, g = evaluate(X, None)
    eye = torch.eye(Pp, dtype=torch.float64, device=dev)
    H = eye.repeat(M, 1, 1)
    Hg = g.clone()                                              # H g, carried along: one pass over H per launch
    active = g.abs().amax(1) > gtol
    frozen = torch.zeros(M, dtype=torch.bool, device=dev)       # line search failed twice in a row
    restarts = torch.zeros(M, dtype=torch.int64, device=dev)
    iters = torch.zeros(M, dtype=torch.int64, device=dev)       # BFGS iterations of every neuron
    nhalf = torch.zeros(M, dtype=torch.int64, device=dev)       # step halvings of the current line search
    fresh = torch.ones(M, dtype=torch.bool, device=dev)         # H is still the identity of a (re)start

    def first_step(gg):
        return torch.clamp(1.0 / gg.norm(dim=1).clamp_min(1e-300), max=1.0)

    # Every neuron runs its own BFGS state machine -- "line search at step alpha along p" -- and every launch
    # evaluates the pending trial point of the listed neurons, whether that is the first trial of a new
    # iteration or a backtracking trial: a neuron whose trial succeeds moves on to its next iteration without
    # waiting for the neurons that still backtrack.  The iterates of a neuron are those of the iteration-
    # synchronous loop (its trial points do not depend on the others); the number of launches is the largest
    # number of trials any neuron needs instead of the sum over iterations of the per-iteration maximum.
    p = -Hg
    slope = (p * g).sum(1)
    alpha = first_step(g)
    max_launches = maxiter * 31 + 2
    # the active mask reaches the host `lag` launches late (pinned ring + events)
    ring = [torch.empty(M, dtype=torch.bool).pin_memory() for _ in range(lag + 1)]
    pending = []

    def publish(k):
        hb = ring[k % (lag + 1)]
        hb.copy_(active, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        pending.append((hb, ev))

    publish(0)
    rows, L = None, M                                           # launch list (None: all M rows) and its length
    launches = 0
    while launches < max_launches:
        if len(pending) > lag:
            hb, ev = pending.pop(0)
            ev.synchronize()                                    # a launch `lag` back: normally long done
            act_h = hb.numpy()
            n_act = int(act_h.sum())
            if n_act == 0:
                break
            if n_act < L:
                idx_h = np.nonzero(act_h)[0]
                L = n_act
                rows = torch.from_numpy(idx_h).to(dev, non_blocking=True)
            if verbose:
                print("batched BFGS launch %d: %d neurons active %d launches ago, list of %d"
                      % (launches, n_act, lag, L))
        launches += 1
        ridx = rows_all if rows is None else rows
        Xr, ar = X[ridx], alpha[ridx]
        Xt = Xr + ar[:, None] * p[ridx]
        ft, gt = evaluate(Xt, rows, X)
        ok = active[ridx] & (ft <= f[ridx] + 1e-4 * ar * slope[ridx])
        acc = torch.zeros(M, dtype=torch.bool, device=dev)
        acc[ridx] = ok
        fail = active & ~acc
        # ---- failed trials: halve the step; after 30 halvings restart once from steepest descent, then freeze
        alpha = torch.where(fail, alpha * 0.5, alpha)
        nhalf = torch.where(fail, nhalf + 1, nhalf)
        stalled = fail & (nhalf >= 30)
        # ---- accepted trials: BFGS update of H with (s, y), new direction
        Xn = X.index_copy(0, ridx, torch.where(ok[:, None], Xt, Xr))
        fn = f.index_copy(0, ridx, torch.where(ok, ft, f[ridx]))
        gn = g.index_copy(0, ridx, torch.where(ok[:, None], gt, g[ridx]))
        s = Xn - X
        y = gn - g
        sy = (s * y).sum(1)
        upd = acc & (sy > 1e-12)
        # first update after a (re)start from H = I: H <- (s.y / y.y) I before the update (Nocedal & Wright 6.20) --
        # BFGS from the identity learns one direction per iteration, and a prior of precision 1e6 on 192 of 199
        # coordinates (spatiotemporal_glm's impulse weights) would cost it ~200 iterations of 20 step halvings each
        first = upd & fresh
        gam = torch.where(first, sy / (y * y).sum(1).clamp_min(1e-300), torch.ones_like(sy))
        if init_scaling:
            handles[0].reset_identity_dev(H.data_ptr(), torch.where(first, gam, torch.zeros_like(gam)).data_ptr(), M, Pp)
            Hg = Hg * gam[:, None]
        fresh = fresh & ~upd
        # H g_new: the one full read of H (M x P x P) per launch; H y = H g_new - H g follows from it
        t = torch.bmm(H, gn[:, :, None])[:, :, 0]
        rho = torch.where(upd, 1.0 / sy.clamp_min(1e-300), torch.zeros_like(sy))
        Hy = torch.where(upd[:, None], t - Hg, torch.zeros_like(t))
        yHy = (y * Hy).sum(1)
        c = (1.0 + rho * yHy) * rho
        # H += c s s^T - rho (Hy s^T + s Hy^T) as ONE rank-3 update: a single read-modify-write of H
        # (rows without an update carry rho = 0: their H is rewritten unchanged)
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
        handles[0].reset_identity_dev(H.data_ptr(), reset.to(torch.float64).data_ptr(), M, Pp)   # only the flagged rows
        fresh = fresh | reset
        Hg = torch.where(reset[:, None], g, Hg)
        pn = torch.where(reset[:, None], -g, pn)
        sl = torch.where(reset, (pn * g).sum(1), sl)
        p = torch.where(newls[:, None], pn, p)
        slope = torch.where(newls, sl, slope)
        alpha = torch.where(newls, torch.where(again, first_step(g), torch.ones_like(alpha)), alpha)
        nhalf = torch.where(newls, torch.zeros_like(nhalf), nhalf)
        active = active & (~frozen) & (g.abs().amax(1) > gtol) & (iters < maxiter)
        publish(launches)
    it = int(iters.max())
    gmax = g.abs().amax(1)
    n_conv = int((gmax <= gtol).sum())
    n_frozen = int((frozen & (gmax > gtol)).sum())
    pk.unpack(x, X.cpu().numpy(), n_lo, n_hi)
    population.last_fit_stats = {'iterations': it, 'evaluations': n_evals[0],
                                 'neuron_evaluations': neuron_evals[0],
                                 'converged_gtol': n_conv, 'stalled': n_frozen,
                                 'maxiter': M - n_conv - n_frozen, 'bookkeeping': 'torch tensor ops',
                                 'lag': lag, 'init_scaling': bool(init_scaling)}
    return f.cpu().numpy(), it, n_evals[0]
