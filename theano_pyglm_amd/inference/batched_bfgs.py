"""
Lock-step BFGS for all neurons with the optimizer state resident on the GPU.

The N per-neuron MAP problems of coord_descent (coord_descent.py:161-204, 243-247) are independent given
the network (neuron n's parameters only enter ll_n), so the whole sweep over n = 0..N-1
runs as one batched BFGS: per trial step one fused ll+grad launch on device pointers, and around it the
optimizer itself as HIP row kernels (pgl_bfgs_*: one workgroup per neuron) -- the algorithm scipy runs for
the reference (BFGS from H = I, More'-Thuente strong-Wolfe line search with scipy's constants and first
trial step, csrc/pglm_linesearch.h), so a neuron's iterates are those of its sequential scipy fit up to
rounding until a line search reports a warning (scipy stops there; see fit_glms_batched_torch); the inverse
Hessians stay implicit (update history) or, for tiny problems, dense (M x P x P, one pass per accepted iteration).  Nothing but the active flags crosses PCIe.  PyTorch is plumbing here (device memory,
and the chain rules / priors of the packings whose rows are not the device's own theta rows).

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

    def __init__(self, population, torch, handles=(), shard=None):
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
        # (pgl_info 'stim_path' == 2; the tap-rate kernels take neuron ranges only) -- on EVERY handle that will be
        # launched, for the whole shard and for a one-neuron list
        sepf = False
        if self.bk == 'st_sep' and handles:
            lo, hi = (0, self.N) if shard is None else shard
            sepf = all(h.info(a, b)['stim_path'] == 2 for h in handles for (a, b) in ((lo, hi), (lo, lo + 1)))
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
                           reduce=None, lag=None, init_scaling=False, max_trials=100, hessian_bytes=None, hessian=None):
    """In-place MAP fit of x['glms'][n_lo:n_hi]; returns (nlp (M,), iterations, evaluations).

    The dense inverse Hessians take M * P^2 doubles (C3: 0.4 GB; 512 neurons of a standard_glm: 27 GB; 2 048: 1.7 TB).
    When they would exceed `hessian_bytes` (default: 40 % of the free device memory) the shard is fitted in consecutive
    groups of neurons that fit -- the fits are independent, so the result is the same -- and the statistics are merged.

    `hessian`: 'dense' (M, P, P) matrices, one read-modify-write pass of 2 P^2 numbers per accepted iteration;
    'implicit' the history of update vectors, 4 k P numbers per product after k updates -- less than the dense pass
    while k <= P / 2, and fits rarely get that far (C3: 22 iterations; measured C3 0.112 -> 0.095 s, C2 11 -> 5 ms,
    N = 256 0.75 -> 0.49 s, C5 stress 0.205 -> 0.148 s).  None: implicit unless maxiter > 2 P (tiny problems, where
    the history could outgrow the matrix several times over), then dense.

    Everything runs on one dedicated torch stream that the device handles are switched to
    (pgl_set_stream).  One trial step of all unfinished neurons is TWO host calls: the fused ll+grad evaluation of the
    listed neurons (pgl_ll_grad_list_dev), and pgl_bfgs_step_dev for everything behind it -- priors and NaN rules,
    the line-search step, t = H g for the rows that took a step, the BFGS update, the next direction and the trial
    points of the next launch -- as ONE row kernel while the update history is short (k_bfgs_step<1024>; long histories:
    line search | k_bfgs_hdots | k_bfgs_hcomb | update).  The host never waits for the launch it has just queued: which
    neurons are still active reaches it through pinned memory the row kernel writes itself, read after the evaluation
    of the NEXT launch has been queued (`lag` more launches late if > 0; default 0): the active set only ever
    shrinks, so a launch over the stale (larger) list evaluates a few rows whose results the row kernels ignore.
    Neurons that have converged drop out of the launch list (pgl_ll_grad_list_dev evaluates an arbitrary list of
    neurons), so late, poorly conditioned neurons do not pay for the whole population.

    Every neuron runs its own state machine -- "line search at step alpha along p" -- and every launch
    evaluates the pending trial step of the listed neurons, whichever iteration each of them is in: a neuron
    whose step is accepted moves on without waiting for the others.  Up to the first line-search warning a neuron's
    iterates are scipy's (up to rounding); where scipy's BFGS gives up on a line search ("precision loss",
    coord_descent.py:194-199) the neuron takes the best sufficient-decrease step of that search if there is one,
    else restarts once from steepest descent, then is frozen.

    `init_scaling`: scale the identity of a (re)started inverse Hessian by s.y / y.y before the first update
    (Nocedal & Wright 6.20).  Off by default, like scipy's BFGS.

    `max_trials`: trial steps per line search (scipy's DCSRCH: 100).

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
    # groups of neurons whose inverse Hessians fit the memory budget
    Pp = _Packing(population, torch).Pp
    if hessian is None:
        hessian = 'implicit' if maxiter <= 2 * Pp else 'dense'
    if hessian not in ('dense', 'implicit'):
        raise ValueError("hessian: 'dense', 'implicit' or None")
    per_neuron = 8.0 * Pp * (Pp + (Pp & 1)) if hessian == 'dense' else 8.0 * (2.0 * Pp + 4.0) * maxiter
    if hessian_bytes is None:
        hessian_bytes = 0.4 * torch.cuda.mem_get_info(dev)[0]
    group = M if hessian_bytes >= per_neuron * M else max(1, int(hessian_bytes / per_neuron))
    if reduce is not None:
        # a time-sharded fit: every rank must run the identical sequence of launches with identically sized
        # all-reduces, but the free memory the default budget is taken from differs between ranks
        from theano_pyglm_amd import parallel as PL
        group = PL.allreduce_min_int(group, dev)
    if group < M:
        nlps, its, evs, stats = [], 0, 0, None
        for lo in range(n_lo, n_hi, group):
            hi = min(n_hi, lo + group)
            f_, it_, ev_ = fit_glms_batched_torch(population, x, maxiter, gtol, lo, hi, verbose, reduce, lag, init_scaling,
                                                  max_trials, hessian_bytes=float('inf'), hessian=hessian)
            nlps.append(f_)
            its, evs = max(its, it_), evs + ev_
            st = population.last_fit_stats
            if stats is None:
                stats = dict(st, per_neuron={k: list(v) for k, v in st['per_neuron'].items()}, groups=1)
            else:
                for k in ('evaluations', 'neuron_evaluations', 'line_search_steps', 'neuron_iterations', 'converged_gtol',
                          'stalled', 'maxiter', 'launch_cap'):
                    stats[k] += st[k]
                stats['iterations'] = max(stats['iterations'], st['iterations'])
                for k in stats['per_neuron']:
                    stats['per_neuron'][k] += list(st['per_neuron'][k])
                stats['groups'] += 1
        population.last_fit_stats = stats
        return np.concatenate(nlps), its, evs
    handles = []
    for data in population.data_sequences:
        population.set_data(data)
        handles.append(population._handle(data))
    # the optimiser's stream: one per device, kept between fits (a fresh stream costs its first wait ~6 ms: the hardware
    # queue behind it is created on first use)
    stream = _STREAMS.get(dev.index)
    if stream is None:
        stream = _STREAMS[dev.index] = torch.cuda.Stream(dev)
    stream.wait_stream(torch.cuda.current_stream(dev))        # whatever the caller queued comes first
    for h in handles:
        h.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            out = _lockstep_bfgs(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M,
                                 verbose, reduce, 0 if lag is None else max(0, int(lag)), init_scaling, max_trials, hessian)
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


_HOST_BUFFERS = {}
_STREAMS = {}


def _host_buffers(torch, M, nring):
    """Pinned flag buffers, events and list staging of a fit, kept between fits (a pinned allocation costs a driver call
    of ~0.1 ms, and coord_descent runs one fit per sweep): {(M, nring): (flags, events, staging)}."""
    key = (int(M), int(nring))
    hb = _HOST_BUFFERS.get(key)
    if hb is None:
        if len(_HOST_BUFFERS) > 16:
            _HOST_BUFFERS.clear()
        hb = ([torch.ones(M, dtype=torch.float64).pin_memory() for _ in range(nring)],
              [torch.cuda.Event() for _ in range(nring)],
              [torch.empty(3 * M, dtype=torch.int32).pin_memory() for _ in range(nring + 2)])
        _HOST_BUFFERS[key] = hb
    return hb


def _lockstep_bfgs(population, torch, dev, stream, handles, x, maxiter, gtol, n_lo, n_hi, M, verbose, reduce,
                   lag, init_scaling, max_trials, hessian='dense'):
    """The launch loop.  Per trial step: the fused ll+grad evaluation of the listed neurons on every data sequence
    (+ `reduce`), then ONE call -- pgl_bfgs_step_dev -- for everything behind it: priors, line-search step, t = H g,
    update, and the trial points of the next launch's list (one row kernel while the update history is short)."""
    from theano_pyglm_amd._lib import PglError
    pk = _Packing(population, torch, handles, (n_lo, n_hi))
    h0 = handles[0]
    P = pk.Pp                                                 # length of an optimisation row
    Pth = population.glm.P                                    # length of a device theta row
    ld = P + (P & 1)
    nst = h0.bfgs_state_doubles(M, P)
    st = torch.zeros(nst, dtype=torch.float64, device=dev)
    MP = M * P
    X, g = st[0:MP].view(M, P), st[MP:2 * MP].view(M, P)
    sc = st[15 * MP:].view(-1, M)
    f, iters, active, frozen, nfev = sc[0], sc[6], sc[8], sc[9], sc[16]
    if hessian == 'dense':
        # dense inverse Hessians: touched by k_bfgs_hmul only, written before they are read
        H = torch.empty((M, P, ld), dtype=torch.float64, device=dev)
        hist = coef = cb = None
    else:
        # the update history (s_j, H y_j) [M][maxiter][2][P] and its scalars (appended by the update phase before read)
        H = None
        hist = torch.empty((M, maxiter, 2, P), dtype=torch.float64, device=dev)
        coef = torch.empty((M, maxiter, 2), dtype=torch.float64, device=dev)
        cb = torch.empty((M, maxiter, 2), dtype=torch.float64, device=dev)
    X.copy_(torch.tensor(pk.pack(x, n_lo, n_hi), dtype=torch.float64, device=dev))
    Weff = torch.tensor(population.W_eff(x), dtype=torch.float64, device=dev)
    prm = pk.prior_params() if pk.identity else None
    n_evals, neuron_evals = [0], [0]
    # per data sequence one [ll | grad] block, reused by every launch (everything is ordered by the stream)
    bufs = [torch.empty(M * (1 + Pth), dtype=torch.float64, device=dev) for _ in handles]

    def evaluate(Xt, rows32, idx32, L):
        """The evaluation of the L rows Xt of the neurons idx32 (None: the whole shard).  Returns (f, g, prior): rows that
        ARE theta rows come back as the summed (ll, grad) of the data sequences with prior = the parameters the row
        kernel adds them with; other packings as f = -(log prior + ll) and its gradient (fit_glm's NaN rules applied), prior None."""
        scatter = idx32 is not None and not pk.list_launch
        if scatter:
            # no neuron lists on this device path: the whole shard with the trial rows scattered into the current point
            Xe = X.index_copy(0, rows32.long(), Xt)
            th, ecnt, eidx = (Xe if pk.identity else pk.theta(Xe)), M, None
        else:
            th, ecnt, eidx = (Xt if pk.identity else pk.theta(Xt)), L, idx32
        tot = None
        for h, full in zip(handles, bufs):
            buf = full[:ecnt * (1 + Pth)]                      # [ll | grad]: one all-reduce
            if eidx is None:
                h.ll_grad_dev(th.data_ptr(), Weff.data_ptr(), buf.data_ptr(), buf[ecnt:].data_ptr(), n_lo, n_hi)
            else:
                h.ll_grad_list_dev(eidx.data_ptr(), ecnt, th.data_ptr(), Weff.data_ptr(), buf.data_ptr(),
                                   buf[ecnt:].data_ptr())
            if reduce is not None:
                reduce(buf)
            tot = buf if tot is None else tot.add_(buf)
        n_evals[0] += 1
        neuron_evals[0] += ecnt
        llt, Gth = tot[:ecnt], tot[ecnt:].view(ecnt, Pth)
        if scatter:
            llt, Gth = llt[rows32.long()].contiguous(), Gth[rows32.long()].contiguous()
        if pk.identity:                                       # priors + NaN rules: the row kernel's first phase
            return llt, Gth, prm
        lp, G = pk.prior(Xt)
        fv = -(lp + llt)
        gv = -(G + pk.chain(Xt, Gth))
        fv = torch.where(torch.isnan(fv), torch.full_like(fv, 1e16), fv)
        gv = torch.where(torch.isnan(gv).any(1)[:, None], torch.zeros_like(gv), gv)
        return fv.contiguous(), gv.contiguous(), None

    def evaluate_list(Xt, rows32, idx32, L):
        try:
            return evaluate(Xt, rows32, idx32, L)
        except PglError:
            # a list length whose launch plan the device path does not serve (probed in _Packing for the whole shard and
            # a single neuron only): from here on the whole shard with the trial rows scattered into the current point
            if idx32 is None or not pk.list_launch:
                raise
            pk.list_launch = False
            return evaluate(Xt, rows32, idx32, L)

    f0, g0, pr0 = evaluate(X, None, None, M)
    if pr0 is not None:
        h0.bfgs_objective_dev(M, P, X.data_ptr(), f0.data_ptr(), g0.data_ptr(), *pr0)
    f.copy_(f0)
    g.copy_(g0)
    h0.bfgs_init_dev(st.data_ptr(), M, P, gtol)
    Xts = [torch.empty((M, P), dtype=torch.float64, device=dev) for _ in range(2)]
    h0.bfgs_trial_dev(st.data_ptr(), M, P, 0, M, Xts[0].data_ptr())
    # a search takes at most max_trials steps, an iteration one search (+ one launch for a restart)
    max_launches = maxiter * (max_trials + 1) + 2
    nring = lag + 2
    ring, events, stage = _host_buffers(torch, M, nring)
    for fl in ring:
        fl.fill_(1.0)
    # a new launch list travels as ONE copy of [position of every row in it | its rows | its neurons] (int32)
    lists = torch.empty((2, 3 * M), dtype=torch.int32, device=dev)
    nlist = 0
    pending = {}                                              # launch -> (flags, event, rows of its list)
    rows_h = np.arange(M, dtype=np.int32)
    rows32, idx32, L = None, None, M
    launches, cut = 0, False
    while True:
        if launches >= max_launches:
            cut = True
            break
        launches += 1
        cur, nxt = Xts[(launches - 1) & 1], Xts[launches & 1]
        ft, gt, prior = evaluate_list(cur[:L], rows32, idx32, L)
        # the next launch's list, from the flags the step `lag` + 1 launches back has written (normally long done:
        # the host queues this launch's step while the GPU is busy with its evaluation)
        nrows32, nidx32, nL, nrows_h, pos = rows32, idx32, L, rows_h, None
        k = launches - 1 - lag
        if k in pending:
            fl, ev, lrows = pending.pop(k)
            ev.synchronize()
            alive = lrows[fl.numpy()[lrows] != 0.0]
            if alive.size == 0:
                break
            if alive.size < L:
                nrows_h, nL = alive, int(alive.size)
                sg = stage[nlist % len(stage)].numpy()         # (a slot is reused nring list changes later: long copied)
                sg[:M] = -1
                sg[alive] = np.arange(nL, dtype=np.int32)
                sg[M:M + nL] = alive
                sg[2 * M:2 * M + nL] = alive + n_lo
                dl = lists[nlist & 1]
                dl.copy_(stage[nlist % len(stage)], non_blocking=True)
                nlist += 1
                pos, nrows32, nidx32 = dl[:M], dl[M:M + nL], dl[2 * M:2 * M + nL]
            if verbose:
                print("batched BFGS launch %d: %d neurons active after launch %d, next list %d" % (launches, alive.size, k, nL))
        slot = launches % nring
        h0.bfgs_step_dev(st.data_ptr(), M, P, rows32.data_ptr() if rows32 is not None else 0, L, cur.data_ptr(),
                         ft.data_ptr(), gt.data_ptr(), prior, max_trials, gtol, maxiter, init_scaling,
                         hist.data_ptr() if hist is not None else 0, coef.data_ptr() if coef is not None else 0,
                         maxiter if hist is not None else 0, cb.data_ptr() if cb is not None else 0, launches - 1,
                         H.data_ptr() if H is not None else 0, ld if H is not None else 0,
                         pos.data_ptr() if pos is not None else 0, nxt.data_ptr(), ring[slot].data_ptr())
        events[slot].record(stream)
        pending[launches] = (ring[slot], events[slot], rows_h)
        rows32, idx32, L, rows_h = nrows32, nidx32, nL, nrows_h
    # one copy of the state's head (X, g) and one of its scalars; the statistics are host arithmetic
    Xg = st[:2 * MP].cpu().numpy()
    sch = sc.cpu().numpy()
    Xh, gh = Xg[:MP].reshape(M, P), Xg[MP:].reshape(M, P)
    fh, itv, acth, frozh, nfevh = sch[0], sch[6], sch[8], sch[9], sch[16]
    it = int(itv.max())
    gmax = np.abs(gh).max(axis=1)
    n_conv = int((gmax <= gtol).sum())
    n_frozen = int(((frozh != 0) & (gmax > gtol)).sum())
    n_cut = int((acth != 0).sum()) if cut else 0              # rows the launch cap cut off in the middle of a search
    pk.unpack(x, Xh, n_lo, n_hi)
    population.last_fit_stats = {'iterations': it, 'evaluations': n_evals[0],
                                 'neuron_evaluations': neuron_evals[0],
                                 'line_search_steps': int(nfevh.sum()),
                                 'neuron_iterations': int(itv.sum()),
                                 'converged_gtol': n_conv, 'stalled': n_frozen,
                                 'maxiter': M - n_conv - n_frozen - n_cut, 'launch_cap': n_cut,
                                 'bookkeeping': 'hip row kernels' + ('' if pk.identity else ' (priors / chain rule: torch)'),
                                 'line_search': "More'-Thuente strong Wolfe (scipy's DCSRCH constants)",
                                 'lag': lag, 'init_scaling': bool(init_scaling), 'inverse_hessian': hessian,
                                 'per_neuron': {'iterations': [int(v) for v in itv],
                                                'line_search_steps': [int(v) for v in nfevh]}}
    return fh.copy(), it, n_evals[0]
