"""
MAP coordinate descent with the post-synaptic neurons sharded over the GPUs of one node --
counterpart of pyglm/inference/parallel_coord_descent.py (and of utils/parallel_util.py's
engine setup), one process per GPU over torch.distributed (RCCL on the GPU box).

The reference pushes the data to every IPython engine (parallel_util.py:154-183), maps
`_parallel_fit_glm` over range(N) (parallel_coord_descent.py:137-139), gathers the fitted
per-neuron dicts (:147) and evaluates the population log p by mapping `_compute_glm_lp` over the
neurons and summing (:16-55).  Here rank r

  * receives the data once (parallel.broadcast_data, X1) and keeps them resident on ITS GPU,
  * fits the neurons [N*r/G, N*(r+1)/G) with the lock-step batched BFGS on its own GPU -- no
    communication at all during the fits (the per-neuron problems are independent),
  * all-gathers the fitted parameter rows at the end of the sweep (X3) and the per-neuron
    log posterior terms for the convergence test (X2); the network prior is evaluated on
    every rank from the replicated state (the reference does it on the master only, X5).

Every rank returns the same state dict.

`shard='time'` is the alternative split for populations whose per-rank neuron blocks would be narrow
(at N = 128 on 8 GPUs a rank owns 16 neurons and the MFMA tiles run a third full, docs/NOTEBOOK.md §4.1c): every
rank evaluates ALL neurons on its own range of time bins (`Population.set_time_shard`; the likelihood
is additive over time segments, population.py:41-43), one all-reduce of the packed (ll, grad) block per
evaluation replaces the end-of-sweep gathers, and all ranks run the identical lock-step optimizer on
the reduced values.
"""
import numpy as np

from theano_pyglm_amd import parallel as PL
from theano_pyglm_amd.inference import coord_descent as cd
from theano_pyglm_amd.inference.smart_init import initialize_with_data
from theano_pyglm_amd.utils.packvec import packdict, unpackdict, get_vars, set_vars


def _device_of(population):
    if PL.alone():
        return None
    import torch.distributed as dist
    return None if dist.get_backend() == 'gloo' else 'cuda:%d' % population.device


def parallel_compute_log_p(population, x, shard='neurons'):
    """parallel_coord_descent.py:16-55: log p = latent + network prior + sum_n (glm prior_n + ll_n), the
    per-neuron terms computed by the rank that owns the neuron and all-gathered (N doubles); with
    shard='time' every rank evaluates all neurons on its bins and the ll vector is all-reduced."""
    N = population.N
    world, rank = PL.world_rank()
    if shard == 'time' and not PL.alone():
        if world > 1 and population._time_shard != (rank, world):
            raise RuntimeError("parallel_compute_log_p(shard='time') needs Population.set_time_shard(rank, world) to be "
                               "active: without it every rank evaluates the whole recording and the all-reduce returns "
                               "world_size times the log likelihood")
        dev = _device_of(population)
        if dev is not None:
            # device path (RCCL): the per-neuron ll of every data sequence stays in HBM, is summed there and
            # all-reduced there; one small copy to the host at the end
            import torch
            with torch.cuda.device(dev):
                theta = torch.from_numpy(population.theta_matrix(x)).to(dev)
                Weff = torch.from_numpy(np.ascontiguousarray(population.W_eff(x), dtype=np.float64)).to(dev)
                tot = torch.zeros(N, dtype=torch.float64, device=dev)
                ll = torch.empty(N, dtype=torch.float64, device=dev)
                st = torch.cuda.Stream(dev)
                st.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(st):
                    for data in population.data_sequences:
                        population.set_data(data)
                        h = population._handle(data)
                        h.set_stream(st.cuda_stream)
                        try:
                            h.ll_grad_dev(theta.data_ptr(), Weff.data_ptr(), ll.data_ptr(), 0)
                            tot += ll
                        finally:
                            h.sync()
                            h.set_stream(None)
                    PL.allreduce_sum_t(tot)
                    st.synchronize()
                lp_n = tot.cpu().numpy()
        else:
            lp_n = np.zeros(N)
            for data in population.data_sequences:
                population.set_data(data)
                lp_n += population.compute_ll_vector(x)
            lp_n = PL.allreduce_sum(lp_n, None)
        for n in range(N):
            population._check_vars(x, n)
            lp_n[n] += population.glm.log_prior(x['glms'][n])
        lp = population.latent.log_p(x.get('latent', {})) + population.network.log_p(x['net'])
        return float(lp + np.sum(lp_n)), lp_n
    lo, hi = PL.shard_bounds(N, rank, world)
    lp_n = np.zeros(hi - lo)
    for i, n in enumerate(range(lo, hi)):
        population._check_vars(x, n)
        lp_n[i] = population.glm.log_prior(x['glms'][n])
    if hi > lo:
        for data in population.data_sequences:
            population.set_data(data)
            lp_n += population.compute_ll_vector(x, lo, hi)
    lp_all = PL.allgather_rows(lp_n, N, _device_of(population))
    lp = population.latent.log_p(x.get('latent', {})) + population.network.log_p(x['net'])
    return float(lp + np.sum(lp_all)), lp_all


def gather_glms(population, x, lo, hi):
    """X3: every rank ends up with the parameters of all N neurons; rank r contributes [lo, hi)."""
    N = population.N
    syms = population.glm_syms()
    shapes, rows = None, []
    v0, shapes = packdict(get_vars(syms, x['glms'][0]))
    for n in range(lo, hi):
        rows.append(packdict(get_vars(syms, x['glms'][n]))[0])
    rows = np.array(rows).reshape(hi - lo, v0.size)
    import torch
    t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64))
    dev = _device_of(population)
    if dev is not None:
        t = t.to(dev)                           # RCCL all-gather on the rank's own GPU; one copy back below
    full = PL.allgather_rows_t(t, N).cpu().numpy()
    for n in range(N):
        if not (lo <= n < hi):
            set_vars(syms, x['glms'][n], unpackdict(full[n].copy(), shapes))
    return x


def _time_sharded_sweeps(population, x, maxiter, atol, verbose):
    """shard='time': all ranks run the same lock-step BFGS; each evaluation = local bins + one all-reduce."""
    import torch
    from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
    world, rank = PL.world_rank()
    population.set_time_shard(rank, world)
    try:
        lp_prev, _ = parallel_compute_log_p(population, x, shard='time')
        net_inf_prms = cd.prep_first_order_network_inference(population)
        converged, it = False, 0
        while not converged and it < maxiter:
            it += 1
            fit_glms_batched_torch(population, x, reduce=PL.allreduce_sum_t)
            cd.fit_network(x, net_inf_prms)
            lp, _ = parallel_compute_log_p(population, x, shard='time')
            if verbose and rank == 0:
                print("Iteration %d: LP=%.2f. Change in LP: %.2f" % (it, lp, lp - lp_prev))
            converged = np.abs(lp - lp_prev) < atol
            lp_prev = lp
    finally:
        population.set_time_shard(None)
    return x


NARROW_SHARD = 64      # neurons per rank below which the time split is the default (a 16-neuron
                       # shard of a 128-neuron population runs at a third of the 128-wide MFMA rate)


def parallel_coord_descent(population, x0=None, maxiter=50, atol=1e-5, batched=None, verbose=False,
                           shard=None):
    """parallel_coord_descent.py:57-156.  Call on every rank of an initialised process group with the
    same x0 (e.g. drawn from the same seed); without a process group this is coord_descent.
    shard: 'neurons' (the reference's split), 'time' (see the module docstring; needs the GPU lock-step
    optimizer) or None = 'time' when a rank's neuron block would be narrower than NARROW_SHARD neurons
    and the lock-step optimizer serves the model, else 'neurons'."""
    N = population.N
    world, rank = PL.world_rank()
    batched = cd.resolve_batched(population, batched)
    if PL.alone():
        return cd.coord_descent(population, x0=x0, maxiter=maxiter, atol=atol, batched=batched, verbose=verbose)
    if shard is None:
        shard = 'time' if (batched == 'torch' and N // world < NARROW_SHARD) else 'neurons'
    if shard == 'time':
        if x0 is None:
            raise ValueError("parallel_coord_descent needs the same x0 on every rank")
        if batched != 'torch':
            raise ValueError("shard='time' runs the lock-step GPU optimizer (batched='torch')")
        initialize_with_data(population, population.data_sequences[-1], x0)
        return _time_sharded_sweeps(population, x0, maxiter, atol, verbose)
    lo, hi = PL.shard_bounds(N, rank, world)
    if x0 is None:
        raise ValueError("parallel_coord_descent needs the same x0 on every rank")
    initialize_with_data(population, population.data_sequences[-1], x0)
    x = x0
    lp_prev, _ = parallel_compute_log_p(population, x)
    if verbose and rank == 0:
        print("Initial LP=%.2f." % lp_prev)
    glm_inf_prms = cd.prep_first_order_glm_inference(population) if not batched else None
    net_inf_prms = cd.prep_first_order_network_inference(population)
    converged, it = False, 0
    while not converged and it < maxiter:
        it += 1
        if hi > lo:
            if batched == 'torch':
                from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
                fit_glms_batched_torch(population, x, n_lo=lo, n_hi=hi)
            elif batched:
                cd.fit_glms_batched(population, x, n_lo=lo, n_hi=hi)
            else:
                for n in range(lo, hi):
                    nvars = population.extract_vars(x, n)
                    cd.fit_glm(nvars, n, glm_inf_prms)
                    x['glms'][n] = nvars['glm']
        gather_glms(population, x, lo, hi)
        cd.fit_network(x, net_inf_prms)          # replicated (no-op for constant weights / complete graphs)
        lp, _ = parallel_compute_log_p(population, x)
        if verbose and rank == 0:
            print("Iteration %d: LP=%.2f. Change in LP: %.2f" % (it, lp, lp - lp_prev))
        converged = np.abs(lp - lp_prev) < atol
        lp_prev = lp
    return x
