"""
Sharding one population over the GPUs of a node -- the data-parallel pattern of
pyglm/inference/parallel_coord_descent.py:137-147, parallel_gibbs.py:24-37, 162-165 and
utils/parallel_util.py:16-79, 154-183 (push the data to every engine once, map over the
post-synaptic neuron index, gather the results) on torch.distributed: backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in CPU tests.  One process per GPU.

  X1  broadcast_data        data dict (S as uint8, stim) from rank 0, once        parallel_util.py:154-183
  X2  allgather_rows(ll)    per-neuron ll / log p of the rank's shard             parallel_util.py:26-31
  X3  allgather_rows(rows)  packed parameter rows fitted by each rank             parallel_coord_descent.py:147
  X4  allgather_cols        columns of A and W resampled by each rank             parallel_gibbs.py:24-37
  all-reduce of the packed (ll, grad) block for the time-sharded evaluation (bench.py).

Rank r owns neurons [N*r/G, N*(r+1)/G); the spike data are replicated (every neuron's likelihood
needs all presynaptic trains).  Collectives run on the tensors' own device: with nccl the payload
never leaves HBM; gloo (CPU tests, single-GPU debug runs) stages through host memory itself.
"""
import numpy as np


def shard_bounds(N, rank, world):
    """Block partition of neurons: [lo, hi) of `rank`."""
    return (N * rank) // world, (N * (rank + 1)) // world


def all_shard_bounds(N, world):
    return [shard_bounds(N, r, world) for r in range(world)]


def _dist():
    import torch.distributed as dist
    return dist


def world_rank():
    """(world_size, rank) of the default process group; (1, 0) without one."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def alone():
    """True when there is nothing to exchange: one rank (or no process group).  PYGLM_COLLECTIVES_AT_WORLD1=1 runs
    the collectives on a one-rank group all the same -- how tests/test_gpu_parallel.py executes the RCCL branches
    (device-tensor all_reduce / all_gather_into_tensor / broadcast) on a one-GPU box."""
    import os
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get('PYGLM_COLLECTIVES_AT_WORLD1') != '1'


def _collective_tensor(t):
    """The tensor a collective of the current backend can take: nccl needs device tensors and gets
    them as they are; gloo gets host tensors."""
    dist = _dist()
    if dist.get_backend() == 'gloo' and t.is_cuda:
        return t.cpu()
    return t


def allgather_rows_t(local, N):
    """All-gather a (n_local, ...) tensor of this rank's neuron shard into the (N, ...) tensor on every
    rank, on the tensor's device.  Uneven shards are padded to the largest one (one collective)."""
    import torch
    dist = _dist()
    world, rank = world_rank()
    if alone():
        return local
    bounds = all_shard_bounds(N, world)
    width = max(hi - lo for lo, hi in bounds)
    tail = tuple(local.shape[1:])
    src = _collective_tensor(local)
    pad = torch.zeros((width,) + tail, dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    out = torch.empty((world, width) + tail, dtype=src.dtype, device=src.device)
    if src.is_cuda:
        dist.all_gather_into_tensor(out.view(-1), pad.reshape(-1).contiguous())
    else:
        dist.all_gather(list(out.unbind(0)), pad)
    full = torch.cat([out[r, :hi - lo] for r, (lo, hi) in enumerate(bounds)], dim=0)
    return full.to(local.device)


def allgather_rows(local, N, device=None):
    """numpy front end of allgather_rows_t (float64)."""
    import torch
    local = np.ascontiguousarray(local, dtype=np.float64)
    if alone():
        return local
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    return allgather_rows_t(t, N).cpu().numpy()


def allgather_cols(local_cols, N, device=None):
    """concatenate_parallel_updates (parallel_gibbs.py:24-37): rank r holds the columns
    [lo_r, hi_r) of an (N, N) matrix (shape (N, hi-lo)); returns the whole matrix on every rank."""
    rows = np.ascontiguousarray(np.asarray(local_cols, dtype=np.float64).T)      # (n_local, N)
    return np.ascontiguousarray(allgather_rows(rows, N, device).T)


def time_shard_bounds(nT, rank, world, align=16):
    """Time range [lo, hi) of `rank` when one evaluation is sharded over time bins (boundaries
    are multiples of `align` = the kernel's 16-row tile; the last shard takes the remainder).
    The likelihood is additive over time segments (population.py:41-43), so partial (ll, grad)
    of the ranges all-reduce (sum) to the full evaluation; every rank keeps the whole spike
    matrix because features reach R bins back across the shard boundary."""
    ntiles = (nT + align - 1) // align
    if ntiles < world:
        raise ValueError("cannot shard %d bins (%d tiles of %d) over %d ranks: a rank would get an "
                         "empty time range" % (nT, ntiles, align, world))
    lo = ((ntiles * rank) // world) * align
    hi = min(nT, ((ntiles * (rank + 1)) // world) * align) if rank + 1 < world else nT
    return lo, hi


def allreduce_sum_t(t):
    """In-place sum over all ranks of a tensor, on its device.  With PYGLM_CHECK_RANKS=1 the element count is first
    compared across ranks (one extra small all-reduce + host sync per call): a rank whose launch list has diverged
    fails with a message instead of hanging the job in a size-mismatched collective."""
    import os
    dist = _dist()
    if alone():
        return t
    if os.environ.get('PYGLM_CHECK_RANKS') == '1':
        import torch
        n = torch.tensor([t.numel(), -t.numel()], dtype=torch.int64, device=_collective_tensor(t).device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        hi, lo = int(n[0]), -int(n[1])
        if hi != lo:
            raise RuntimeError("all-reduce payloads differ across ranks: %d .. %d elements (rank %d has %d) -- the "
                               "lock-step launch lists have diverged" % (lo, hi, dist.get_rank(), t.numel()))
    c = _collective_tensor(t)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    if c is not t:
        t.copy_(c)
    return t


def allreduce_min_int(value, device=None):
    """The minimum of a host integer over all ranks (itself without a process group): how every rank of a time-sharded
    fit arrives at the SAME neuron-group size although the free device memory differs from rank to rank."""
    import torch
    dist = _dist()
    if alone():
        return int(value)
    on_dev = dist.get_backend() != 'gloo' and device is not None
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_dev else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t[0])


def allreduce_sum(local, device=None):
    """numpy front end of allreduce_sum_t (float64)."""
    import torch
    local = np.ascontiguousarray(local, dtype=np.float64)
    if alone():
        return local
    t = torch.from_numpy(local.copy())
    if device is not None:
        t = t.to(device)
    return allreduce_sum_t(t).cpu().numpy()


def broadcast_data(data, src=0, device=None):
    """X1: push the data dict from rank `src` to every rank (parallel_util.py:154-183 pushes the
    pickled dict to the engines one after the other -- "the bottleneck for large datasets").  Here the
    spike matrix travels once as uint8 counts (exact: <= 10 per bin, population.py:345-349; 77 MB at
    N=128, T=600 s instead of 614 MB of float64) in one broadcast on `device`, the stimulus as
    float64, the small entries as one pickled object.  Ranks other than `src` pass data=None."""
    import torch
    dist = _dist()
    world, rank = world_rank()
    if alone():
        return data
    meta = [None]
    if rank == src:
        S = np.asarray(data['S'])
        if S.dtype != np.uint8:
            if np.any(S < 0) or np.any(S > 255) or np.any(S != np.floor(S)):
                raise ValueError("spike counts must be integers in 0..255")
            S = S.astype(np.uint8)
        stim = data.get('stim', None)
        stim = None if stim is None else np.ascontiguousarray(stim, dtype=np.float64)
        small = dict((k, v) for k, v in data.items()
                     if k not in ('S', 'stim', 'fS', 'fstim', 'preprocessed', 'X') and not k.startswith('_'))
        meta = [{'S_shape': S.shape, 'stim_shape': None if stim is None else stim.shape, 'small': small}]
    dist.broadcast_object_list(meta, src=src)
    m = meta[0]
    dev = torch.device('cpu') if (device is None or dist.get_backend() == 'gloo') else torch.device(device)
    tS = torch.from_numpy(np.ascontiguousarray(S)).to(dev) if rank == src else \
        torch.empty(m['S_shape'], dtype=torch.uint8, device=dev)
    dist.broadcast(tS, src=src)
    out = dict(m['small'])
    out['S'] = tS.cpu().numpy()
    if m['stim_shape'] is not None:
        tq = torch.from_numpy(stim).to(dev) if rank == src else \
            torch.empty(m['stim_shape'], dtype=torch.float64, device=dev)
        dist.broadcast(tq, src=src)
        out['stim'] = tq.cpu().numpy()
    else:
        out['stim'] = None
    return out


def population_ll_grad_time_sharded(local_eval, nT, device=None):
    """(ll (N,), grad (N,P)) of the whole population with every rank evaluating only its
    time range: `local_eval(t_lo, t_hi)` returns the partial (ll, grad) of that range
    (DeviceGlm.set_time_range + ll_grad on the GPU; tests inject a CPU function)."""
    world, rank = world_rank()
    lo, hi = time_shard_bounds(nT, rank, world)
    ll, g = local_eval(lo, hi)
    packed = np.concatenate((np.asarray(ll, dtype=np.float64).reshape(-1, 1),
                             np.asarray(g, dtype=np.float64)), axis=1)
    tot = allreduce_sum(packed, device)
    return tot[:, 0], tot[:, 1:]


def population_ll(local_eval, N, device=None):
    """Population ll = sum_n ll_n with every rank evaluating only its shard (X2).
    `local_eval(lo, hi)` returns the (hi-lo,) per-neuron ll of this rank's shard
    (Population.compute_ll_vector on the GPU; tests inject a CPU function)."""
    world, rank = world_rank()
    lo, hi = shard_bounds(N, rank, world)
    ll_local = np.asarray(local_eval(lo, hi), dtype=np.float64).reshape(hi - lo)
    ll_all = allgather_rows(ll_local, N, device)
    return float(np.sum(ll_all)), ll_all


def gather_glm_params(local_rows, N, device=None):
    """x['glms'] = x_glms.get() (parallel_coord_descent.py:147): all-gather of the packed
    per-neuron parameter rows fitted by each rank (X3)."""
    return allgather_rows(local_rows, N, device)
