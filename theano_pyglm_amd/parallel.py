"""
Neuron-sharded evaluation over the GPUs of one node -- the data-parallel pattern of
pyglm/inference/parallel_coord_descent.py:137-147 and utils/parallel_util.py:16-79
(map over post-synaptic neuron index, gather the results), on torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

One process per GPU.  Rank r owns neurons [N*r/G, N*(r+1)/G); the spike data are
replicated (every neuron's likelihood needs all presynaptic trains).  The only
data-path collectives are all-gathers of per-neuron scalars / parameter rows.
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


def allgather_rows(local, N, device=None):
    """All-gather a (n_local, ...) float64 array into the (N, ...) array, every rank.
    Uneven shards are padded to the largest shard (one collective)."""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = all_shard_bounds(N, world)
    width = max(hi - lo for lo, hi in bounds)
    tail = local.shape[1:]
    pad = np.zeros((width,) + tail)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.reshape(-1).contiguous()) \
        if hasattr(dist, 'all_gather_into_tensor') and t.device.type != 'cpu' \
        else dist.all_gather(list(out.unbind(0)), t)
    out = out.cpu().numpy()
    return np.concatenate([out[r, :hi - lo] for r, (lo, hi) in enumerate(bounds)], axis=0)


def time_shard_bounds(nT, rank, world, align=16):
    """Time range [lo, hi) of `rank` when one evaluation is sharded over time bins (boundaries
    are multiples of `align` = the kernel's 16-row tile; the last shard takes the remainder).
    The likelihood is additive over time segments (population.py:41-43), so partial (ll, grad)
    of the ranges all-reduce (sum) to the full evaluation; every rank keeps the whole spike
    matrix because features reach R bins back across the shard boundary."""
    ntiles = (nT + align - 1) // align
    lo = ((ntiles * rank) // world) * align
    hi = min(nT, ((ntiles * (rank + 1)) // world) * align) if rank + 1 < world else nT
    return lo, hi


def allreduce_sum(local, device=None):
    """Sum a float64 array over all ranks (RCCL all-reduce on the GPU box, gloo in tests)."""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    t = torch.from_numpy(local.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def population_ll_grad_time_sharded(local_eval, nT, device=None):
    """(ll (N,), grad (N,P)) of the whole population with every rank evaluating only its
    time range: `local_eval(t_lo, t_hi)` returns the partial (ll, grad) of that range
    (DeviceGlm.set_time_range + ll_grad on the GPU; tests inject a CPU function)."""
    dist = _dist()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = time_shard_bounds(nT, rank, world)
    ll, g = local_eval(lo, hi)
    packed = np.concatenate((np.asarray(ll, dtype=np.float64).reshape(-1, 1),
                             np.asarray(g, dtype=np.float64)), axis=1)
    tot = allreduce_sum(packed, device)
    return tot[:, 0], tot[:, 1:]


def population_ll(local_eval, N, device=None):
    """Population ll = sum_n ll_n with every rank evaluating only its shard.
    `local_eval(lo, hi)` returns the (hi-lo,) per-neuron ll of this rank's shard
    (Population.compute_ll_vector on the GPU; tests inject a CPU function)."""
    dist = _dist()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(N, rank, world)
    ll_local = np.asarray(local_eval(lo, hi), dtype=np.float64).reshape(hi - lo)
    ll_all = allgather_rows(ll_local, N, device)
    return float(np.sum(ll_all)), ll_all


def gather_glm_params(local_rows, N, device=None):
    """x['glms'] = x_glms.get() (parallel_coord_descent.py:147): all-gather of the packed
    per-neuron parameter rows fitted by each rank."""
    return allgather_rows(local_rows, N, device)
