"""
Gibbs sampling with the post-synaptic neurons sharded over the GPUs of one node -- counterpart of
pyglm/inference/parallel_gibbs.py: the per-neuron updates (bias / stimulus / impulse HMC blocks,
collapsed network columns) are mapped over the engines (parallel_gibbs.py:162-165) and the sampled
pieces merged into one state (concatenate_parallel_updates, :24-37).  Here rank r advances the chains
of its neurons [N*r/G, N*(r+1)/G) on its own GPU -- HMC blocks in lock step over its shard, its
network columns in batched launches -- and one round of all-gathers per sweep merges

   * the packed GLM parameter rows (X3) and
   * the columns of A and W it resampled (X4).

Every rank ends a sweep with the same state.  Each rank draws from its own random stream
(seed + rank), like the reference's engines.
"""
import copy
import time

import numpy as np

from theano_pyglm_amd import parallel as PL
from theano_pyglm_amd.inference import gibbs as G
from theano_pyglm_amd.inference.parallel_coord_descent import (gather_glms, parallel_compute_log_p,
                                                               _device_of)


def gather_network_columns(population, x, lo, hi):
    """X4 (parallel_gibbs.py:24-37): column n of A and W comes from the rank that owns neuron n."""
    N = population.N
    net = x['net']
    dev = _device_of(population)
    if 'A' in net.get('graph', {}):
        A = np.asarray(net['graph']['A']).reshape(N, N)
        full = PL.allgather_cols(A[:, lo:hi], N, dev)
        net['graph']['A'] = np.rint(full).astype(A.dtype)
    if 'W' in net.get('weights', {}):
        W = np.asarray(net['weights']['W'], dtype=float).reshape(N, N)
        net['weights']['W'] = PL.allgather_cols(W[:, lo:hi], N, dev).ravel()
    return x


def parallel_gibbs_sample(population, N_samples=1000, x0=None, callback=None, seed=0, verbose=True):
    """parallel_gibbs.py:39-200.  Call on every rank with the same x0; returns the list of states."""
    N = population.N
    world, rank = PL.world_rank()
    lo, hi = PL.shard_bounds(N, rank, world)
    if x0 is None:
        raise ValueError("parallel_gibbs_sample needs the same x0 on every rank")
    rng = np.random.RandomState(seed + 7919 * rank)
    serial_updates, parallel_updates = G.initialize_updates(population, rng)
    x = x0
    x_smpls = [copy.deepcopy(x0)]
    start = time.time()
    for smpl in range(N_samples):
        if callback is not None:
            callback(x)
        lp, _ = parallel_compute_log_p(population, x)
        stop = time.time()
        if verbose and rank == 0:
            print("Gibbs iteration %d. Iter/s = %f. Log prob: %.3f" % (smpl, 1.0 / max(stop - start, 1e-9), lp))
        start = stop
        for upd in parallel_updates:
            if hi <= lo:
                continue
            if isinstance(upd, G.CollapsedGibbsNetworkColumnUpdate):
                upd.update_all(x, cols=np.arange(lo, hi))
            else:
                upd.update_range(x, lo, hi)
        gather_glms(population, x, lo, hi)
        gather_network_columns(population, x, lo, hi)
        for upd in serial_updates:
            upd.update(x)
        x_smpls.append(copy.deepcopy(x))
    return x_smpls
