"""World-size-2 gloo test of the neuron-sharded path (theano_pyglm_amd/parallel.py):
each rank evaluates only its shard (here with the oracle injected as the local
evaluator -- the plumbing, not the kernel, is under test) and the all-gathered
population ll equals the single-process value."""
import os
import socket

import numpy as np
import pytest

from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, out):
    import torch.distributed as dist
    from theano_pyglm_amd import parallel as PL
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = H.Problem(N, 700, H.std_ibasis(), seed=50, weighted=True)
    calls = []

    def local_eval(lo, hi):
        calls.append((lo, hi))
        return p.oracle_ll_grad(lo, hi)[0]

    total, ll_all = PL.population_ll(local_eval, N)
    lo, hi = PL.shard_bounds(N, rank, world)
    rows = PL.gather_glm_params(p.theta[lo:hi], N)
    out.put((rank, total, ll_all, calls, rows))
    dist.destroy_process_group()


@pytest.mark.parametrize('N', [6, 7])
def test_sharded_population_ll(N):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = H.Problem(N, 700, H.std_ibasis(), seed=50, weighted=True)
    ll0 = p.oracle_ll_grad()[0]
    for rank, total, ll_all, calls, rows in res:
        assert np.allclose(ll_all, ll0, rtol=1e-13) and np.isclose(total, ll0.sum(), rtol=1e-13)
        assert calls == [((N * rank) // 2, (N * (rank + 1)) // 2)]      # only its own shard
        assert np.array_equal(rows, p.theta)


def test_shard_bounds_cover():
    from theano_pyglm_amd import parallel as PL
    for N, G in ((128, 8), (7, 3), (4, 8)):
        b = PL.all_shard_bounds(N, G)
        assert b[0][0] == 0 and b[-1][1] == N
        assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
