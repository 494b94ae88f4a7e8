"""World-size-2 and world-size-8 gloo tests of the sharded paths (theano_pyglm_amd/parallel.py):
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

    # time-sharded evaluation: partial sums over this rank's bins, all-reduced
    from oracle import glm_oracle as O
    tcalls = []

    def local_time_eval(t_lo, t_hi):
        tcalls.append((t_lo, t_hi))
        Sf = p.S.astype(float)
        lls, gs = np.zeros(N), np.zeros((N, p.P))
        for n in range(N):
            th = p.theta[n]
            ll, gb, _, gw = O.glm_ll_grad(n, Sf[t_lo:t_hi], p.fS[t_lo:t_hi], th[1:].reshape(N, p.B),
                                          p.Weff[:, n], th[0], p.dt, p.kind)
            lls[n] = ll
            gs[n, 0] = gb
            gs[n, 1:] = gw.reshape(-1)
        return lls, gs

    ll_t, g_t = PL.population_ll_grad_time_sharded(local_time_eval, p.nT)
    # the neuron-group size of a time-sharded fit: every rank proposes what its free memory allows, all take the minimum
    assert PL.allreduce_min_int(100 + 7 * ((rank * 5) % world)) == 100
    out.put((rank, total, ll_all, calls, rows, ll_t, g_t, tcalls))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,N', [(2, 6), (2, 7), (8, 128), (8, 100)])
def test_sharded_population_ll(world, N):
    """Both shardings on `world` gloo ranks: neuron shards (north_star's split; N = 100 on 8 ranks is uneven: 12 or 13
    neurons per rank, padded all-gather) and time shards (all-reduce of the packed block); world = 8 with N = 128 is the
    shape of BASELINE config 3."""
    import torch.multiprocessing as mp
    from theano_pyglm_amd import parallel as PL
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    p = H.Problem(N, 700, H.std_ibasis(), seed=50, weighted=True)
    ll0, g0 = p.oracle_ll_grad()
    assert sorted(r[0] for r in res) == list(range(world))
    sizes = set()
    for rank, total, ll_all, calls, rows, ll_t, g_t, tcalls in res:
        assert np.allclose(ll_all, ll0, rtol=1e-13) and np.isclose(total, ll0.sum(), rtol=1e-13)
        assert calls == [((N * rank) // world, (N * (rank + 1)) // world)]      # only its own shard
        sizes.add(calls[0][1] - calls[0][0])
        assert np.array_equal(rows, p.theta)
        assert np.allclose(ll_t, ll0, rtol=1e-12) and H.rel_err(g_t, g0) < 1e-12
        assert tcalls == [PL.time_shard_bounds(700, rank, world)]
    if (world, N) == (8, 100):
        assert sizes == {12, 13}
    if world == 2:
        assert PL.time_shard_bounds(700, 0, 2) == (0, 352) and PL.time_shard_bounds(700, 1, 2) == (352, 700)


def test_shard_bounds_cover():
    from theano_pyglm_amd import parallel as PL
    for nT, G in ((600000, 8), (700, 2), (1000, 3), (17, 2)):
        b = [PL.time_shard_bounds(nT, r, G) for r in range(G)]
        assert b[0][0] == 0 and b[-1][1] == nT
        assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))
        assert all(lo % 16 == 0 for lo, hi in b)
    with pytest.raises(ValueError):
        PL.time_shard_bounds(17, 0, 4)                 # 2 tiles over 4 ranks: a rank would be empty
    for N, G in ((128, 8), (7, 3), (4, 8)):
        b = PL.all_shard_bounds(N, G)
        assert b[0][0] == 0 and b[-1][1] == N
        assert all(b[i][1] == b[i + 1][0] for i in range(G - 1))


def _worker_bcast(rank, world, port, out):
    import torch.distributed as dist
    from theano_pyglm_amd import parallel as PL
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    data = None
    if rank == 0:
        rng = np.random.RandomState(3)
        data = {'S': (rng.rand(500, 5) < 0.05).astype(float), 'N': 5, 'dt': 0.001, 'T': 0.5,
                'stim': rng.randn(5, 2), 'dt_stim': 0.1, 'vars': {'a': np.arange(3)},
                'fS': 'must not travel', '_private': object()}
    got = PL.broadcast_data(data, src=0)
    # X4: each rank holds its own columns of a matrix
    M = np.arange(25.0).reshape(5, 5)
    lo, hi = PL.shard_bounds(5, rank, world)
    full = PL.allgather_cols(M[:, lo:hi], 5)
    import torch
    t = torch.full((3,), float(rank + 1), dtype=torch.float64)
    PL.allreduce_sum_t(t)
    rows = PL.allgather_rows_t(torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.ones(1, 2, dtype=torch.float64), 5)
    # PYGLM_CHECK_RANKS: equal payloads pass, diverged payloads raise on every rank instead of hanging
    os.environ['PYGLM_CHECK_RANKS'] = '1'
    t2 = torch.full((4,), 1.0, dtype=torch.float64)
    PL.allreduce_sum_t(t2)
    assert np.array_equal(t2.numpy(), np.full(4, float(world)))
    try:
        PL.allreduce_sum_t(torch.zeros(3 + rank, dtype=torch.float64))
        diverged = False
    except RuntimeError as e:
        diverged = 'diverged' in str(e)
    assert diverged
    os.environ['PYGLM_CHECK_RANKS'] = '0'
    out.put((rank, got, full, t.numpy(), rows.numpy()))
    dist.destroy_process_group()


def test_broadcast_data_and_column_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bcast, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    rng = np.random.RandomState(3)
    S = (rng.rand(500, 5) < 0.05).astype(float)
    stim = rng.randn(5, 2)
    for rank, got, full, t, rows in res:
        assert got['S'].dtype == np.uint8 and np.array_equal(got['S'], S)      # counts travel as uint8
        assert np.array_equal(got['stim'], stim)
        assert got['N'] == 5 and got['dt'] == 0.001 and got['dt_stim'] == 0.1 and got['T'] == 0.5
        assert np.array_equal(got['vars']['a'], np.arange(3))
        assert 'fS' not in got and '_private' not in got
        assert np.array_equal(full, np.arange(25.0).reshape(5, 5))
        assert np.array_equal(t, np.full(3, 3.0))
        assert np.array_equal(rows[:, 0], np.arange(5.0)) and rows.shape == (5, 2)
