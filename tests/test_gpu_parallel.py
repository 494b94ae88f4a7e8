"""
World-size-2 runs of the PRODUCT multi-GPU path on one GPU: two fresh child processes (spawned, not
re-exec'ed) both drive the HIP library on cuda:0, collectives over gloo.  The sharded MAP sweep
(inference/parallel_coord_descent.py) and the sharded Gibbs sweep (inference/parallel_gibbs.py) must
leave every rank with the same state, equal to / consistent with the single-process result.
"""
import copy
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(procs, q, n, timeout=600):
    """n results from the workers' queue; fails as soon as a worker has died without delivering (a plain
    q.get(timeout) would sit out the whole timeout on the GPU box)."""
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < n:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [pr.exitcode for pr in procs if pr.exitcode not in (None, 0)]
            assert not dead, "worker exited with %s before delivering its result" % dead
            assert time.time() - t0 < timeout, "workers timed out"
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    return res


def _pack_state(popn, x):
    from theano_pyglm_amd.utils.packvec import packdict, get_vars
    syms = popn.glm_syms()
    rows = np.array([packdict(get_vars(syms, xn))[0] for xn in x['glms']])
    net = x['net']
    A = np.asarray(net['graph']['A'], float).ravel() if 'A' in net.get('graph', {}) else np.zeros(0)
    W = np.asarray(net['weights']['W'], float).ravel() if 'W' in net.get('weights', {}) else np.zeros(0)
    return rows, A, W


def _worker(rank, world, port, data0, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from theano_pyglm_amd import parallel as PL
    from theano_pyglm_amd.inference.parallel_coord_descent import parallel_coord_descent, parallel_compute_log_p
    from theano_pyglm_amd.inference.parallel_gibbs import parallel_gibbs_sample
    from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
    from theano_pyglm_amd.population import Population
    # X1: only rank 0 holds the data
    data = PL.broadcast_data(data0 if rank == 0 else None, src=0)
    N = data['N']
    popn = Population(make_model('standard_glm', N=N, dt=0.001), device=0)
    popn.add_data(data)
    x0 = popn.sample(np.random.RandomState(11))
    lp0, lp_n = parallel_compute_log_p(popn, x0)
    x = parallel_coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    lp1, _ = parallel_compute_log_p(popn, x)
    rows, _, _ = _pack_state(popn, x)
    # the time-sharded form: every rank evaluates all neurons on its own bins, all-reduce per evaluation
    xt = parallel_coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch', shard='time')
    lpt = popn.compute_log_p(xt)                     # full recording again (the shard is lifted on return)
    rows_t, _, _ = _pack_state(popn, xt)
    # ... and with a memory budget that differs between the ranks (room for 4 neurons' inverse Hessians on rank 0, for 2
    # on rank 1): the group size is the minimum over the ranks, so both run the same three groups of two neurons with
    # identically sized all-reduces (a rank-local split would hang or sum garbage)
    from theano_pyglm_amd.inference.batched_bfgs import fit_glms_batched_torch
    xg = copy.deepcopy(x0)
    P = popn.glm.P
    popn.set_time_shard(rank, world)
    try:
        fit_glms_batched_torch(popn, xg, reduce=PL.allreduce_sum_t, hessian='dense',
                               hessian_bytes=8.0 * P * (P + (P & 1)) * (4.5 if rank == 0 else 2.5))
        groups = popn.last_fit_stats.get('groups')
    finally:
        popn.set_time_shard(None)
    rows_g, _, _ = _pack_state(popn, xg)
    assert groups == 3, groups
    assert np.allclose(rows_g, rows_t, rtol=1e-4, atol=1e-5)
    # sharded Gibbs on the sparse_weighted_model
    m2 = make_model('sparse_weighted_model', N=N, dt=0.001)
    stabilize_sparsity(m2)
    pop2 = Population(m2, device=0)
    pop2.add_data(data)
    y0 = pop2.sample(np.random.RandomState(12))
    y0['net']['weights']['W'] = 0.2 * np.asarray(y0['net']['weights']['W'])
    smpls = parallel_gibbs_sample(pop2, N_samples=2, x0=copy.deepcopy(y0), seed=5, verbose=False)
    y = smpls[-1]
    lpy, _ = parallel_compute_log_p(pop2, y)
    out.put((rank, lp0, lp_n, lp1, rows, _pack_state(pop2, y), lpy, pop2.compute_log_p(y), lpt, rows_t))
    dist.destroy_process_group()


def test_world2_sharded_map_and_gibbs_on_one_gpu():
    import torch.multiprocessing as mp
    from theano_pyglm_amd.harness.generate_synth_data import make_dataset
    from theano_pyglm_amd.inference.coord_descent import coord_descent
    N = 6
    model, popn, data = make_dataset('standard_glm', N, 8.0, seed=21)
    clean = dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed'))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, clean if r == 0 else None, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = _collect(procs, q, len(procs))
    res.sort(key=lambda r: r[0])
    # single-process reference on the same data and the same x0
    x0 = popn.sample(np.random.RandomState(11))
    lp0 = popn.compute_log_p(x0)
    x_ref = coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    lp_ref = popn.compute_log_p(x_ref)
    rows_ref, _, _ = _pack_state(popn, x_ref)
    for rank, lp0_r, lp_n, lp1, rows, ystate, lpy, lpy_single, lpt, rows_t in res:
        assert np.isclose(lp0_r, lp0, rtol=1e-12)
        assert lp_n.shape == (N,)
        assert np.isclose(lp1, lp_ref, rtol=1e-9)
        # each rank ran its own 3-neuron batched BFGS: same optimum as the 6-neuron lock-step run
        assert np.allclose(rows, rows_ref, rtol=1e-4, atol=1e-5)
        assert np.isfinite(lpy) and np.isclose(lpy, lpy_single, rtol=1e-12)
        # time-sharded MAP: the same 6-neuron lock-step run on all-reduced values
        assert np.isclose(lpt, lp_ref, rtol=1e-9)
        assert np.allclose(rows_t, rows_ref, rtol=1e-4, atol=1e-5)
    # both ranks hold the same state after the sharded sweeps (MAP rows, Gibbs rows, A and W columns)
    assert np.array_equal(res[0][4], res[1][4])
    assert np.array_equal(res[0][9], res[1][9])           # identical optimizer trajectories on both ranks
    for a, b in zip(res[0][5], res[1][5]):
        assert np.array_equal(a, b)
    A = res[0][5][1]
    assert set(np.unique(A)) <= {0.0, 1.0}


def _worker_rccl(port, data0, out):
    """One rank, backend nccl (= RCCL), collectives forced on the one-rank group: every device-tensor collective
    of the product path executes on the GPU box."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['PYGLM_COLLECTIVES_AT_WORLD1'] = '1'
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from theano_pyglm_amd import parallel as PL
    from theano_pyglm_amd.inference.coord_descent import coord_descent
    from theano_pyglm_amd.inference.parallel_coord_descent import parallel_coord_descent, parallel_compute_log_p
    from theano_pyglm_amd.inference.parallel_gibbs import parallel_gibbs_sample
    from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
    from theano_pyglm_amd.population import Population
    assert not PL.alone() and dist.get_backend() == 'nccl'
    # raw collectives on device tensors (float64 payloads, the padded all-gather, the uint8 spike broadcast)
    t = torch.arange(12, dtype=torch.float64, device='cuda:0').reshape(4, 3)
    assert torch.equal(PL.allgather_rows_t(t.clone(), 4), t)
    assert torch.equal(PL.allreduce_sum_t(t.clone()), t)
    assert np.array_equal(PL.allgather_cols(np.arange(9.0).reshape(3, 3), 3, 'cuda:0'), np.arange(9.0).reshape(3, 3))
    data = PL.broadcast_data(data0, src=0, device='cuda:0')
    assert np.array_equal(data['S'], np.asarray(data0['S']).astype(np.uint8)) and data['N'] == data0['N']
    N = data['N']
    popn = Population(make_model('standard_glm', N=N, dt=0.001), device=0)
    popn.add_data(data)
    x0 = popn.sample(np.random.RandomState(11))
    lp_n, _ = parallel_compute_log_p(popn, x0)                                   # all-gather of the per-neuron terms
    xn = parallel_coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch', shard='neurons')
    xt = parallel_coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch', shard='time')  # all-reduce per evaluation
    m2 = make_model('sparse_weighted_model', N=N, dt=0.001)
    stabilize_sparsity(m2)
    pop2 = Population(m2, device=0)
    pop2.add_data(data)
    y0 = pop2.sample(np.random.RandomState(12))
    y0['net']['weights']['W'] = 0.2 * np.asarray(y0['net']['weights']['W'])
    y = parallel_gibbs_sample(pop2, N_samples=1, x0=copy.deepcopy(y0), seed=5, verbose=False)[-1]
    lpy = pop2.compute_log_p(y)
    # the same work without collectives
    os.environ['PYGLM_COLLECTIVES_AT_WORLD1'] = '0'
    assert PL.alone()
    lp_ref = popn.compute_log_p(x0)
    x_ref = coord_descent(popn, x0=copy.deepcopy(x0), maxiter=1, batched='torch')
    out.put((lp_n, lp_ref, _pack_state(popn, xn)[0], _pack_state(popn, xt)[0], _pack_state(popn, x_ref)[0],
             popn.compute_log_p(xt), popn.compute_log_p(x_ref), lpy))
    dist.destroy_process_group()


def test_rccl_collectives_execute_on_a_one_rank_group():
    """The nccl (= RCCL) branches of theano_pyglm_amd.parallel and of the sharded MAP / Gibbs drivers -- device
    all_reduce, all_gather_into_tensor, broadcast -- run on the GPU box, on a one-rank group (the box has one GPU),
    and leave the single-process numbers unchanged."""
    import torch.multiprocessing as mp
    from theano_pyglm_amd.harness.generate_synth_data import make_dataset
    N = 6
    model, popn, data = make_dataset('standard_glm', N, 8.0, seed=21)
    clean = dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed'))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_worker_rccl, args=(_free_port(), clean, q))
    pr.start()
    lp_n, lp_ref, rows_n, rows_t, rows_ref, lpt, lpr, lpy = _collect([pr], q, 1)[0]
    assert np.isclose(lp_n, lp_ref, rtol=1e-12)
    assert np.allclose(rows_n, rows_ref, rtol=1e-9, atol=1e-12)
    assert np.allclose(rows_t, rows_ref, rtol=1e-9, atol=1e-12)
    assert np.isclose(lpt, lpr, rtol=1e-10) and np.isfinite(lpy)


def test_bench_multi_rank_path_on_rccl_one_rank():
    """bench.py's N > 1 code path (process group on nccl bound to the rank's GPU, barrier, all-reduce of the packed
    (ll, grad) block resp. all-gather of the ll shards on the bench stream, MAX over ranks, per-rank gather) executes
    on RCCL with one rank; the JSON record is the last line of stdout and carries the per-rank block."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for shard in ('time', 'neurons'):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
        r = subprocess.run([sys.executable, 'bench.py', '--rccl-selftest', '--shard', shard, '--steps', '4', '--warmup', '1',
                            '--seconds', '60', '--neurons', '64'], cwd=root, env=env, capture_output=True, text=True,
                           timeout=400)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec['n_gpus'] == 1 and rec['steps'] == 4 and rec['value'] > 0
        assert rec['per_rank'] and rec['per_rank'][0]['collective_ms'] is not None
        assert 'roofline' in rec


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset) starts both ranks itself (here on the one GPU of the
    box, gloo collectives): the record says n_gpus = 2, carries both shardings -- the time-sharded headline with its
    all-reduce and the neuron-sharded step with its all-gather (north_star's split) -- and is the last line of stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict((k, v) for k, v in os.environ.items()
               if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--debug-single-device', '--steps', '3', '--warmup', '1',
                        '--seconds', '60', '--neurons', '64'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['n_gpus'] == 2 and rec['ranks'] == 2 and rec['steps'] == 3 and rec['value'] > 0
    assert rec['rccl_ranks'] == 0 and rec['collective_backend'] == 'gloo'       # a gloo run exercises no RCCL rank
    assert len(rec['per_rank']) == 2 and sorted(p['rank'] for p in rec['per_rank']) == [0, 1]
    sn = rec['sharding_neurons']
    assert sn['ms_per_step'] > 0 and len(sn['per_rank']) == 2
    assert sorted(p['neurons'] for p in sn['per_rank']) == [32, 32]
    assert all(p['bins'] == 60000 for p in sn['per_rank'])
    # a rank that cannot get its GPU makes the whole run fail (exit code 3), before any rendezvous
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--seconds', '60',
                        '--neurons', '64'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 3, (r.returncode, r.stderr[-2000:])


def test_bench_eight_ranks_uneven_neuron_shards_on_one_gpu():
    """`python bench.py --gpus 8 --neurons 100` (eight ranks on the one GPU of the box, gloo): 100 neurons over 8 ranks
    are shards of 12 or 13 neurons -- the neuron-sharded step gathers them through the padded all-gather of
    parallel.allgather_rows_t -- and both sharded population lls equal the single-rank evaluation (asserted inside
    bench.py at rtol 1e-10); then the same with --shard neurons as the headline sharding."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict((k, v) for k, v in os.environ.items()
               if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    for extra in ([], ['--shard', 'neurons']):
        r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--debug-single-device', '--steps', '2', '--warmup', '1',
                            '--seconds', '30', '--neurons', '100'] + extra, cwd=root, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec['n_gpus'] == 8 and rec['ranks'] == 8 and rec['rccl_ranks'] == 0 and rec['value'] > 0
        assert len(rec['per_rank']) == 8
        if extra:
            assert sorted(p['neurons'] for p in rec['per_rank']) == [12] * 4 + [13] * 4
        else:
            sn = rec['sharding_neurons']
            assert sorted(p['neurons'] for p in sn['per_rank']) == [12] * 4 + [13] * 4
            assert all(p['bins'] == 30000 for p in sn['per_rank'])
