"""Does a hipGraph of one evaluation (prep_w, pass 1, pass 2, finalize) shorten the launch gaps of a short step?
1/8 time shard of C3 and the C1 / C2 shapes: stream launches against graph replays.  Dev tool."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from theano_pyglm_amd import _lib, parallel as PL
import bench

def run(N, T, G):
    dt = 0.001
    S = bench.make_workload(N, T, dt, seed=1234 + 3)
    nT = S.shape[0]
    ib = bench.standard_ibasis()
    R, B = ib.shape
    P = 1 + N * B
    rng = np.random.default_rng(99)
    theta = np.zeros((N, P)); theta[:, 0] = 20.0 + 0.1 * rng.standard_normal(N)
    theta[:, 1:] = 0.5 * rng.standard_normal((N, N * B))
    dev = _lib.DeviceGlm(N, nT, B, R, 'explinear', dt)
    dev.set_spikes(S); dev.set_basis(ib)
    dev.set_option(_lib.OPT_TIMING, 0)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    dev.set_stream(st.cuda_stream)
    d_theta = torch.from_numpy(theta).cuda(); d_W = torch.ones((N, N), dtype=torch.float64, device='cuda')
    d_out = torch.zeros(N * (1 + P), dtype=torch.float64, device='cuda')
    d_ll = d_out[:N]; d_g = d_out[N:].view(N, P)
    t_lo, t_hi = PL.time_shard_bounds(nT, 0, G)
    dev.set_time_range(t_lo, t_hi)
    ev = lambda: dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    for _ in range(8): ev()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): ev()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    ref = d_out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        ev()
    for _ in range(8): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): g.replay()
    torch.cuda.synchronize()
    wall_g = (time.perf_counter() - t0) / K * 1e3
    same = bool(torch.equal(ref, d_out))
    print("N=%d T=%g shard 1/%d: stream launches %.4f ms/step, graph replays %.4f ms/step, same result %s" % (N, T, G, wall, wall_g, same))
    dev.close()

run(128, 600.0, 8)
run(128, 600.0, 1)
run(32, 300.0, 1)
run(4, 60.0, 1)
