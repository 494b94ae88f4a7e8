"""Per-rank step time of a 1/G time shard on ONE GPU (no collective): what the 8-GPU run pays per
step besides the all-reduce.  Dev tool.   python tools/shard_step_bench.py [G ...]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from theano_pyglm_amd import _lib, parallel as PL
import bench

import os
Gs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
N, dt = 128, 0.001
S = bench.make_workload(N, 600.0, dt, seed=1234 + 3)
nT = S.shape[0]
ib = bench.standard_ibasis()
R, B = ib.shape
P = 1 + N * B
rng = np.random.default_rng(99)
theta = np.zeros((N, P)); theta[:, 0] = 20.0 + 0.1 * rng.standard_normal(N)
theta[:, 1:] = 0.5 * rng.standard_normal((N, N * B))
dev = _lib.DeviceGlm(N, nT, B, R, 'explinear', dt)
dev.set_spikes(S); dev.set_basis(ib)
TIMING = int(os.environ.get('TIMING', '1'))
dev.set_option(_lib.OPT_TIMING, TIMING)
if os.environ.get('FINW'): dev.set_option(97, int(os.environ['FINW']))      # finalize waves per fragment (A/B)
_st = torch.cuda.Stream(); torch.cuda.set_stream(_st)      # handle 0 (default stream) cannot be named
dev.set_stream(_st.cuda_stream)
d_theta = torch.from_numpy(theta).cuda(); d_W = torch.ones((N, N), dtype=torch.float64, device='cuda')
d_out = torch.zeros(N * (1 + P), dtype=torch.float64, device='cuda')
d_ll = d_out[:N]; d_g = d_out[N:].view(N, P)
wall1 = None
for G in Gs:
    t_lo, t_hi = PL.time_shard_bounds(nT, 0, G)
    dev.set_time_range(t_lo, t_hi)
    for _ in range(8):
        dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize()
    if TIMING: dev.timing_summary(reset=True)
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    n, fused, total = dev.timing_summary(reset=True) if TIMING else (0, float('nan'), float('nan'))
    wall1 = wall if (wall1 is None and G == 1) else wall1
    print("G=%d bins %d: step %.3f ms (fused kernels %.3f ms, prep+fused+finalize %.3f ms)%s"
          % (G, t_hi - t_lo, wall, fused, total,
             "" if wall1 is None else " -> %.2fx the whole recording's step, before any all-reduce" % (wall1 / wall)))
