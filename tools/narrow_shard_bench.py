"""Narrow post blocks of a wide population (16 / 32 of 128 neurons, K = 640: what a neuron-sharded rank or a late
line-search launch of the lock-step MAP evaluates): k_fused6 with one image buffer against k_fused2.  Dev tool."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import helpers as H
N, nT = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 600000
p = H.Problem(N, nT, H.std_ibasis(), kind='explinear', seed=1234, w_scale=0.5)
dev = p.device()
W = np.ascontiguousarray(p.Weff)
for width in (16, 32):
    res = {}
    for opt in (2, 0):
        dev.set_option(95, opt)
        ll, g = dev.ll_grad(p.theta[:width], W, 0, width)
        for _ in range(3):
            dev.ll_grad(p.theta[:width], W, 0, width)
        info = dev.info(0, width)
        d_theta = torch.from_numpy(p.theta[:width].copy()).cuda(); d_W = torch.from_numpy(W).cuda()
        d_ll = torch.zeros(width, dtype=torch.float64, device='cuda'); d_g = torch.zeros((width, p.P), dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr(), 0, width)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        res[opt] = (ll, g)
        print("%d neurons, option 95 = %d: kernel %d, %.3f ms per evaluation (%.1f TFLOP/s alg.)" % (width, opt, info['kernel_version'], ms, info['flops'] / ms / 1e9))
    print("   max rel diff ll %.2e grad %.2e" % (np.max(np.abs(res[0][0] - res[2][0]) / np.abs(res[2][0])), np.max(np.abs(res[0][1] - res[2][1])) / np.max(np.abs(res[2][1]))))
