"""Time one rank's shard of the C3 workload for G = 1,2,4,8 (dev tool; single GPU)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
N, nT = 128, 600000
p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5)
dev = p.device(f32=(len(sys.argv) > 1 and sys.argv[1] == 'f32'))
for G in (1, 2, 4, 8):
    lo, hi = 0, N // G
    for i in range(4):
        ll, g = dev.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
    fused, total = dev.last_timing()
    info = dev.info(lo, hi)
    print("G=%d shard %3d neurons: fused %.3f ms total %.3f ms  blocks %d x %d  -> %.1f TF/s; ideal-scaling speedup %.2fx"
          % (G, hi - lo, fused, total, info['blocks'], info['threads'], info['flops'] / fused / 1e9, 0))
