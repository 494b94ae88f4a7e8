"""Populations beyond the named configurations (N = 256, 512): cost of one ll+grad on the sliced path (dev tool)."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
for N, nT in ((128, 300000), (192, 300000), (256, 300000), (512, 150000)):
    p = H.Problem(N, nT, H.std_ibasis(), kind='explinear', seed=N, w_scale=0.3 * np.sqrt(128.0 / N))
    dev = p.device()
    ll, g = dev.ll_grad(p.theta, p.Weff)
    t0 = time.perf_counter()
    for _ in range(3):
        ll, g = dev.ll_grad(p.theta, p.Weff)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    info = dev.info()
    print("N = %3d nT = %d: %.2f ms per ll+grad (host-pointer API), %.1f TFLOP/s (alg.) = %.2f of peak, kernel version %s, finite %s"
          % (N, nT, ms, info['flops'] / ms / 1e9, info['flops'] / ms / 1e9 / 78.6, info['kernel_version'], bool(np.all(np.isfinite(ll)))))
    dev.close()
