"""Fused-kernel time of a small population against recording length and chunk count (dev tool)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for T in (300.0, 1200.0):
    nT = int(T * 1000)
    p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5)
    for kern in (7, 6):
        for nch in (0, 256, 512, 1024):
            dev = p.device(nchunks=nch)
            dev.set_option(_lib.OPT_KERNEL, kern)
            for i in range(5):
                dev.ll_grad(p.theta, p.Weff)
            fused, total = dev.last_timing()
            info = dev.info()
            print("N=%d T=%g kernel %d (v%d) nchunks %d (blocks %d): fused %.3f ms total %.3f ms -> %.1f TFLOP/s"
                  % (N, T, kern, info['kernel_version'], nch, info['blocks'], fused, total, info['flops'] / fused / 1e9))
            dev.close()
