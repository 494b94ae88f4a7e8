"""One BASELINE configuration evaluated a few times (for rocprofv3 / PMC passes; dev tool).
    python tools/cfg_loop.py C1|C2|C3|C5 [reps]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, T, ib, kind, Ds, ws = {'C1': (4, 60.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C2': (32, 300.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C3': (128, 600.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C5': (64, 300.0, H.st_ibasis(), 'exp', 9, 0.02)}[cfg]
p = H.Problem(N, int(round(T / 0.001)), ib, kind=kind, Dstim=Ds, seed=1234, w_scale=ws)
dev = p.device()
for i in range(reps):
    ll, g = dev.ll_grad(p.theta, p.Weff)
print(cfg, dev.info()['kernel_version'], "fused %.3f ms total %.3f ms" % dev.last_timing())
