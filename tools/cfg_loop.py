"""One BASELINE configuration evaluated a few times (for rocprofv3 / PMC passes; dev tool).
    python tools/cfg_loop.py C1|C2|C3|C5|C5S [reps]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, T, ib, kind, Ds, ws = {'C1': (4, 60.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C2': (32, 300.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C3': (128, 600.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C5': (64, 300.0, H.st_ibasis(), 'exp', 9, 0.02),
                          'C5S': (64, 300.0, H.st_ibasis(), 'exp', 0, 0.02)}[cfg]
p = H.Problem(N, int(round(T / 0.001)), ib, kind=kind, Dstim=Ds, seed=1234, w_scale=ws)
dev = p.device()
theta = p.theta
if cfg == 'C5S':          # stress variant: D_stim = 1024 pixels, identity spatial basis, Bt = 3, separable device path
    rng = np.random.default_rng(1234 + 5)
    stim = rng.standard_normal((int(round(T / 0.1)), 1024))
    dev.set_stimulus_separable(stim, 0.1, np.ascontiguousarray(H.golden()['lr2d_ibasis_t']), None)
    theta = np.concatenate((p.theta[:, :1], 0.3 * rng.standard_normal((N, 3)), 0.05 * rng.standard_normal((N, 1024)),
                            p.theta[:, 1:]), axis=1)
if len(sys.argv) > 3:
    dev.set_option(94, int(sys.argv[3]))      # dev: 2 tap-rate kernels, 3 stimulus current through the slab, 4 residual slab + k_sepf_bwd
import time
for i in range(reps):
    ll, g = dev.ll_grad(theta, p.Weff)
print(cfg, dev.info()['kernel_version'], "fused %.3f ms total %.3f ms" % dev.last_timing())
ts = []
for i in range(20):
    ll, g = dev.ll_grad(theta, p.Weff)
    ts.append(dev.last_timing())
print("median of 20: fused %.4f ms total %.4f ms" % (np.median([t[0] for t in ts]), np.median([t[1] for t in ts])))
