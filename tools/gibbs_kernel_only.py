"""A few launches of the batched Gibbs inner-ll kernel at the C4 shape (for rocprofv3 --pmc; dev tool)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
N, nT = 128, 600000
p = H.Problem(N, nT, H.std_ibasis(), seed=1238, w_scale=0.2, weighted=True)
theta = p.theta.copy()
theta[:, 1:] = np.abs(theta[:, 1:])
dev = p.device()
dev.gibbs_prepare_all(theta, p.Weff)
cols = np.arange(N); pre = (cols * 37 + 11) % N
ws = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
aw = p.Weff[pre, cols]
for _ in range(6):
    ll = dev.gibbs_ll_cols(cols, pre, aw, ws)
print(np.isfinite(ll).mean())
