import sys, time
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
ws11 = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
aw = p.Weff[pre, cols]
for dbg in (0, 1, 2, 3, 4, 8, 15):
    dev.set_option(99, dbg)
    for K in (1, 11):
        ws = ws11[:, :K].copy()
        for _ in range(3): dev.gibbs_ll_cols(cols, pre, aw, ws)
        t0 = time.perf_counter()
        for _ in range(10): dev.gibbs_ll_cols(cols, pre, aw, ws)
        print("dbg %2d K %2d: %.3f ms per call" % (dbg, K, (time.perf_counter() - t0) * 100))
