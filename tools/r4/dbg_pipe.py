import sys, numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
p = H.Problem(128, 1536, H.std_ibasis(), seed=4, w_scale=0.5)
ll0, g0 = p.oracle_ll_grad()
for nch in (0, 1, 3):
    d = p.device(nchunks=nch)
    ll, g = d.ll_grad(p.theta, p.Weff)
    print("nchunks", nch, d.info()['kernel_version'], "ll relerr", np.max(np.abs(ll - ll0) / np.abs(ll0)),
          "grad relerr bias", H.rel_err(g[:, 0], g0[:, 0]), "L cols", H.rel_err(g[:, 1:1 + 256], g0[:, 1:1 + 256]),
          "H cols", H.rel_err(g[:, 257:], g0[:, 257:]))
    ll2, _ = d.ll_grad(p.theta, p.Weff, want_grad=False)
    print("   ll only relerr", np.max(np.abs(ll2 - ll0) / np.abs(ll0)))
    # per column-tile error of the gradient
    e = np.abs(g - g0).max(axis=0)[1:].reshape(40, 16).max(axis=1) / np.abs(g0).max()
    print("   per k-tile:", np.array2string(e, precision=1))
    d.close()
