"""MCMC inner-ll batches/s at the C4 shape (sparse_weighted_model, N=128, T=600 s): one batch =
the 11 ll values (10 Gauss-Hermite nodes + w=0) of one (n_pre, n_post) pair. Dev tool."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
from oracle import glm_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
nT = int(round(T / 0.001))
p = H.Problem(N, nT, H.std_ibasis(), seed=1234 + 4, weighted=True, w_scale=0.2)
# Dirichlet-like nonnegative normalised weights
w = np.abs(p.theta[:, 1:]).reshape(N, N, p.B)
w = w / w.sum(2, keepdims=True)
p.theta[:, 1:] = w.reshape(N, -1)
dev = p.device()
ws, _ = O.gauss_hermite_nodes(0.0, 1.0)
ws = np.concatenate((ws, [0.0]))
n_post = 5
t0 = time.time()
dev.gibbs_prepare(n_post, p.theta[n_post], p.Weff[:, n_post])
t_prep = time.time() - t0
t0 = time.time()
dev.gibbs_prepare(n_post, p.theta[n_post], p.Weff[:, n_post])
t_prep2 = time.time() - t0
for n_pre in range(8):
    dev.gibbs_ll(n_pre, p.Weff[n_pre, n_post], ws)
t0 = time.time()
K = 256
for i in range(K):
    n_pre = i % N
    ll = dev.gibbs_ll(n_pre, p.Weff[n_pre, n_post], ws)
dt_ = (time.time() - t0) / K
bytes_alg = 3 * nT * 8
print("prepare (I_imp of all %d presyn + I_net): %.2f ms (first %.2f ms)" % (N, t_prep2 * 1e3, t_prep * 1e3))
print("inner-ll batch (11 ll values, nT=%d): %.1f us -> %.0f batches/s; algorithmic %.1f MB/batch -> %.2f TB/s"
      % (nT, dt_ * 1e6, 1.0 / dt_, bytes_alg / 1e6, bytes_alg / dt_ / 1e12))
print("sweep estimate (N^2 pairs + N prepares): %.2f s" % (N * N * dt_ + N * t_prep2))
