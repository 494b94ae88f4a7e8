"""C5 stress variant: spatiotemporal_glm N=64, T=300 s, D_stim=1024 (32x32 pixels),
identity spatial basis (Bx=1024), Bt=3 -> 3072 dense stimulus columns + 192 impulse columns.
Device feature build time and ll+grad time through the sliced path.  Dev tool."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd import _lib
from tests import helpers as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dt, dt_stim = 0.001, 0.1
nT = int(round(T / dt))
rng = np.random.default_rng(1234 + 5)
ib = H.st_ibasis()                     # impulse basis B=3, R=300
g = H.golden()
ibt = g['lr2d_ibasis_t']               # temporal stimulus basis (300,3), norm
S = np.minimum(rng.poisson(20.0 * dt, size=(nT, N)), 10).astype(np.uint8)
stim = rng.standard_normal((int(round(T / dt_stim)), D))
dev = _lib.DeviceGlm(N, nT, 3, 300, 'exp', dt)
dev.set_spikes(S)
dev.set_basis(ib)
t0 = time.time()
dev.set_stimulus(stim, dt_stim, ibt, None, layout=0)
t_build = time.time() - t0
P = 1 + 3 * D + N * 3
theta = np.zeros((N, P))
theta[:, 0] = 1.0 + 0.3 * rng.standard_normal(N)
theta[:, 1:1 + 3 * D] = 0.002 * rng.standard_normal((N, 3 * D))
theta[:, 1 + 3 * D:] = 0.02 * rng.standard_normal((N, N * 3))
Weff = np.ones((N, N))
for i in range(3):
    t0 = time.time()
    ll, gr = dev.ll_grad(theta, Weff)
    wall = time.time() - t0
fused, total = dev.last_timing()
flops = 4.0 * nT * (N * 3 + 3 * D) * N
print("C5 stress N=%d T=%gs D_stim=%d: feature build %.2f s (%.1f GB on device); ll+grad %.1f ms "
      "(wall %.1f ms) = %.1f TFLOP/s; finite=%s"
      % (N, T, D, t_build, nT * 3 * D * 8 / 1e9, total, wall * 1e3, flops / total / 1e9,
         bool(np.all(np.isfinite(ll)) and np.all(np.isfinite(gr)))))

# the same model with the stimulus kept separable on the device (pgl_set_stimulus_separable): theta rows
# [bias, w_t(3), w_x(D), w_imp]; the rank-1 weights above are not rank-1, so time a rank-1 draw on both paths
w_t, w_x = 0.3 * rng.standard_normal((N, 3)), 0.05 * rng.standard_normal((N, D))
theta[:, 1:1 + 3 * D] = np.einsum('nt,nx->ntx', w_t, w_x).reshape(N, -1)
ll_d, g_d = dev.ll_grad(theta, Weff)
dense_ms = dev.last_timing()[1]
dev.close()
sep = _lib.DeviceGlm(N, nT, 3, 300, 'exp', dt)
sep.set_spikes(S)
sep.set_basis(ib)
t0 = time.time()
sep.set_stimulus_separable(stim, dt_stim, ibt, None)
t_build_s = time.time() - t0
th_s = np.concatenate((theta[:, :1], w_t, w_x, theta[:, 1 + 3 * D:]), axis=1)
for i in range(3):
    t0 = time.time()
    ll_s, g_s = sep.ll_grad(th_s, Weff)
    wall = time.time() - t0
fused, total = sep.last_timing()
G = g_d[:, 1:1 + 3 * D].reshape(N, 3, D)
g_chain = np.concatenate((g_d[:, :1], np.einsum('ntx,nx->nt', G, w_x), np.einsum('ntx,nt->nx', G, w_t), g_d[:, 1 + 3 * D:]), axis=1)
print("separable path: setup %.3f s (%.0f MB on device); ll+grad %.2f ms (wall %.1f ms) vs dense %.1f ms = %.1fx; "
      "max rel ll diff %.1e, rel grad diff %.1e"
      % (t_build_s, 2 * stim.size * 8 / 1e6, total, wall * 1e3, dense_ms, dense_ms / total,
         np.max(np.abs(ll_s - ll_d) / np.abs(ll_d)), np.max(np.abs(g_s - g_chain)) / np.max(np.abs(g_chain))))
