"""Wide populations (N > 128) on the resident-tile two-pass kernels against the in-kernel-feature K-split path and the
oracle (dev tool): ranges, a neuron list, a time range, ll only."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

for N, nT, Ds in ((130, 3000, 0), (160, 2500, 0), (200, 2000, 7), (256, 2000, 0), (300, 1500, 0)):
    p = H.Problem(N, nT, H.std_ibasis(), seed=N, Dstim=Ds, w_scale=0.5)
    d = p.device()
    info = d.info()
    ll, g = d.ll_grad(p.theta, p.Weff)
    d2 = p.device()
    d2.set_option(_lib.OPT_KERNEL, 2)
    ll2, g2 = d2.ll_grad(p.theta, p.Weff)
    llo, go = p.oracle_ll_grad(0, 3)
    e1 = np.max(np.abs(ll - ll2) / np.abs(ll2)); e2 = H.rel_err(g, g2)
    e3 = np.max(np.abs(ll[:3] - llo) / np.abs(llo)); e4 = H.rel_err(g[:3], go)
    # ll only
    ll_only, _ = d.ll_grad(p.theta, p.Weff, want_grad=False)
    # a neuron range and a time range
    lo, hi = 17, N - 9
    llr, gr = d.ll_grad(p.theta[lo:hi], p.Weff, lo, hi)
    d.set_time_range(160, nT - 37); d2.set_time_range(160, nT - 37)
    llt, gt = d.ll_grad(p.theta, p.Weff); llt2, gt2 = d2.ll_grad(p.theta, p.Weff)
    print("N=%d Dstim=%d: kernel v%d resident %.0f MB | vs K-split: ll %.1e grad %.1e | vs oracle: ll %.1e grad %.1e | ll-only equal %s | "
          "range equal %s %.1e | time range vs K-split %.1e %.1e"
          % (N, Ds, info['kernel_version'], info['resident_feature_bytes'] / 1e6, e1, e2, e3, e4, np.array_equal(ll_only, ll),
             np.allclose(llr, ll[lo:hi], rtol=1e-12), H.rel_err(gr, g[lo:hi]), np.max(np.abs(llt - llt2) / np.abs(llt2)), H.rel_err(gt, gt2)),
          flush=True)
    d.close(); d2.close()
