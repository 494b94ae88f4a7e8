"""Regime-split batched Gibbs kernels against the all-f64 kernel on wild parameters (dev tool): biases in [-800, 800],
candidate weights up to 1e3, N = 1..40, recordings shorter and longer than a block.  python tools/fuzz_gibbs.py [seed]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nb = 0; ncmp = 0; nnf = 0
for trial in range(60):
    N = int(rng.choice([1, 3, 9, 17, 40]))
    nT = int(rng.choice([37, 300, 1000, 4099]))
    p = H.Problem(N, nT, H.std_ibasis(), kind='explinear', seed=int(rng.randint(1 << 30)), weighted=True,
                  w_scale=float(rng.choice([0.1, 1.0, 10.0])), rate_hz=float(rng.choice([5.0, 30.0, 100.0])),
                  bias_mu=float(rng.uniform(-30, 30)))
    th = p.theta.copy()
    for n in rng.choice(N, size=min(N, int(rng.randint(0, 3))), replace=False):
        th[n, 0] = rng.choice([-800.0, -740.0, -720.0, 0.0, 700.0, 800.0])
    d = p.device()
    d.gibbs_prepare_all(th, p.Weff)
    for _ in range(3):
        K = int(rng.randint(1, 17)); nc = int(rng.randint(1, 2 * N + 2))
        cols = rng.randint(0, N, size=nc); pre = rng.randint(0, N, size=nc)
        if rng.rand() < 0.4:
            pre[:] = pre[0]          # a sweep step: one presynaptic neuron for all columns (filtered-spike-train path)
        ws = rng.standard_normal((nc, K)) * rng.choice([0.1, 1.0, 30.0, 1e3])
        aw = p.Weff[pre, cols]
        d.set_option(_lib.OPT_GIBBS_KERNEL, 1); old = d.gibbs_ll_cols(cols, pre, aw, ws)
        d.set_option(_lib.OPT_GIBBS_KERNEL, 0); new = d.gibbs_ll_cols(cols, pre, aw, ws)
        f0, f1 = np.isfinite(old), np.isfinite(new)
        both = f0 & f1
        ok = np.allclose(new[both], old[both], rtol=1e-10, atol=0)
        ncmp += old.size; nnf += int((~f1).sum())
        # old NaN where new finite is the known last-binade case; new non-finite where old finite must not happen
        if not ok or (f0 & ~f1).any():
            nb += 1
            print("trial", trial, "N", N, "nT", nT, "K", K, "ok", ok, "new nonfinite where old finite:", int((f0 & ~f1).sum()), "old nonfinite where new finite:", int((~f0 & f1).sum()))
            if not ok:
                r = np.abs(new[both] - old[both]) / np.abs(old[both]); print("   max rel", r.max(), "max abs", np.abs(new[both] - old[both]).max(), "|old| at worst", np.abs(old[both])[r.argmax()])
            if (f0 & ~f1).any():
                ii = np.argwhere(f0 & ~f1)[0]
                c, k = int(ii[0]), int(ii[1])
                from oracle import glm_oracle as O
                A = (p.Weff != 0).astype(float)
                w = th[cols[c], 1:].reshape(N, p.B)
                I_imp = O.impulse_currents(p.fS, w)
                I_other = O.other_current(I_imp, A, p.Weff, pre[c], cols[c])
                x = th[cols[c], 0] + I_other + ws[c, k] * I_imp[:, pre[c]]
                with np.errstate(all='ignore'):
                    ref = O.mcmc_inner_ll(ws[c, k:k + 1], th[cols[c], 0], 0.0, I_other, I_imp[:, pre[c]], p.S[:, cols[c]].astype(float), p.dt, p.kind)
                print("   entry", c, k, "old", old[c, k], "new", new[c, k], "oracle", ref, "x min %.3f max %.3f" % (x.min(), x.max()), "x at spikes", np.round(x[p.S[:, cols[c]] > 0][:5], 2))
    d.close()
print("gibbs fuzz done: %d discrepancies, %d values, %d non-finite" % (nb, ncmp, nnf))
