import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
from oracle import glm_oracle as O
N = 11
p = H.Problem(N, 2100 + 77, H.std_ibasis(), kind='explinear', seed=93, weighted=True, w_scale=1.0, rate_hz=25.0, bias_mu=6.0)
p.theta[:4, 0] = [20.0, -3.0, 14.0, 0.5]
d = p.device()
d.gibbs_prepare_all(p.theta, p.Weff)
A = (p.Weff != 0).astype(float)
for K in (1, 11, 16):
    for cols in (np.arange(N), np.array([2]), np.array([7, 0])):
        pre = (cols * 5 + 1) % N
        base = np.concatenate(([0.0], np.geomspace(0.01, 300.0, 15)))[:K]
        ws = np.where(np.arange(K)[None, :] % 2 == 0, 1.0, -1.0) * base[None, :] * (1.0 + 0.05 * cols[:, None])
        d.set_option(_lib.OPT_GIBBS_KERNEL, 1)
        old = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
        d.set_option(_lib.OPT_GIBBS_KERNEL, 0)
        new = d.gibbs_ll_cols(cols, pre, p.Weff[pre, cols], ws)
        bad = np.isnan(new) != np.isnan(old)
        rel = np.nanmax(np.abs(new - old) / np.abs(old)) if np.isfinite(old).any() else 0
        print(K, cols, 'nan mismatch', np.argwhere(bad).tolist(), 'max rel', rel)
        for i, k in np.argwhere(bad):
            c = cols[i]
            w = p.theta[c, 1:].reshape(N, p.B)
            I_imp = O.impulse_currents(p.fS, w)
            I_other = O.other_current(I_imp, A, p.Weff, pre[i], c)
            x = p.theta[c, 0] + I_other + ws[i, k] * I_imp[:, pre[i]]
            print('   col', c, 'k', k, 'w', ws[i, k], 'old', old[i, k], 'new', new[i, k], 'x min', x.min(), 'x at spikes min', x[p.S[:, c] > 0].min(), 'n x<-745', (x < -745.13).sum(), ((x < -745.13) & (p.S[:, c] == 0)).sum())
