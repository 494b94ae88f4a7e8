"""Fused ll+grad kernels against the oracle on wild parameters (dev tool): neurons with biases in [-800, 800], at the
underflow / overflow edges, weights scaled up to 200x; finite pattern per neuron and values.  python tools/fuzz_oracle.py [seed]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nb = 0
for trial in range(30):
    N = int(rng.choice([5, 17, 33]))
    kind = str(rng.choice(['explinear', 'exp']))
    nT = int(rng.choice([300, 900]))
    p = H.Problem(N, nT, H.std_ibasis(), kind=kind, seed=int(rng.randint(1 << 30)), weighted=bool(rng.rand() < 0.5),
                  rate_hz=float(rng.choice([5.0, 20.0, 60.0])))
    th = p.theta.copy()
    for n in rng.choice(N, size=int(rng.randint(1, 4)), replace=False):
        mode = rng.randint(3)
        if mode == 0: th[n, 0] = rng.uniform(-800, 800)
        elif mode == 1: th[n, 1:] *= rng.choice([20.0, 200.0])
        else: th[n, 0] = rng.choice([-730.0, -744.0, -745.0, -746.0, -709.0, 709.0, 745.0])
    if kind == 'exp':
        th[:, 0] = np.clip(th[:, 0], -800, 700)
    p.theta = th
    with np.errstate(all='ignore'):
        ll0, g0 = p.oracle_ll_grad()
    for kern in (0, 2, 4):
        d = p.device(); d.set_option(_lib.OPT_KERNEL, kern)
        ll, g = d.ll_grad(th, p.Weff); ver = d.info()['kernel_version']; d.close()
        f0, f1 = np.isfinite(ll0), np.isfinite(ll)
        gf0, gf1 = np.isfinite(g0).all(1), np.isfinite(g).all(1)
        both = f0 & f1; gb = gf0 & gf1
        okll = np.allclose(ll[both], ll0[both], rtol=1e-9, atol=0)
        okg = (not gb.any()) or np.allclose(g[gb], g0[gb], rtol=1e-8, atol=1e-9 * np.abs(g0[gb]).max())
        if not (okll and okg) or (f0 != f1).any() or (gf0 != gf1).any():
            nb += 1
            bad = np.where((f0 != f1) | (gf0 != gf1))[0]
            print("trial %d N=%d %s kernel %d (v%d): ll ok %s grad ok %s pattern diff rows %s bias %s oracle ll %s dev ll %s" %
                  (trial, N, kind, kern, ver, okll, okg, bad.tolist(), np.round(th[bad, 0], 2).tolist(), ll0[bad][:3], ll[bad][:3]))
print("oracle fuzz done: %d discrepancies" % nb)
