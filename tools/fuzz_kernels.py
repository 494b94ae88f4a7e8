"""Kernel families against each other on wild parameters (dev tool; a seeded short run is in tests/test_gpu_kernels.py):
random population sizes, nonlinearities, stimulus columns, and a few neurons per trial with extreme biases / weights
(rates at the underflow and overflow edges).  python tools/fuzz_kernels.py [seed]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nbad = 0
ncmp = {}
nnf = 0
for trial in range(40):
    N = int(rng.choice([3, 9, 17, 30, 33, 48, 64, 80, 100, 128, 130, 160, 200, 257, 300]))   # (> 128: column slices on resident tiles)
    kind = str(rng.choice(['explinear', 'exp']))
    nT = int(rng.choice([700, 2500, 6000]))
    Dstim = int(rng.choice([0, 0, 3]))
    p = H.Problem(N, nT, H.std_ibasis() if rng.rand() < 0.7 else H.st_ibasis(), kind=kind, seed=int(rng.randint(1 << 30)),
                  weighted=bool(rng.rand() < 0.5), Dstim=Dstim, rate_hz=float(rng.choice([5.0, 20.0, 60.0])))
    th = p.theta.copy()
    # a few wild neurons
    for n in rng.choice(N, size=min(N, int(rng.randint(0, 5))), replace=False):
        mode = rng.randint(4)
        if mode == 0: th[n, 0] = rng.uniform(-800, 800)
        elif mode == 1: th[n, 1:] *= rng.choice([20.0, 200.0, 2000.0])
        elif mode == 2: th[n, 0] = rng.choice([-730.0, -745.0, -746.0, -709.0, 709.0, 745.0])
        else: th[n, 1 + Dstim:1 + Dstim + p.B] = rng.choice([1e3, -1e3, 1e5])
    if kind == 'exp':
        th[:, 0] = np.clip(th[:, 0], -800, 700)
    res = {}
    for kern in (2, 0, 3, 4, 6, 7):
        d = p.device()
        d.set_option(_lib.OPT_KERNEL, kern)
        try:
            ll, g = d.ll_grad(th, p.Weff)
            res[kern] = (ll, g, d.info()['kernel_version'])
        except Exception as e:
            res[kern] = None
        d.close()
    ll0, g0, _ = res[2]
    for kern in (0, 3, 4, 6, 7):
        if res[kern] is None: continue
        ll, g, ver = res[kern]
        if ver == 2 and kern != 0: continue
        ncmp[ver] = ncmp.get(ver, 0) + 1
        nnf += int((~np.isfinite(g)).any(1).sum())
        f0, f1 = np.isfinite(ll0), np.isfinite(ll)
        gf0, gf1 = np.isfinite(g0).all(1), np.isfinite(g).all(1)
        both = f0 & f1
        okll = np.allclose(ll[both], ll0[both], rtol=1e-9, atol=0)
        gb = gf0 & gf1
        scale = np.abs(g0[gb]).max() if gb.any() else 1.0
        okg = np.allclose(g[gb], g0[gb], rtol=1e-8, atol=1e-9 * scale)
        # finite patterns may differ only in the last binade above the underflow / near overflow: report
        pat = (f0 != f1).sum() + (gf0 != gf1).sum()
        if not (okll and okg) or pat:
            nbad += 1
            print("trial %d N=%d %s nT=%d D=%d kernel %d (v%d): ll ok %s grad ok %s, finite-pattern diffs ll %s grad %s" %
                  (trial, N, kind, nT, Dstim, kern, ver, okll, okg, np.where(f0 != f1)[0].tolist(), np.where(gf0 != gf1)[0].tolist()))
            if not okg:
                r = np.abs(g[gb] - g0[gb]).max(1) / scale
                bad = np.where(gb)[0][r > 1e-8]
                print("    worst grad rows", bad[:8].tolist(), "rel", np.round(r[r > 1e-8][:8], 12).tolist(), "theta0", np.round(th[bad[:8], 0], 1).tolist())
    # a narrow shard (one post tile against a row of 25 .. 40 k-tiles: k_fused8, block-form images through per-wave rings)
    if 78 <= N <= 128 and Dstim == 0:
        cnt = int(rng.randint(1, 17))
        lo = int(rng.randint(0, N - cnt + 1))
        d = p.device()
        names = _lib.plan_kernels(N, B=p.B, R=p.ibasis.shape[0], nT=nT, n_lo=lo, count=cnt)
        ll, g = d.ll_grad(th[lo:lo + cnt], p.Weff, lo, lo + cnt)
        d.close()
        l0, gg0 = ll0[lo:lo + cnt], g0[lo:lo + cnt]
        ncmp[names[0][:8]] = ncmp.get(names[0][:8], 0) + 1
        f0, f1 = np.isfinite(l0), np.isfinite(ll)
        gf0, gf1 = np.isfinite(gg0).all(1), np.isfinite(g).all(1)
        both, gb = f0 & f1, gf0 & gf1
        scale = np.abs(gg0[gb]).max() if gb.any() else 1.0
        ok = np.allclose(ll[both], l0[both], rtol=1e-9, atol=0) and np.allclose(g[gb], gg0[gb], rtol=1e-8, atol=1e-9 * scale)
        pat = (f0 != f1).sum() + (gf0 != gf1).sum()
        if not ok or pat:
            nbad += 1
            print("trial %d N=%d %s nT=%d shard [%d, %d) %s: values ok %s, finite-pattern diffs %d"
                  % (trial, N, kind, nT, lo, lo + cnt, names, ok, pat))
print("fuzz done, %d discrepancies; comparisons per kernel version %s; non-finite gradient rows seen %d" % (nbad, ncmp, nnf))
