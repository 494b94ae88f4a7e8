"""Random shapes for the separable stimulus at the frame rate (the body of tests/test_gpu_population.py::
test_separable_stimulus_frame_rate_randomised_shapes with a free seed): default path (stimulus current inside the forward
contraction where it applies) against the slab form (option 94 = 3) and the tap-rate kernels (94 = 2).  Dev tool:
python3 tools/fuzz_sepf.py [seed] [cases]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
done = fused = 0
worst = 0.0
while done < ncase:
    Rt = int(rng.choice([20, 47, 100, 233, 300]))
    q = int(rng.randint(max(2, -(-Rt // 6)), 151))
    if -(-Rt // q) + 2 > 8:
        continue
    Bt = int(rng.randint(1, 5))
    N = int(rng.choice([1, 5, 16, 17, 33, 48, 64, 80, 128]))
    nT = int(rng.randint(20, 400)) * 16 + int(rng.randint(0, 16))
    D = int(rng.randint(2, 24))
    Bx = D if rng.rand() < 0.5 else int(rng.randint(1, 12))
    Tstim = max(2, int(nT / q * rng.choice([0.3, 0.5, 1.0, 1.3])) + int(rng.randint(0, 3)))
    ibt = rng.randn(Rt, Bt) / np.sqrt(Rt)
    ibx = None if Bx == D else rng.randn(D, Bx)
    stim = rng.randn(Tstim, D)
    p = H.Problem(N, nT, H.st_ibasis(), kind='exp' if rng.rand() < 0.5 else 'explinear', seed=100 + done, w_scale=0.02, bias_mu=1.0)
    dev = p.device()
    dev.set_stimulus_separable(stim, q * 0.001, ibt, ibx)
    assert dev.info()['stim_path'] == 2
    th = np.concatenate((p.theta[:, :1], 0.3 * rng.randn(N, Bt), 0.3 * rng.randn(N, Bx) / np.sqrt(Bx), p.theta[:, 1:]), axis=1)
    n_lo = int(rng.randint(0, N)); n_hi = int(rng.randint(n_lo + 1, N + 1))
    t_lo = 16 * int(rng.randint(0, nT // 32)); t_hi = int(rng.randint(t_lo + 1, nT + 1))
    dev.set_time_range(t_lo, t_hi)
    ll_f, g_f = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
    dev.set_option(94, 3); ll_b, g_b = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
    dev.set_option(94, 2); ll_t, g_t = dev.ll_grad(th[n_lo:n_hi], p.Weff, n_lo, n_hi)
    case = (Rt, q, Bt, N, nT, D, Bx, Tstim, n_lo, n_hi, t_lo, t_hi)
    is_f = not np.array_equal(g_f, g_b)
    fused += int(is_f)
    e1, e2 = H.rel_err(g_f, g_b), H.rel_err(g_f, g_t)
    worst = max(worst, e1, e2)
    ok = np.allclose(ll_f, ll_b, rtol=1e-11, atol=1e-12) and np.allclose(ll_f, ll_t, rtol=1e-11, atol=1e-12) and e1 < 1e-10 and e2 < 1e-10
    if not ok:
        print("MISMATCH", case, "fused" if is_f else "slab", e1, e2)
    dev.close()
    done += 1
print("seed done: %d cases, %d on the fused forward, worst relative gradient difference %.1e" % (done, fused, worst))
