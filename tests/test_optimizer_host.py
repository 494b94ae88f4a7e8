"""CPU tests of the lock-step optimizer's line search (theano_pyglm_amd/csrc/pglm_linesearch.h, compiled for the host
with gcc through tests/csrc/ls_host.c): the reverse-communication state machine takes the trial steps of scipy's own
search (DCSRCH, what the reference's fit_glm runs through scipy.optimize.minimize(method="bfgs"),
coord_descent.py:194-199), and a BFGS loop built on it the way the HIP row kernels are (k_bfgs_init / linesearch /
hmul / update) reproduces scipy's iterates."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.optimize as opt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FTOL, GTOL, XTOL, STPMIN, STPMAX = 1e-4, 0.9, 1e-14, 1e-100, 1e100


@pytest.fixture(scope='module')
def ls(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('ls') / 'ls_host.so')
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'csrc', 'ls_host.c')])
    lib = C.CDLL(so)
    lib.ls_step.restype = C.c_int
    lib.ls_step.argtypes = [C.c_void_p] + [C.c_double] * 7
    lib.ls_start.argtypes = [C.c_void_p] + [C.c_double] * 6
    lib.ls_first_step.restype = C.c_double
    lib.ls_first_step.argtypes = [C.c_double] * 3
    assert lib.ls_ndoubles() == 18                    # PGL_LS_NDOUBLES: the (18, M) block of the device state
    return lib


def _search(lib, phi, dphi, a0, max_trials=100):
    st = np.zeros(18)
    lib.ls_start(st.ctypes.data, a0, phi(0.0), dphi(0.0), FTOL, STPMIN, STPMAX)
    steps = []
    for _ in range(max_trials):
        a = float(st[0])
        steps.append(a)
        rc = lib.ls_step(st.ctypes.data, phi(a), dphi(a), FTOL, GTOL, XTOL, STPMIN, STPMAX)
        if rc != 0:
            return rc, a, steps
    return 2, a, steps


def _scipy_search(phi, dphi, a0):
    from scipy.optimize._dcsrch import DCSRCH
    steps = []

    def p(a):
        steps.append(float(a))
        return phi(a)

    stp, phi1, phi0, task = DCSRCH(p, dphi, FTOL, GTOL, XTOL, STPMIN, STPMAX)(a0, phi0=phi(0.0), derphi0=dphi(0.0),
                                                                              maxiter=100)
    return task, stp, steps


def test_line_search_takes_scipys_trial_steps(ls):
    """Random one-dimensional problems of the kinds the fits meet -- quadratics with curvatures from 1e-3 to 1e8 (prior
    precision 1e6 beside O(1) directions), exponential growth (the exp nonlinearity), a non-convex quartic, and a wall
    of fit_glm's 1e16 sentinel (NaN objective) behind a quadratic: the sequence of trial steps equals that of scipy's
    DCSRCH to 1e-13 and the two agree on convergence."""
    pytest.importorskip('scipy.optimize._dcsrch')
    rng = np.random.default_rng(1)
    conv = checked = 0
    for trial in range(1200):
        kind = trial % 4
        if kind == 0:
            c, b = 10 ** rng.uniform(-3, 8), -10 ** rng.uniform(-2, 3)
            phi, dphi = (lambda a: 0.5 * c * a * a + b * a + 3.0), (lambda a: c * a + b)
        elif kind == 1:
            k, b = 10 ** rng.uniform(-1, 2), -10 ** rng.uniform(-2, 3)
            phi, dphi = (lambda a: np.exp(min(k * a, 600)) / k + b * a), (lambda a: np.exp(min(k * a, 600)) + b)
            if dphi(0.0) >= 0:
                continue
        elif kind == 2:
            c1, c2, b = rng.uniform(0.1, 5), rng.uniform(0.1, 5), -rng.uniform(0.1, 10)
            phi, dphi = (lambda a: c1 * a ** 4 - c2 * a ** 2 + b * a), (lambda a: 4 * c1 * a ** 3 - 2 * c2 * a + b)
        else:
            wall, c, b = 10 ** rng.uniform(-6, 0), 10 ** rng.uniform(0, 6), -10 ** rng.uniform(-1, 2)
            phi = lambda a: 1e16 if a > wall else 0.5 * c * a * a + b * a
            dphi = lambda a: 0.0 if a > wall else c * a + b
        a0 = min(1.0, 10 ** rng.uniform(-4, 0.5))
        with np.errstate(all='ignore'):
            rc, a, s1 = _search(ls, phi, dphi, a0)
            task, stp, s2 = _scipy_search(phi, dphi, a0)
        assert len(s1) == len(s2) and np.allclose(s1, s2, rtol=1e-13, atol=0), (trial, kind, s1[:8], s2[:8])
        assert (rc == 1) == (task[:4] == b'CONV'), (trial, rc, task)
        if rc == 1:
            assert a == stp
            conv += 1
        checked += 1
    assert checked > 1000 and conv > 0.6 * checked


def test_first_trial_step_is_scipys(ls):
    """min(1, 1.01 * 2 (f - f_prev) / slope), 1 when that is not positive (scalar_search_wolfe1); at the start
    f_prev = f + |g| / 2 gives min(1, 1.01 / |g|)."""
    assert ls.ls_first_step(10.0, 12.0, -8.0) == min(1.0, 1.01 * 2 * (10.0 - 12.0) / -8.0)
    assert ls.ls_first_step(10.0, 100.0, -8.0) == 1.0
    assert ls.ls_first_step(10.0, 9.0, -8.0) == 1.0                   # increase before: not positive -> 1
    assert ls.ls_first_step(10.0, 10.0, -8.0) == 1.0
    g = 37.5
    assert np.isclose(ls.ls_first_step(3.0, 3.0 + g / 2, -g * g), 1.01 / g, rtol=1e-15)


def _lockstep_like_bfgs(lib, fun, x0, maxiter=225, gtol=1e-5):
    """The row kernels' algorithm for ONE row in numpy: H = I, direction -H g, the search above, the update
    H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T applied as H += U V^T with the three-column factors of
    k_bfgs_update; stops on max|g| <= gtol or maxiter iterations.  Returns (x, f, iterations, line-search steps)."""
    x = np.array(x0, float)
    f, g = fun(x)
    P = x.size
    H = np.eye(P)
    fprev = f + np.linalg.norm(g) / 2
    Hg = g.copy()
    it = nls = 0
    while np.max(np.abs(g)) > gtol and it < maxiter:
        p = -Hg
        slope = float(p @ g)
        st = np.zeros(18)
        lib.ls_start(st.ctypes.data, lib.ls_first_step(f, fprev, slope), f, slope, FTOL, STPMIN, STPMAX)
        while True:
            a = float(st[0])
            ft, gt = fun(x + a * p)
            nls += 1
            rc = lib.ls_step(st.ctypes.data, ft, float(gt @ p), FTOL, GTOL, XTOL, STPMIN, STPMAX)
            if rc != 0:
                break
        assert rc == 1, "test problems are chosen so that every search converges"
        s, y = a * p, gt - g
        x, fprev, f, g = x + s, f, ft, gt
        it += 1
        rho = 1.0 / float(s @ y)
        t = H @ g
        Hy = t - Hg
        c0 = (1.0 + rho * float(y @ Hy)) * rho
        U = np.stack((c0 * s, -rho * Hy, -rho * s), axis=1)
        V = np.stack((s, s, Hy), axis=1)
        Hg = t + U @ (V.T @ g)
        H += U @ V.T
    return x, f, it, nls


def test_bfgs_on_the_line_search_reproduces_scipy(ls):
    """Rosenbrock in 6 dimensions and a badly scaled Poisson regression with a Gaussian prior of precision 1e6 on half
    of the coordinates (the shape of spatiotemporal_glm's per-neuron problem): iteration count, number of function
    evaluations and the minimiser of scipy.optimize.minimize(method='bfgs')."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((400, 12))
    A[:, 6:] *= 20.0
    cnt = rng.poisson(np.exp(0.3 * A[:, :6].sum(1) * 0.2))
    prec = np.concatenate((np.ones(6), 1e6 * np.ones(6)))

    def glm(x):
        eta = A @ x
        lam = np.exp(eta)
        return float(np.sum(lam - cnt * eta) + 0.5 * np.sum(prec * x * x)), A.T @ (lam - cnt) + prec * x

    def rosen(x):
        return float(opt.rosen(x)), opt.rosen_der(x)

    for fun, x0 in ((rosen, np.full(6, -1.2)), (glm, 0.01 * rng.standard_normal(12))):
        res = opt.minimize(lambda v: fun(v)[0], x0, jac=lambda v: fun(v)[1], method='bfgs', options={'maxiter': 225})
        x, f, it, nls = _lockstep_like_bfgs(ls, fun, x0)
        assert it == res.nit and nls == res.nfev - 1, (it, res.nit, nls, res.nfev)
        assert abs(f - res.fun) <= 1e-9 * max(1.0, abs(res.fun))
        assert np.allclose(x, res.x, rtol=1e-6, atol=1e-8)
