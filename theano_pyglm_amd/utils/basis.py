"""
Basis construction and the one-time host-side feature convolutions.

Counterpart of pyglm/utils/basis.py.  The device never needs the (nT,N,B) impulse
features (the fused kernel rebuilds them from spike events); the functions here serve
(1) the interpolated basis tables uploaded to the device, and (2) the dense *stimulus*
features `data['fstim']` (bkgd.py:122-154, 303-340), which are computed once per data set.
"""
import numpy as np


def _log_time_cosines(n_pts, n_eye, n_cos, a, b):
    """Raised cosines on a log-warped axis (basis.py:71-93)."""
    u = np.log(a * np.arange(n_pts) + b)
    idx = np.floor(np.linspace(n_eye, n_pts / 2.0, n_cos)).astype(int)
    centers = u[idx]
    width = centers / 2 if n_cos == 1 else (centers[-1] - centers[0]) / (n_cos - 1)
    phase = np.clip((u[:, None] - centers[None, :]) * np.pi / width / 2.0, -np.pi, np.pi)
    return 0.5 * (np.cos(phase) + 1.0)


def orth_table_key(prms):
    """Key of an orthonormalised basis in the packaged tables (utils/basis_tables.npz)."""
    return '%s_eye%d_n%d_a%.9g_b%.9g' % (prms['type'].lower(), prms.get('n_eye', 0),
                                        prms.get('n_cos', prms.get('n_exp', 0)),
                                        prms.get('a', 0.0), prms.get('b', 0.0))


_TABLES = None


def _orth(basis, prms):
    """basis.py:97-98 orthonormalises with scipy.linalg.orth -- an SVD whose column signs depend on
    the LAPACK build.  The bases of the model templates therefore ship as
    package data produced by the reference's own create_basis (tools/make_basis_tables.py) and are
    looked up, never recomputed.  Other parameters get a deterministic numpy SVD basis of the same
    column space (largest-magnitude entry of every column made positive)."""
    global _TABLES
    if _TABLES is None:
        import os
        f = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'basis_tables.npz')
        _TABLES = dict(np.load(f)) if os.path.exists(f) else {}
    tab = _TABLES.get(orth_table_key(prms))
    if tab is not None and tab.shape[0] == basis.shape[0]:
        return tab.copy()
    u, sv, _ = np.linalg.svd(basis, full_matrices=False)
    rank = int(np.sum(sv > sv.max() * max(basis.shape) * np.finfo(float).eps))
    u = u[:, :rank]
    sign = np.sign(u[np.argmax(np.abs(u), axis=0), np.arange(rank)])
    return u * np.where(sign == 0, 1.0, sign)[None, :]


def _finish(basis, prms, n_norm):
    """Optional orthonormalisation (basis.py:97-98) and unit-area normalisation (99-104)."""
    if prms.get('orth', False):
        basis = _orth(basis, prms)
    if prms.get('norm', False):
        if np.any(basis < 0):
            raise Exception("We can only normalize nonnegative impulse responses!")
        basis = basis / basis.sum(axis=0, keepdims=True) * n_norm
    return basis


def create_cosine_basis(prms):
    """basis.py:56-106: n_eye identity columns + n_cos raised cosines at 100 points."""
    n_pts, n_eye, n_cos = 100, prms['n_eye'], prms['n_cos']
    basis = np.zeros((n_pts, n_eye + n_cos))
    basis[:n_eye, :n_eye] = np.eye(n_eye)
    basis[:, n_eye:] = _log_time_cosines(n_pts, n_eye, n_cos, prms['a'], prms['b'])
    return _finish(basis, prms, float(n_pts))


def create_exp_basis(prms):
    """basis.py:108-143: exponentials with log-spaced time constants."""
    n_pts, n_eye, n_exp = 100, prms['n_eye'], prms['n_exp']
    basis = np.zeros((n_pts, n_eye + n_exp))
    basis[:n_eye, :n_eye] = np.eye(n_eye)
    # note: basis.py:127 uses integer n_pts/3 under Python 2
    taus = np.logspace(np.log10(1), np.log10(n_pts // 3), n_exp)
    basis[:, n_eye:] = np.exp(-np.arange(n_pts)[:, None] / taus[None, :])
    return _finish(basis, prms, float(n_pts))


def create_gaussian_basis(prms):
    """basis.py:145-185: unit-variance Gaussian bumps on an integer grid."""
    n_gauss = tuple(prms['n_gauss'])
    n_eye = prms['n_eye']
    G = int(np.prod(n_gauss))
    basis = np.zeros((n_eye + G, n_eye + G))
    basis[:n_eye, :n_eye] = np.eye(n_eye)
    grid = np.array(np.unravel_index(np.arange(G), n_gauss)).T.astype(float)   # (G, ndim)
    d2 = ((grid[:, None, :] - grid[None, :, :]) ** 2).sum(-1)
    basis[n_eye:, n_eye:] = np.exp(-0.5 * d2)
    return _finish(basis, prms, 1.0)


def create_identity_basis(prms):
    """basis.py:187-199."""
    return np.eye(prms['n_eye'])


def create_basis(prms):
    """basis.py:9-26."""
    typ = prms['type'].lower()
    if typ == 'exp':
        return create_exp_basis(prms)
    if typ == 'cosine':
        return create_cosine_basis(prms)
    if typ == 'gaussian':
        return create_gaussian_basis(prms)
    if typ in ('identity', 'eye'):
        return create_identity_basis(prms)
    raise Exception("Unrecognized basis type: %s" % typ)


def interpolate_columns(basis, x_new, x_old):
    """Column-wise np.interp (impulse.py:99-103, bkgd.py:109-112)."""
    out = np.empty((len(x_new), basis.shape[1]))
    for j in range(basis.shape[1]):
        out[:, j] = np.interp(x_new, x_old, basis[:, j])
    return out


def n_taps(dt_max, dt):
    """R = dt_max/dt (float in the reference)."""
    return int(round(dt_max / dt))


def convolve_with_basis(stim, basis):
    """basis.py:201-236: out[t,d,b] = sum_{tau=1..R} stim[t-tau,d] * basis[tau-1,b]
    (strictly causal: a zero tap is prepended; first T samples of the full convolution)."""
    import scipy.signal as sig
    stim = np.asarray(stim, dtype=float)
    T, D = stim.shape
    R, B = basis.shape
    out = np.empty((T, D, B))
    for b in range(B):
        kern = np.concatenate(([0.0], basis[:, b]))[:, None]
        out[:, :, b] = sig.fftconvolve(stim, kern, 'full')[:T, :]
    return out


def convolve_with_low_rank_2d_basis(stim, basis_x, basis_t):
    """basis.py:238-273: project on the spatial basis, then causal temporal filtering.
    Returns (T, Bx, Bt)."""
    assert basis_x.shape[0] == stim.shape[1], \
        "ERROR: Spatial basis must be the same size as the stimulus"
    return convolve_with_basis(np.dot(stim, basis_x), basis_t)


def project_onto_basis(f, basis, lam=0):
    """basis.py:416-436: ridge-regularised least-squares coefficients (B,1)."""
    R, B = basis.shape
    f = np.asarray(f, dtype=float)
    assert f.shape[0] == R, "Function is not the same length as the basis!"
    if f.ndim == 1:
        f = f.reshape(R, 1)
    if lam == 0 and R == B and np.array_equal(basis, np.eye(R)):
        return f.copy()                         # identity basis (a pixel stimulus): the solve returns f itself
    return np.linalg.solve(basis.T.dot(basis) + lam * np.eye(B), basis.T.dot(f))
