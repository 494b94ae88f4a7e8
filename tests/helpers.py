"""Shared synthetic-problem builders for the parity tests (test infrastructure)."""
import os
import numpy as np

from oracle import glm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'basis_golden.npz'))


def std_ibasis(R=200):
    """standard_glm impulse basis (golden 100-point table) interpolated to R taps."""
    return O.interp_basis_unit(golden()['std_imp_basis'], R)


def st_ibasis(R=300):
    """spatiotemporal_glm impulse basis (B=3, norm) interpolated to R taps, /dt_max."""
    return O.interp_basis_unit(golden()['st_imp_basis'], R) / (R * 0.001)


class Problem(object):
    """A seeded synthetic population problem in the flat feature-weight layout."""

    def __init__(self, N, nT, ibasis, kind='explinear', dt=0.001, rate_hz=20.0, Dstim=0,
                 seed=0, weighted=False, bias_mu=None, w_scale=None):
        rng = np.random.default_rng(seed)
        self.N, self.nT, self.dt, self.kind = N, nT, dt, kind
        self.ibasis = np.ascontiguousarray(ibasis)
        self.R, self.B = ibasis.shape
        self.Dstim = Dstim
        S = rng.poisson(rate_hz * dt, size=(nT, N))
        S = np.minimum(S, 10).astype(np.uint8)
        self.S = S
        self.fstim = rng.standard_normal((nT, Dstim)) if Dstim > 0 else None
        P = 1 + Dstim + N * self.B
        self.P = P
        if bias_mu is None:
            bias_mu = 20.0 if kind == 'explinear' else 1.0
        if w_scale is None:
            w_scale = 2.0 if kind == 'explinear' else 0.05
        theta = np.zeros((N, P))
        theta[:, 0] = bias_mu + 0.3 * rng.standard_normal(N)
        if Dstim > 0:
            theta[:, 1:1 + Dstim] = (0.1 if Dstim < 64 else 0.01) * rng.standard_normal((N, Dstim))
        theta[:, 1 + Dstim:] = w_scale * rng.standard_normal((N, N * self.B))
        self.theta = theta
        if weighted:
            A = (rng.random((N, N)) < 0.5).astype(float)
            W = rng.standard_normal((N, N))
            self.Weff = A * W
        else:
            self.Weff = np.ones((N, N))
        self._fS = None

    @property
    def fS(self):
        if self._fS is None:
            self._fS = O.convolve_with_basis_fft(self.S.astype(float), self.ibasis)
        return self._fS

    def oracle_ll_grad(self, n_lo=0, n_hi=None):
        n_hi = self.N if n_hi is None else n_hi
        Sf = self.S.astype(float)
        lls = np.zeros(n_hi - n_lo)
        grads = np.zeros((n_hi - n_lo, self.P))
        D = self.Dstim
        for i, n in enumerate(range(n_lo, n_hi)):
            th = self.theta[n]
            w_imp = th[1 + D:].reshape(self.N, self.B)
            w_stim = th[1:1 + D] if D > 0 else None
            ll, gb, gs, gw = O.glm_ll_grad(n, Sf, self.fS, w_imp, self.Weff[:, n], th[0], self.dt,
                                           self.kind, self.fstim, w_stim)
            lls[i] = ll
            grads[i, 0] = gb
            if D > 0:
                grads[i, 1:1 + D] = gs
            grads[i, 1 + D:] = gw.reshape(-1)
        return lls, grads

    def device(self, device=0, f32=False, nchunks=0):
        from theano_pyglm_amd import _lib
        d = _lib.DeviceGlm(self.N, self.nT, self.B, self.R, self.kind, self.dt, device)
        d.set_spikes(self.S)
        d.set_basis(self.ibasis)
        if self.Dstim > 0:
            d.set_stim_features(self.fstim)
        if f32:
            d.set_option(_lib.OPT_FEATURE_F32, 1)
        if nchunks:
            d.set_option(_lib.OPT_NCHUNKS, nchunks)
        return d


def rel_err(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
