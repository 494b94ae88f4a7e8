"""ctypes wrapper of oracle/glm_oracle.c (test infrastructure / cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libglm_oracle.so')
_lib = None


def load():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ('glm_oracle.c', 'glm_blocked.c', 'Makefile')]
        if not os.path.exists(_SO) or any(os.path.getmtime(f) > os.path.getmtime(_SO) for f in srcs):
            subprocess.check_call(['make', '-s', '-C', _HERE])
        _lib = C.CDLL(_SO)
        _lib.oracle_ll_grad_neuron.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def features(S_u8, ibasis):
    lib = load()
    S = np.ascontiguousarray(S_u8, dtype=np.uint8)
    ib = np.ascontiguousarray(ibasis, dtype=np.float64)
    nT, N = S.shape
    R, B = ib.shape
    out = np.empty((nT, N, B))
    lib.oracle_features(_p(S), C.c_int64(nT), C.c_int(N), _p(ib), C.c_int(R), C.c_int(B), _p(out))
    return out


def ll_grad(S_u8, fS, theta, Weff, kind, dt, n_lo=0, n_hi=None, fstim=None, threads=1,
            want_grad=True):
    lib = load()
    S = np.ascontiguousarray(S_u8, dtype=np.uint8)
    nT, N = S.shape
    B = fS.shape[2]
    n_hi = N if n_hi is None else n_hi
    Dstim = 0 if fstim is None else fstim.shape[1]
    P = 1 + Dstim + N * B
    th = np.ascontiguousarray(theta, dtype=np.float64)
    assert th.shape == (n_hi - n_lo, P)
    We = np.ascontiguousarray(Weff, dtype=np.float64)
    fS = np.ascontiguousarray(fS, dtype=np.float64)
    fst = None if fstim is None else np.ascontiguousarray(fstim, dtype=np.float64)
    ll = np.empty(n_hi - n_lo)
    g = np.empty((n_hi - n_lo, P)) if want_grad else None
    lib.oracle_ll_grad(C.c_int(n_lo), C.c_int(n_hi), _p(S), C.c_int64(nT), C.c_int(N), C.c_int(B),
                       _p(fS), _p(fst), C.c_int(Dstim), _p(th), _p(We),
                       C.c_int(1 if kind == 'explinear' else 0), C.c_double(dt), _p(ll), _p(g),
                       C.c_int(threads))
    return ll, g


def ll_grad_blocked(S_u8, fS, theta, Weff, kind, dt, fstim=None, threads=1, want_grad=True):
    """B2: all neurons in one time-tiled sweep over fS (oracle/glm_blocked.c), OpenMP over time."""
    lib = load()
    S = np.ascontiguousarray(S_u8, dtype=np.uint8)
    nT, N = S.shape
    B = fS.shape[2]
    Dstim = 0 if fstim is None else fstim.shape[1]
    P = 1 + Dstim + N * B
    th = np.ascontiguousarray(theta, dtype=np.float64)
    assert th.shape == (N, P)
    We = np.ascontiguousarray(Weff, dtype=np.float64)
    fS = np.ascontiguousarray(fS, dtype=np.float64)
    fst = None if fstim is None else np.ascontiguousarray(fstim, dtype=np.float64)
    ll = np.empty(N)
    g = np.empty((N, P)) if want_grad else None
    lib.oracle_ll_grad_blocked(_p(S), C.c_int64(nT), C.c_int(N), C.c_int(B), _p(fS), _p(fst),
                               C.c_int(Dstim), _p(th), _p(We), C.c_int(1 if kind == 'explinear' else 0),
                               C.c_double(dt), _p(ll), _p(g), C.c_int(threads))
    return ll, g


def set_data_copy_seconds(fS, S_f64):
    """Seconds one Glm.set_data copy of (fS, S) takes (glm.py:99-110), single thread."""
    import time
    lib = load()
    fS = np.ascontiguousarray(fS, dtype=np.float64)
    Sf = np.ascontiguousarray(S_f64, dtype=np.float64)
    dst = np.empty(fS.size + Sf.size)
    lib.oracle_set_data_copy(_p(fS), C.c_size_t(fS.size), _p(Sf), C.c_size_t(Sf.size), _p(dst))   # first touch
    t0 = time.perf_counter()
    lib.oracle_set_data_copy(_p(fS), C.c_size_t(fS.size), _p(Sf), C.c_size_t(Sf.size), _p(dst))
    return time.perf_counter() - t0
