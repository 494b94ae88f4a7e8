"""
ctypes binding of the C ABI declared in include/pyglm_hip.h (libpyglm_hip.so,
built from theano_pyglm_amd/csrc by __graft_entry__.build()).

There is NO CPU fallback: if the shared library is missing, or no HIP device is
visible, every compute entry point raises PglError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PYGLM_HIP_LIB', os.path.join(_HERE, 'libpyglm_hip.so'))   # env: dev A/B builds

NLIN_EXP = 0
NLIN_EXPLINEAR = 1
OPT_FEATURE_F32 = 1
OPT_NCHUNKS = 2
OPT_KERNEL = 3
OPT_GIBBS_KERNEL = 4
OPT_EPI_F64 = 5
OPT_TIMING = 6
OPT_BFGS_MERGE = 7

# every symbol include/pyglm_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    'pgl_last_error', 'pgl_version', 'pgl_device_count', 'pgl_create', 'pgl_destroy',
    'pgl_set_option', 'pgl_set_time_range', 'pgl_set_spikes_u8', 'pgl_set_spikes_f64', 'pgl_set_basis',
    'pgl_set_stim_features', 'pgl_set_stimulus', 'pgl_get_stim_features', 'pgl_ll_grad', 'pgl_ll_grad_dev', 'pgl_sync', 'pgl_features',
    'pgl_impulse_currents', 'pgl_state', 'pgl_ll_from_current', 'pgl_gibbs_prepare',
    'pgl_gibbs_ll', 'pgl_gibbs_update', 'pgl_last_timing', 'pgl_info', 'pgl_simulate', 'pgl_sta',
    'pgl_timing_summary', 'pgl_set_stream',
    'pgl_set_stimulus_separable', 'pgl_ll_grad_list_dev', 'pgl_gibbs_prepare_all', 'pgl_gibbs_ll_cols', 'pgl_gibbs_update_cols', 'pgl_gibbs_currents',
    'pgl_bfgs_state_doubles', 'pgl_bfgs_init_dev', 'pgl_bfgs_trial_dev', 'pgl_bfgs_objective_dev',
    'pgl_bfgs_linesearch_dev', 'pgl_bfgs_hmul_dev', 'pgl_bfgs_hmul_hist_dev', 'pgl_bfgs_update_dev', 'pgl_bfgs_step_dev', 'pgl_plan_kernels', 'pgl_leading_singular_pairs',
]


class PglError(RuntimeError):
    pass


def plan_kernels(N, B=5, R=200, Dstim=0, nT=300000, stim=0, n_lo=0, count=None, path=0, opt_kernel=0, opt_f32=0):
    """Names of the fused kernel instantiations the dispatcher would launch for this shape (pgl_plan_kernels: a dry run,
    works without a GPU); raises PglError where no plan exists."""
    lib = load()
    buf = C.create_string_buffer(4096)
    rc = lib.pgl_plan_kernels(int(N), int(B), int(R), int(Dstim), int(nT), int(stim), int(n_lo), int(N - n_lo if count is None else count),
                              int(path), int(opt_kernel), int(opt_f32), buf, 4096)
    if rc != 0:
        raise PglError(lib.pgl_last_error().decode())
    return [ln for ln in buf.value.decode().splitlines() if ln]


_lib = None


def _preload_torch_hip():
    """One HIP runtime per process.  PyTorch (used for device tensors / torch.distributed by
    bench.py, parallel.py and the GPU-resident optimizer) bundles its own libamdhip64 /
    libhsa-runtime64 with the same SONAMEs as the system ROCm ones this library links to.  If the
    system copies are mapped first, a later `import torch` resolves its HIP symbols against them
    and cannot initialise ("No HIP GPUs are available"); mapped in the other order everything
    works.  So when torch is installed its runtime is mapped before libpyglm_hip.so -- without
    importing torch."""
    if os.environ.get('PYGLM_NO_TORCH_PRELOAD'):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
        for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libpyglm_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PglError("HIP library %s not found: run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
                       % LIB_PATH)
    _preload_torch_hip()
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    lib.pgl_last_error.restype = C.c_char_p
    lib.pgl_last_error.argtypes = []
    lib.pgl_version.restype = C.c_int
    lib.pgl_device_count.restype = C.c_int
    lib.pgl_create.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                               C.POINTER(vp)]
    lib.pgl_destroy.argtypes = [vp]
    lib.pgl_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.pgl_set_time_range.argtypes = [vp, C.c_int64, C.c_int64]
    lib.pgl_set_spikes_u8.argtypes = [vp, vp]
    lib.pgl_set_spikes_f64.argtypes = [vp, vp]
    lib.pgl_set_basis.argtypes = [vp, vp]
    lib.pgl_set_stim_features.argtypes = [vp, vp, C.c_int]
    lib.pgl_set_stimulus.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_double, vp, C.c_int, vp, C.c_int,
                                     C.c_int, C.c_int]
    lib.pgl_get_stim_features.argtypes = [vp, vp]
    lib.pgl_set_stimulus_separable.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_double, vp, C.c_int, vp, C.c_int,
                                               C.c_int]
    lib.pgl_timing_summary.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.pgl_set_stream.argtypes = [vp, vp]
    lib.pgl_sta.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_double, C.c_int, vp, C.c_int, vp]
    if hasattr(lib, 'pgl_leading_singular_pairs'):
        lib.pgl_leading_singular_pairs.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.pgl_ll_grad.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.pgl_ll_grad_dev.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.pgl_ll_grad_list_dev.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    lib.pgl_sync.argtypes = [vp]
    if hasattr(lib, 'pgl_bfgs_hmul_dev'):                     # (older dev A/B builds named by PYGLM_HIP_LIB lack it)
        lib.pgl_bfgs_state_doubles.argtypes = [C.c_int, C.c_int]
        lib.pgl_bfgs_init_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double]
        lib.pgl_bfgs_trial_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
        lib.pgl_bfgs_objective_dev.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int] + [C.c_double] * 6
        lib.pgl_bfgs_linesearch_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int]
        lib.pgl_bfgs_hmul_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        lib.pgl_bfgs_update_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, vp, vp, C.c_int]
        lib.pgl_bfgs_hmul_hist_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
    if hasattr(lib, 'pgl_bfgs_step_dev'):
        lib.pgl_bfgs_step_dev.argtypes = ([vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int] + [C.c_double] * 6 +
                                          [C.c_int, C.c_double, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                           vp, vp, vp])
    lib.pgl_features.argtypes = [vp, vp]
    lib.pgl_impulse_currents.argtypes = [vp, vp, vp]
    lib.pgl_state.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.pgl_ll_from_current.argtypes = [vp, C.c_int, C.c_double, vp, vp, vp, vp, C.c_int, vp]
    lib.pgl_gibbs_prepare.argtypes = [vp, C.c_int, vp, vp]
    lib.pgl_gibbs_ll.argtypes = [vp, C.c_int, C.c_double, vp, C.c_int, vp]
    lib.pgl_gibbs_update.argtypes = [vp, C.c_int, C.c_double]
    lib.pgl_gibbs_prepare_all.argtypes = [vp, vp, vp]
    lib.pgl_gibbs_ll_cols.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    lib.pgl_gibbs_update_cols.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.pgl_gibbs_currents.argtypes = [vp, C.c_int, vp]
    lib.pgl_last_timing.argtypes = [vp, dp, dp]
    lib.pgl_info.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    if hasattr(lib, 'pgl_plan_kernels'):
        lib.pgl_plan_kernels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.pgl_simulate.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, vp, vp, vp,
                                 C.c_int64, C.c_uint64, vp, vp]
    for name in SYMBOLS:
        if 'PYGLM_HIP_LIB' in os.environ and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        if name == 'pgl_bfgs_state_doubles':
            fn.restype = C.c_longlong
        elif name not in ('pgl_last_error',):
            fn.restype = C.c_int
    _lib = lib
    return lib


def device_count():
    return load().pgl_device_count()


def _chk(rc):
    if rc != 0:
        raise PglError("libpyglm_hip error %d: %s" % (rc, load().pgl_last_error().decode()))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def simulate(X0, AW, nlin, dt, uniforms=None, seed=0):
    """Native Population.simulate (population.py:233-389); host code, needs no GPU.
    X0 (nT,N) background current, AW (N,R,N) [n_pre][tau][n_post].  Returns (S, X, n_exceptions)."""
    lib = load()
    X = np.array(X0, dtype=np.float64, order='C')
    nT, N = X.shape
    AW = _f64(AW)
    assert AW.ndim == 3 and AW.shape[0] == N and AW.shape[2] == N
    R = AW.shape[1]
    S = np.empty((nT, N))
    u = None if uniforms is None else _f64(uniforms)
    nexc = C.c_int64(0)
    kind = {'exp': NLIN_EXP, 'explinear': NLIN_EXPLINEAR}.get(nlin, nlin)
    _chk(lib.pgl_simulate(N, nT, R, int(kind), float(dt), _ptr(X), _ptr(AW), _ptr(u),
                          0 if u is None else u.size, int(seed), _ptr(S), C.byref(nexc)))
    return S, X, int(nexc.value)


class DeviceGlm(object):
    """One pgl_handle: the device-resident data of one data sequence
    (the Theano shared variables S / ir / stim of glm.py:17-18, impulse.py:41-42,
    bkgd.py:70-71) plus the kernels that evaluate glm.ll and its gradient on it."""

    def __init__(self, N, nT, B, R, nlin, dt, device=0):
        self.lib = load()
        self.N, self.nT, self.B, self.R, self.dt = int(N), int(nT), int(B), int(R), float(dt)
        self.nlin = {'exp': NLIN_EXP, 'explinear': NLIN_EXPLINEAR}.get(nlin, nlin)
        self.Dstim = 0
        h = C.c_void_p()
        _chk(self.lib.pgl_create(self.N, self.nT, self.B, self.R, int(self.nlin), self.dt,
                                 int(device), C.byref(h)))
        self.h = h

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.pgl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def P(self):
        return 1 + self.Dstim + self.N * self.B

    # -- data ---------------------------------------------------------------
    def set_option(self, opt, value):
        _chk(self.lib.pgl_set_option(self.h, int(opt), int(value)))

    def set_time_range(self, t_lo, t_hi):
        """Evaluate only bins [t_lo, t_hi) (partial ll / gradient; t_lo multiple of 16)."""
        _chk(self.lib.pgl_set_time_range(self.h, int(t_lo), int(t_hi)))

    def set_spikes(self, S):
        S = np.asarray(S)
        if S.shape != (self.nT, self.N):
            raise ValueError("S must be (nT,N)=(%d,%d), got %s" % (self.nT, self.N, S.shape))
        if S.dtype == np.uint8:
            S = np.ascontiguousarray(S)
            _chk(self.lib.pgl_set_spikes_u8(self.h, _ptr(S)))
        else:
            S = _f64(S)
            _chk(self.lib.pgl_set_spikes_f64(self.h, _ptr(S)))

    def set_basis(self, ibasis):
        ib = _f64(ibasis, (self.R, self.B))
        _chk(self.lib.pgl_set_basis(self.h, _ptr(ib)))

    def set_stim_features(self, fstim):
        if fstim is None:
            _chk(self.lib.pgl_set_stim_features(self.h, None, 0))
            self.Dstim = 0
            return
        f = _f64(fstim)
        if f.ndim != 2 or f.shape[0] != self.nT:
            raise ValueError("fstim must be (nT,Dstim)")
        _chk(self.lib.pgl_set_stim_features(self.h, _ptr(f), int(f.shape[1])))
        self.Dstim = int(f.shape[1])

    def set_stimulus(self, stim, dt_stim, basis_t, basis_x=None, layout=0):
        """Build the dense stimulus feature columns on the device from the raw stimulus
        (interp -> spatial projection -> causal temporal filtering); see pgl_set_stimulus."""
        stim = _f64(stim)
        if stim.ndim != 2:
            raise ValueError("stim must be (Tstim, D)")
        bt = _f64(basis_t)
        bx = None if basis_x is None else _f64(basis_x, (stim.shape[1], np.shape(basis_x)[1]))
        Bx = stim.shape[1] if bx is None else bx.shape[1]
        _chk(self.lib.pgl_set_stimulus(self.h, _ptr(stim), stim.shape[0], stim.shape[1], float(dt_stim),
                                       _ptr(bx), int(Bx), _ptr(bt), bt.shape[0], bt.shape[1],
                                       int(layout)))
        self.Dstim = int(Bx * bt.shape[1])

    def set_stimulus_separable(self, stim, dt_stim, basis_t, basis_x=None):
        """Rank-1 (w_t (x) w_x) stimulus kept separable on the device; theta rows become
        [bias, w_t, w_x, w_imp] (see pgl_set_stimulus_separable)."""
        stim = _f64(stim)
        if stim.ndim != 2:
            raise ValueError("stim must be (Tstim, D)")
        bt = _f64(basis_t)
        bx = None if basis_x is None else _f64(basis_x, (stim.shape[1], np.shape(basis_x)[1]))
        Bx = stim.shape[1] if bx is None else bx.shape[1]
        _chk(self.lib.pgl_set_stimulus_separable(self.h, _ptr(stim), stim.shape[0], stim.shape[1],
                                                 float(dt_stim), _ptr(bx), int(Bx), _ptr(bt), bt.shape[0],
                                                 bt.shape[1]))
        self.Dstim = int(Bx + bt.shape[1])

    def get_stim_features(self):
        out = np.empty((self.nT, self.Dstim))
        _chk(self.lib.pgl_get_stim_features(self.h, _ptr(out)))
        return out

    def sta(self, stim, dt_stim, L, Ns=None, keep_on_device=False):
        """Spike-triggered average (n_sel, L, D) of the resident spikes; see pgl_sta.  keep_on_device: the averages stay
        on the device for leading_singular_pairs(None, shape) and the shape (n_sel, L, D) is returned instead."""
        stim = _f64(stim)
        if stim.ndim != 2:
            raise ValueError("stim must be (Tstim, D)")
        sel = None if Ns is None else np.ascontiguousarray(np.atleast_1d(Ns), dtype=np.int32)
        nsel = self.N if sel is None else int(sel.size)
        out = None if keep_on_device else np.empty((nsel, int(L), stim.shape[1]))
        _chk(self.lib.pgl_sta(self.h, _ptr(stim), stim.shape[0], stim.shape[1], float(dt_stim), int(L),
                              _ptr(sel), nsel, _ptr(out)))
        return (nsel, int(L), int(stim.shape[1])) if keep_on_device else out

    def leading_singular_pairs(self, A, shape=None):
        """(U (n, L), sigma (n,), V (n, D)): leading singular pair of every matrix of the batch A (n, L, D); see
        pgl_leading_singular_pairs."""
        if A is None:                                      # the averages sta(keep_on_device=True) left on the device
            n, L, D = shape
        else:
            A = _f64(A)
            if A.ndim != 3:
                raise ValueError("A must be (n, L, D)")
            n, L, D = A.shape
        U, sig, V = np.empty((n, L)), np.empty(n), np.empty((n, D))
        _chk(self.lib.pgl_leading_singular_pairs(self.h, _ptr(A), n, L, D, _ptr(U), _ptr(sig), _ptr(V)))
        return U, sig, V

    # -- hot path -------------------------------------------------------------
    def ll_grad(self, theta, Weff, n_lo=0, n_hi=None, want_grad=True):
        n_hi = self.N if n_hi is None else n_hi
        npost = n_hi - n_lo
        th = _f64(theta, (npost, self.P))
        We = _f64(Weff, (self.N, self.N))
        ll = np.empty(npost)
        g = np.empty((npost, self.P)) if want_grad else None
        _chk(self.lib.pgl_ll_grad(self.h, int(n_lo), int(n_hi), _ptr(th), _ptr(We), _ptr(ll),
                                  _ptr(g)))
        return ll, g

    def ll_grad_dev(self, d_theta, d_Weff, d_ll, d_grad, n_lo=0, n_hi=None):
        """Device-pointer form (integers, e.g. torch.Tensor.data_ptr()); asynchronous."""
        n_hi = self.N if n_hi is None else n_hi
        _chk(self.lib.pgl_ll_grad_dev(self.h, int(n_lo), int(n_hi), C.c_void_p(d_theta),
                                      C.c_void_p(d_Weff), C.c_void_p(d_ll),
                                      C.c_void_p(d_grad) if d_grad else None))

    def ll_grad_list_dev(self, d_idx, count, d_theta, d_Weff, d_ll, d_grad):
        """ll+grad of the `count` neurons listed in d_idx (device int32 pointer); row j of the device
        arrays belongs to neuron d_idx[j].  Asynchronous like ll_grad_dev."""
        _chk(self.lib.pgl_ll_grad_list_dev(self.h, C.c_void_p(d_idx), int(count), C.c_void_p(d_theta),
                                           C.c_void_p(d_Weff), C.c_void_p(d_ll),
                                           C.c_void_p(d_grad) if d_grad else None))

    # -- lock-step optimiser bookkeeping (device pointers as integers; asynchronous on the handle's stream) --
    def bfgs_state_doubles(self, M, P):
        return int(self.lib.pgl_bfgs_state_doubles(int(M), int(P)))

    def bfgs_trial_dev(self, d_state, M, P, d_rows, L, d_Xt):
        _chk(self.lib.pgl_bfgs_trial_dev(self.h, C.c_void_p(d_state), int(M), int(P),
                                         C.c_void_p(d_rows) if d_rows else None, int(L), C.c_void_p(d_Xt)))

    def bfgs_objective_dev(self, L, P, d_Xt, d_ll_f, d_grad_g, prior_kind, mu_b, sg_b, stim_sigma, mu, sigma, lam):
        _chk(self.lib.pgl_bfgs_objective_dev(self.h, int(L), int(P), C.c_void_p(d_Xt), C.c_void_p(d_ll_f),
                                             C.c_void_p(d_grad_g), int(prior_kind), float(mu_b), float(sg_b),
                                             float(stim_sigma), float(mu), float(sigma), float(lam)))

    def bfgs_init_dev(self, d_state, M, P, gtol):
        _chk(self.lib.pgl_bfgs_init_dev(self.h, C.c_void_p(d_state), int(M), int(P), float(gtol)))

    def bfgs_linesearch_dev(self, d_state, M, P, d_rows, L, d_Xt, d_f, d_g, max_trials=100):
        _chk(self.lib.pgl_bfgs_linesearch_dev(self.h, C.c_void_p(d_state), int(M), int(P),
                                              C.c_void_p(d_rows) if d_rows else None, int(L), C.c_void_p(d_Xt),
                                              C.c_void_p(d_f), C.c_void_p(d_g), int(max_trials)))

    def bfgs_hmul_dev(self, d_state, M, P, d_rows, L, d_H, ld):
        _chk(self.lib.pgl_bfgs_hmul_dev(self.h, C.c_void_p(d_state), int(M), int(P),
                                        C.c_void_p(d_rows) if d_rows else None, int(L), C.c_void_p(d_H), int(ld)))

    def bfgs_hmul_hist_dev(self, d_state, M, P, d_rows, L, d_hist, d_coef, Kmax, d_ab):
        _chk(self.lib.pgl_bfgs_hmul_hist_dev(self.h, C.c_void_p(d_state), int(M), int(P),
                                             C.c_void_p(d_rows) if d_rows else None, int(L), C.c_void_p(d_hist),
                                             C.c_void_p(d_coef), int(Kmax), C.c_void_p(d_ab)))

    def bfgs_update_dev(self, d_state, M, P, gtol, maxiter, init_scaling=False, d_hist=0, d_coef=0, Kmax=0):
        _chk(self.lib.pgl_bfgs_update_dev(self.h, C.c_void_p(d_state), int(M), int(P), float(gtol), int(maxiter),
                                          1 if init_scaling else 0, C.c_void_p(d_hist) if d_hist else None,
                                          C.c_void_p(d_coef) if d_coef else None, int(Kmax)))

    def bfgs_step_dev(self, d_state, M, P, d_rows, L, d_Xt, d_f, d_g, prior=None, max_trials=100, gtol=1e-5, maxiter=225,
                      init_scaling=False, d_hist=0, d_coef=0, Kmax=0, d_ab=0, hk_bound=0, d_H=0, ld=0, d_pos_next=0,
                      d_Xt_next=0, flags_out=0):
        """One whole optimiser iteration behind an evaluation (pgl_bfgs_step_dev).  prior: the tuple of
        bfgs_objective_dev's prior arguments, or None when d_f / d_g already hold the objective and its gradient."""
        vp = lambda a: C.c_void_p(a) if a else None
        pr = (-1, 0.0, 1.0, 1.0, 0.0, 1.0, 0.0) if prior is None else prior
        _chk(self.lib.pgl_bfgs_step_dev(self.h, C.c_void_p(d_state), int(M), int(P), vp(d_rows), int(L), C.c_void_p(d_Xt),
                                        C.c_void_p(d_f), C.c_void_p(d_g), int(pr[0]), *[float(z) for z in pr[1:]],
                                        int(max_trials), float(gtol), int(maxiter), 1 if init_scaling else 0, vp(d_hist),
                                        vp(d_coef), int(Kmax), vp(d_ab), int(hk_bound), vp(d_H), int(ld), vp(d_pos_next),
                                        vp(d_Xt_next), vp(flags_out)))

    def sync(self):
        _chk(self.lib.pgl_sync(self.h))

    def last_timing(self):
        a, b = C.c_double(), C.c_double()
        _chk(self.lib.pgl_last_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_summary(self, reset=True):
        """(n_launches, mean fused-kernel ms, mean whole-call ms) since the last reset."""
        n, a, b = C.c_int(), C.c_double(), C.c_double()
        _chk(self.lib.pgl_timing_summary(self.h, 1 if reset else 0, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def set_stream(self, stream_ptr):
        """Order the handle's work on a caller-owned HIP stream (integer hipStream_t of a
        torch.cuda.Stream(): `.cuda_stream`); None returns to the handle's own stream.
        The NULL stream cannot be named through this call (pgl_set_stream(NULL) means "own stream"):
        torch's DEFAULT stream has handle 0, so `torch.cuda.current_stream().cuda_stream` is only
        valid inside `with torch.cuda.stream(torch.cuda.Stream())` -- a 0 raises instead of silently
        un-ordering the kernels from the caller's collectives."""
        if stream_ptr is None:
            _chk(self.lib.pgl_set_stream(self.h, None))
            return
        if int(stream_ptr) == 0:
            raise ValueError("stream handle 0 is the NULL stream: create a torch.cuda.Stream(), make it "
                             "current and pass its .cuda_stream")
        _chk(self.lib.pgl_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def info(self, n_lo=0, n_hi=None):
        n_hi = self.N if n_hi is None else n_hi
        v = np.zeros(13)
        _chk(self.lib.pgl_info(self.h, int(n_lo), int(n_hi), _ptr(v), 13))
        keys = ['blocks', 'threads', 'chunks', 'ktiles', 'lds_bytes', 'tile_rows', 'flops',
                'bytes', 'events', 'kernel_version', 'resident_feature_bytes', 'streamed_bytes', 'stim_path']
        return dict(zip(keys, v.tolist()))

    # -- direct-form helpers --------------------------------------------------
    def features(self):
        out = np.empty((self.nT, self.N, self.B))
        _chk(self.lib.pgl_features(self.h, _ptr(out)))
        return out

    def impulse_currents(self, w):
        w = _f64(w, (self.N, self.B))
        out = np.empty((self.nT, self.N))
        _chk(self.lib.pgl_impulse_currents(self.h, _ptr(w), _ptr(out)))
        return out

    def state(self, n, theta_n, Weff_col):
        th = _f64(theta_n, (self.P,))
        wc = _f64(Weff_col, (self.N,))
        lam = np.empty(self.nT)
        inet = np.empty(self.nT)
        istim = np.empty(self.nT)
        _chk(self.lib.pgl_state(self.h, int(n), _ptr(th), _ptr(wc), _ptr(lam), _ptr(inet),
                                _ptr(istim)))
        return lam, inet, istim

    def ll_from_current(self, n_post, I_bias, I_stim, I_other, I_col, ws):
        ws = _f64(np.atleast_1d(ws))
        I_other = _f64(I_other, (self.nT,))
        I_col = _f64(I_col, (self.nT,))
        I_stim = None if I_stim is None else _f64(I_stim, (self.nT,))
        out = np.empty(len(ws))
        _chk(self.lib.pgl_ll_from_current(self.h, int(n_post), float(I_bias), _ptr(I_stim),
                                          _ptr(I_other), _ptr(I_col), _ptr(ws), len(ws),
                                          _ptr(out)))
        return out

    def gibbs_prepare(self, n_post, theta_n, Weff_col):
        th = _f64(theta_n, (self.P,))
        wc = _f64(Weff_col, (self.N,))
        _chk(self.lib.pgl_gibbs_prepare(self.h, int(n_post), _ptr(th), _ptr(wc)))

    def gibbs_ll(self, n_pre, aw_cur, ws):
        ws = _f64(np.atleast_1d(ws))
        out = np.empty(len(ws))
        _chk(self.lib.pgl_gibbs_ll(self.h, int(n_pre), float(aw_cur), _ptr(ws), len(ws),
                                   _ptr(out)))
        return out

    def gibbs_update(self, n_pre, delta):
        _chk(self.lib.pgl_gibbs_update(self.h, int(n_pre), float(delta)))

    # -- batched column Gibbs (all post-synaptic columns resident) ---------------
    def gibbs_prepare_all(self, theta, Weff):
        th = _f64(theta, (self.N, self.P))
        We = _f64(Weff, (self.N, self.N))
        _chk(self.lib.pgl_gibbs_prepare_all(self.h, _ptr(th), _ptr(We)))

    def gibbs_ll_cols(self, n_post, n_pre, aw_cur, ws):
        """ll (ncols, K) at the candidate weights ws (ncols, K) of the pairs (n_pre[c], n_post[c])."""
        n_post = np.ascontiguousarray(n_post, dtype=np.int32)
        n_pre = np.ascontiguousarray(n_pre, dtype=np.int32)
        ws = _f64(ws)
        if ws.ndim == 1:
            ws = ws[None, :]
        aw = _f64(aw_cur, (len(n_post),))
        if n_pre.shape != n_post.shape or ws.shape[0] != len(n_post):
            raise ValueError("n_post, n_pre, aw_cur, ws must describe the same columns")
        out = np.empty(ws.shape)
        K = ws.shape[1]
        for k0 in range(0, K, 16):                     # at most 16 candidate weights per launch
            wk = np.ascontiguousarray(ws[:, k0:k0 + 16])
            ok = np.empty(wk.shape)
            _chk(self.lib.pgl_gibbs_ll_cols(self.h, len(n_post), _ptr(n_post), _ptr(n_pre), _ptr(aw),
                                            _ptr(wk), wk.shape[1], _ptr(ok)))
            out[:, k0:k0 + 16] = ok
        return out

    def gibbs_update_cols(self, n_post, n_pre, delta):
        n_post = np.ascontiguousarray(n_post, dtype=np.int32)
        n_pre = np.ascontiguousarray(n_pre, dtype=np.int32)
        d = _f64(delta, (len(n_post),))
        if len(n_post) == 0:
            return
        _chk(self.lib.pgl_gibbs_update_cols(self.h, len(n_post), _ptr(n_post), _ptr(n_pre), _ptr(d)))

    def gibbs_currents(self, n_post, nrows=None):
        out = np.empty(self.nT if nrows is None else int(nrows))
        _chk(self.lib.pgl_gibbs_currents(self.h, int(n_post), _ptr(out)))
        return out
