"""Firing-rate nonlinearities lam = f(x) of the GLM.

Two kinds exist in the reference (pyglm/components/nlin.py): 'exp' (lam = e^x, :25) and
'explinear' (lam = log(1 + e^x), :43).  The HIP kernels implement both (PGL_NLIN_EXP /
PGL_NLIN_EXPLINEAR); `f_nlin` is the numpy twin used by the host-side simulator, written in the
overflow-safe form max(x,0) + log1p(exp(-|x|))."""
import numpy as np

from theano_pyglm_amd.components.component import Component


def _softplus(x):
    x = np.asarray(x, dtype=float)
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


class _Nonlinearity(Component):
    kind = None
    f_nlin = None

    def __init__(self, model=None):
        pass


class ExpNonlinearity(_Nonlinearity):
    kind = 'exp'
    f_nlin = staticmethod(np.exp)


class ExpLinearNonlinearity(_Nonlinearity):
    kind = 'explinear'
    f_nlin = staticmethod(_softplus)


_KINDS = {'exp': ExpNonlinearity, 'explinear': ExpLinearNonlinearity}


def create_nlin_component(model):
    kind = model['nonlinearity']['type'].lower()
    if kind not in _KINDS:
        raise Exception("Unrecognized nonlinearity model: %s" % kind)
    return _KINDS[kind](model)
