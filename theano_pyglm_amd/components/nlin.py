"""Rate nonlinearities (counterpart of pyglm/components/nlin.py).  The device kernels
implement the same two functions (PGL_NLIN_EXP / PGL_NLIN_EXPLINEAR); f_nlin is the numpy
twin the simulator uses (nlin.py:28, 47) in its overflow-safe form."""
import numpy as np

from theano_pyglm_amd.components.component import Component


def create_nlin_component(model):
    typ = model['nonlinearity']['type'].lower()
    if typ == 'exp':
        return ExpNonlinearity(model)
    if typ == 'explinear':
        return ExpLinearNonlinearity(model)
    raise Exception("Unrecognized nonlinearity model: %s" % typ)


class ExpNonlinearity(Component):
    kind = 'exp'

    def __init__(self, model=None):
        self.f_nlin = np.exp


class ExpLinearNonlinearity(Component):
    """lam = log(1+exp(x)) (nlin.py:43)."""
    kind = 'explinear'

    def __init__(self, model=None):
        self.f_nlin = lambda x: np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))
