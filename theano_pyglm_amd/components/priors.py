"""Priors on the hot path (counterpart of pyglm/components/priors.py:125-224)."""
import numpy as np

from theano_pyglm_amd.components.component import Component


def _rng(rng):
    return np.random if rng is None else rng


def create_prior(model, **kwargs):
    typ = model['type'].lower()
    if typ in ('normal', 'gaussian'):
        return Gaussian(model, **kwargs)
    if typ in ('group_lasso', 'grouplasso'):
        return GroupLasso(model, **kwargs)
    raise Exception("Unrecognized prior type: %s" % typ)


class Gaussian(Component):
    """priors.py:125-158: log_p = -0.5/sigma^2 * sum((value-mu)^2)."""

    def __init__(self, model, name='gaussian'):
        self.prms = model
        self.mu = model['mu']
        self.sigma = model['sigma']

    def log_p(self, value):
        return -0.5 / self.sigma ** 2 * np.sum((np.asarray(value) - self.mu) ** 2)

    def grad_log_p(self, value):
        return -(np.asarray(value) - self.mu) / self.sigma ** 2

    def set_hyperparameters(self, model):
        self.mu = model['mu']
        self.sigma = model['sigma']

    def sample(self, acc, size=(1,), rng=None):
        return self.mu + self.sigma * _rng(rng).standard_normal(size)


class GroupLasso(Component):
    """priors.py:188-224: log_p = -lam * sum_groups ||(value-mu)/sigma||_2, value (groups,B)."""

    def __init__(self, model, name='gaussian'):
        self.prms = model
        self.lam = model['lam']
        self.mu = model['mu']
        self.sigma = model['sigma']

    def log_p(self, value):
        z = (np.asarray(value) - self.mu) / self.sigma
        return -1.0 * self.lam * np.sum(np.sqrt(np.einsum('ij,ij->i', z, z)))

    def grad_log_p(self, value):
        """A zero group yields 0/0 = NaN like T.grad of the sqrt in the reference; callers
        zero NaN gradients (coord_descent.py:179-180)."""
        z = (np.asarray(value) - self.mu) / self.sigma
        nrm = np.sqrt(np.sum(z ** 2, axis=1, keepdims=True))
        with np.errstate(invalid='ignore', divide='ignore'):
            return -self.lam * z / nrm / self.sigma

    def set_hyperparameters(self, model):
        self.mu = model['mu']
        self.sigma = model['sigma']
        self.lam = model['lam']

    def sample(self, acc, size=(1,), rng=None):
        """priors.py:215-224: Gaussian direction scaled to a Laplace(0, lam) group norm."""
        r = _rng(rng)
        n = size[0]
        norms = r.laplace(0, self.lam, size=(n, 1))
        v = self.mu + self.sigma * r.standard_normal(size)
        return v * norms / np.sqrt(np.sum(v ** 2, axis=1)).reshape(n, 1)
