"""Constant bias current (counterpart of pyglm/components/bias.py)."""
import numpy as np

from theano_pyglm_amd.components.component import Component
from theano_pyglm_amd.components.priors import _rng


def create_bias_component(model, glm, latent):
    typ = model['bias']['type'].lower()
    if typ == 'constant':
        return ConstantBias(model)
    raise Exception("Unrecognized bias model: %s" % typ)


class ConstantBias(Component):
    """I_bias = bias[0] (bias.py:32), log_p = -0.5/sigma^2 (bias-mu)^2 (bias.py:33)."""

    def __init__(self, model):
        prms = model['bias']
        self.mu_bias = prms['mu']
        self.sig_bias = prms['sigma']

    def get_variables(self):
        return {'bias': (1,)}

    def I_bias(self, vars):
        return float(np.asarray(vars['bias']).reshape(-1)[0])

    def log_p(self, vars):
        return -0.5 / self.sig_bias ** 2 * (self.I_bias(vars) - self.mu_bias) ** 2

    def grad_log_p(self, vars):
        return {'bias': np.array([-(self.I_bias(vars) - self.mu_bias) / self.sig_bias ** 2])}

    def set_hyperparameters(self, model):
        self.mu_bias = model['mu']
        self.sig_bias = model['sigma']

    def get_state(self, vars=None):
        return {} if vars is None else {'bias': vars['bias']}

    def sample(self, acc, rng=None):
        return {'bias': self.mu_bias + self.sig_bias * _rng(rng).standard_normal(1)}
