"""
Network = graph (adjacency A) + weights (W) -- counterpart of
pyglm/components/network.py, graph.py:27-86 and weights.py:21-100.
W_eff[n_pre, n_post] = A * W (glm.py:31-37) is what the device kernels consume.
"""
import numpy as np

from theano_pyglm_amd.components.component import Component
from theano_pyglm_amd.components.priors import create_prior, _rng


def create_graph_component(model, latent):
    typ = model['network']['graph']['type'].lower()
    if typ == 'complete':
        return CompleteGraphModel(model)
    if typ in ('erdos_renyi', 'erdosrenyi'):
        return ErdosRenyiGraphModel(model)
    raise Exception("Unsupported graph model on the MI355X hot path: %s" % typ)


def create_weight_component(model, latent):
    typ = model['network']['weight']['type'].lower()
    if typ == 'constant':
        return ConstantWeightModel(model)
    if typ == 'gaussian':
        return GaussianWeightModel(model)
    raise Exception("Unrecognized weight model: %s" % typ)


class CompleteGraphModel(Component):
    """graph.py:27-41: A = ones."""

    def __init__(self, model):
        self.N = model['N']

    def A(self, vars):
        return np.ones((self.N, self.N))

    def get_state(self, vars=None):
        return {'A': self.A(vars)}


class ErdosRenyiGraphModel(Component):
    """graph.py:44-86: A int8 (N,N), Bernoulli(rho) with rho_refractory on the diagonal."""

    def __init__(self, model):
        self.prms = model['network']['graph']
        self.N = N = model['N']
        self.rho = self.prms['rho'] * np.ones((N, N))
        if 'rho_refractory' in self.prms:
            self.rho[np.diag_indices(N)] = self.prms['rho_refractory']
        self.pA = self.rho
        self.lkhd_scale = 1.0

    def get_variables(self):
        return {'A': (self.N, self.N)}

    def A(self, vars):
        return np.asarray(vars['A']).reshape(self.N, self.N).astype(float)

    def log_p(self, vars):
        A = self.A(vars)
        lk = np.sum(A * np.log(np.minimum(1.0 - 1e-8, self.rho)) +
                    (1 - A) * np.log(np.maximum(1e-8, 1.0 - self.rho)))
        return self.lkhd_scale * lk

    def sample(self, acc, rng=None):
        A = _rng(rng).random_sample((self.N, self.N)) < self.rho
        return {'A': A.astype(np.int8)}

    def get_state(self, vars=None):
        return {} if vars is None else {'A': self.A(vars)}


class ConstantWeightModel(Component):
    """weights.py:21-38: W = value * ones."""

    def __init__(self, model):
        self.N = model['N']
        self.value = model['network']['weight']['value']

    def W(self, vars):
        return self.value * np.ones((self.N, self.N))

    def get_state(self, vars=None):
        return {'W': self.W(vars)}


class GaussianWeightModel(Component):
    """weights.py:40-100: W_flat (N*N,) row-major [n_pre, n_post]; Gaussian prior off the
    diagonal, refractory prior on it."""

    def __init__(self, model):
        self.N = N = model['N']
        prms = model['network']['weight']
        self.prior = create_prior(prms['prior'])
        self.refractory_prior = create_prior(prms['refractory_prior']) \
            if 'refractory_prior' in prms else None
        self._diag = np.eye(N, dtype=bool)

    def get_variables(self):
        return {'W': (self.N * self.N,)}

    def W(self, vars):
        return np.asarray(vars['W'], dtype=float).reshape(self.N, self.N)

    def log_p(self, vars):
        W = self.W(vars)
        if self.refractory_prior is None:
            return self.prior.log_p(W)
        return self.prior.log_p(W[~self._diag]) + self.refractory_prior.log_p(W[self._diag])

    def grad_log_p(self, vars):
        W = self.W(vars)
        g = self.prior.grad_log_p(W)
        if self.refractory_prior is not None:
            g = np.where(self._diag, self.refractory_prior.grad_log_p(W), g)
        return {'W': g.reshape(-1)}

    def sample(self, acc, rng=None):
        N = self.N
        if self.refractory_prior is None:
            return {'W': np.asarray(self.prior.sample(None, (N * N,), rng=rng))}
        W = np.zeros((N, N))
        W[self._diag] = self.refractory_prior.sample(None, (N,), rng=rng)
        W[~self._diag] = self.prior.sample(None, (N * N - N,), rng=rng)
        return {'W': W.reshape(-1)}

    def get_state(self, vars=None):
        return {} if vars is None else {'W': self.W(vars)}


class Network(Component):
    """network.py:4-53."""

    def __init__(self, model, latent):
        self.model = model
        self.latent = latent
        self.prms = model['network']
        self.graph = create_graph_component(model, latent)
        self.weights = create_weight_component(model, latent)

    def get_variables(self):
        return {'graph': self.graph.get_variables(), 'weights': self.weights.get_variables()}

    def log_p(self, vars):
        return self.graph.log_p(vars['graph']) + self.weights.log_p(vars['weights'])

    def W_eff(self, vars):
        """A * W, (N,N) indexed [n_pre, n_post] (glm.py:31-37)."""
        return self.graph.A(vars['graph']) * self.weights.W(vars['weights'])

    def get_state(self, vars=None):
        if vars is None:
            return {'graph': {}, 'weights': {}}
        return {'graph': {'A': self.graph.A(vars['graph'])},
                'weights': {'W': self.weights.W(vars['weights'])}}

    def sample(self, acc, rng=None):
        return {'graph': self.graph.sample(acc, rng=rng), 'weights': self.weights.sample(acc, rng=rng)}

    def set_hyperparameters(self, model):
        pass
