"""
make_model / stabilize_sparsity / check_stability -- counterpart of
pyglm/models/model_factory.py:18-185 for the templates on the hot path.
"""
import copy

import numpy as np

from theano_pyglm_amd.models import templates as _t

_TEMPLATES = {
    'standard_glm': _t.standard_glm, 'standardglm': _t.standard_glm,
    'spatiotemporal_glm': _t.spatiotemporal_glm,
    'sparse_weighted_model': _t.sparse_weighted_model,
    'sparseweightedmodel': _t.sparse_weighted_model,
    # BASELINE.json calls the sparse-coupling-prior model "network_glm"; the reference has
    # no such template (model_factory.py:23-44) -- it is sparse_weighted_model.
    'network_glm': _t.sparse_weighted_model,
}


def make_model(template, N=None, dt=None):
    """model_factory.py:18-67: build from a template name or dict, override N and dt."""
    if isinstance(template, str):
        key = template.lower()
        if key not in _TEMPLATES:
            raise Exception("Unrecognized template model: %s!" % template)
        model = _TEMPLATES[key]()
    elif isinstance(template, dict):
        model = copy.deepcopy(template)
    else:
        raise Exception("Unrecognized template model!")
    if N is not None:
        model['N'] = N
    if dt is not None:
        model['dt'] = dt
    return model


def stabilize_sparsity(model):
    """model_factory.py:69-102 (Erdos-Renyi + Gaussian weights): choose rho so that the
    spectral radius sqrt(N*rho)*sigma stays below 1 - delta (+|refractory mean|)."""
    graph = model['network']['graph']
    weight = model['network']['weight']
    if graph['type'].lower() in ('erdos_renyi', 'erdosrenyi'):
        if weight.get('prior', {}).get('type', '').lower() == 'gaussian':
            maxeig = 1.0 - 0.3
            if 'refractory_prior' in weight:
                maxeig -= weight['refractory_prior']['mu']
            sigma = weight['prior']['sigma']
            graph['rho'] = float(min(maxeig ** 2 / model['N'] / sigma ** 2, 1.0))
    return model


def check_stability(model, x, N):
    """model_factory.py:173-185."""
    if model['network']['weight']['type'].lower() == 'gaussian':
        Weff = x['net']['graph']['A'] * np.reshape(x['net']['weights']['W'], (N, N))
        return bool(np.amax(np.real(np.linalg.eigvals(Weff))) < 1)
    return True
