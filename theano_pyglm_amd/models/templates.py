"""
Model templates: nested dicts of hyper-parameters, deep-copied and overridden by
make_model(name, N=, dt=).  Values follow pyglm/models/standard_glm.py:4-87,
sparse_weighted_model.py:4-107 and spatiotemporal_glm.py:4-92 so that every shape
and prior of the named configurations is the reference's.
"""


def _cosine(n_cos, orth, norm):
    return {'type': 'cosine', 'n_eye': 0, 'n_cos': n_cos, 'a': 1.0 / 120, 'b': 0.5,
            'orth': orth, 'norm': norm}


def _gaussian(mu, sigma):
    return {'type': 'gaussian', 'mu': mu, 'sigma': sigma}


def standard_glm():
    """explinear, bias N(20,0.1), no stimulus, 5 orthogonal cosine impulse bases over
    200 ms under a group-lasso prior, constant unit weights on a complete graph."""
    return {
        'N': 2,
        'nonlinearity': {'type': 'explinear'},
        'bias': {'type': 'constant', 'mu': 20, 'sigma': 0.1},
        'bkgd': {'type': 'none', 'D_stim': 1, 'dt_max': 0.3,
                 'prior': {'type': 'spherical_gaussian', 'mu': 0.0, 'sigma': 0.01},
                 'basis': _cosine(3, True, False)},
        'impulse': {'type': 'basis', 'dt_max': 0.2,
                    'prior': {'type': 'group_lasso', 'mu': 0.0, 'sigma': 10.0, 'lam': 1.0},
                    'basis': _cosine(5, True, False)},
        'network': {'weight': {'type': 'constant', 'value': 1.0},
                    'graph': {'type': 'complete'}},
    }


def sparse_weighted_model():
    """explinear, bias N(20,0.25), Dirichlet(alpha=1) impulses on 5 normalised cosines,
    Gaussian weights (refractory diagonal N(-0.2,0.5)) on an Erdos-Renyi graph."""
    return {
        'N': 1,
        'nonlinearity': {'type': 'explinear'},
        'bias': {'type': 'constant', 'mu': 20.0, 'sigma': 0.25},
        'bkgd': {'type': 'no_stimulus', 'D_stim': 1, 'dt_max': 0.3, 'mu': 0, 'sigma': 0.5,
                 'basis': _cosine(3, False, True)},
        'impulse': {'type': 'dirichlet', 'dt_max': 0.2, 'alpha': 1,
                    'basis': _cosine(5, False, True)},
        'network': {'weight': {'type': 'gaussian', 'prior': _gaussian(0.0, 1.0),
                               'refractory_prior': _gaussian(-0.2, 0.5)},
                    'graph': {'type': 'erdos_renyi', 'rho': 0.5, 'rho_refractory': 1.0}},
    }


def spatiotemporal_glm():
    """exp nonlinearity, bias N(1,1), rank-1 spatiotemporal stimulus filter (3 temporal
    cosines x identity spatial basis, D_stim=3), 3 normalised impulse cosines over 300 ms.
    The reference template has no impulse 'prior' key although LinearBasisImpulses reads
    one (impulse.py:25): the impulse component falls back to the
    Gaussian(mu, sigma) given by the block's own 'mu'/'sigma'."""
    return {
        'N': 2,
        'nonlinearity': {'type': 'exp'},
        'bias': {'type': 'constant', 'mu': 1.0, 'sigma': 1.0},
        'bkgd': {'type': 'spatiotemporal', 'D_stim': 3, 'dt_max': 0.3, 'mu': 0.0, 'sigma': 1.5,
                 'temporal_basis': _cosine(3, False, True),
                 'spatial_basis': {'type': 'identity', 'n_eye': 3}},
        'impulse': {'type': 'basis', 'dt_max': 0.3, 'mu': 0, 'sigma': 0.001,
                    'basis': _cosine(3, False, True)},
        'network': {'weight': {'type': 'constant', 'value': 1.0},
                    'graph': {'type': 'complete'}},
    }


StandardGlm = standard_glm()
SparseWeightedModel = sparse_weighted_model()
SpatiotemporalGlm = spatiotemporal_glm()
