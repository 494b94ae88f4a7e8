"""
theano_pyglm_amd -- MI355X-native population-GLM log-likelihood / gradient hot path
behind the Population / Glm API of slinderman/theano_pyglm.
"""
__version__ = '0.1.0'
