"""Categorical draw from unnormalised log probabilities (counterpart of
pyglm/inference/log_sum_exp.py:4-37)."""
import numpy as np


def log_sum_exp_sample(lnp, rng=None):
    lnp = np.asarray(lnp, dtype=float).ravel()
    r = np.random if rng is None else rng
    m = np.amax(lnp)
    if not np.isfinite(m):
        raise Exception("log_sum_exp_sample: no finite entry")
    p = np.exp(lnp - m)
    p = p / p.sum()
    return int(np.searchsorted(np.cumsum(p), r.random_sample()))
