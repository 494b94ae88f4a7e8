from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity, check_stability  # noqa: F401
