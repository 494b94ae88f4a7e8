"""Latent-variable container (counterpart of pyglm/components/latent.py:26-81).
Only the behaviour of a model WITHOUT a 'latent' block is on the hot path: empty
variable set, log_p = 0 (latent.py:35-40)."""
from theano_pyglm_amd.components.component import Component


class LatentVariables(Component):
    def __init__(self, model):
        self.model = model
        self.latent_model = model.get('latent', {})
        if self.latent_model:
            raise Exception("latent-variable components (types/locations) are outside the "
                            "MI355X hot path; use a model without a 'latent' block")
        self.latentlist = []
        self.latentdict = {}

    def log_p(self, vars):
        return 0.0

    def __getitem__(self, item):
        return self.latentdict[item]
