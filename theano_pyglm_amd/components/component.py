"""Base class of the model components (counterpart of pyglm/components/component.py).

The reference's components hold Theano expression fragments; here each component
evaluates its own piece numerically on the host (priors, parameter <-> flat feature
weight maps and their chain rules), while the time-series arithmetic that touches the
data lives in the HIP kernels."""


class Component(object):
    def get_variables(self):
        """{name: shape} of the variables owned by this component."""
        return {}

    def get_state(self, vars=None):
        return {}

    def preprocess_data(self, data):
        pass

    def set_data(self, data):
        pass

    def set_hyperparameters(self, model):
        pass

    def sample(self, acc, rng=None):
        return {}

    # host-evaluated log prior and its gradient w.r.t. this component's variables
    def log_p(self, vars):
        return 0.0

    def log_p_all(self, vars_list):
        """Sum of log_p over the neurons' copies of this component's variables (Population.compute_log_prior evaluates the
        N per-neuron priors of population.py:47-69 in one go; components whose prior is a plain array expression override it)."""
        lp = 0.0
        for v in vars_list:
            lp += self.log_p(v)
        return lp

    def grad_log_p(self, vars):
        return {}
