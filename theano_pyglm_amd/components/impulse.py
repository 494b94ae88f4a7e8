"""
Impulse-response (coupling filter) models -- counterpart of
pyglm/components/impulse.py:18-133 (LinearBasisImpulses) and :254-397 (DirichletImpulses).

The (nT,N,B) feature tensor data['fS'] of the reference (impulse.py:114-130) is not
materialised: the device rebuilds feature tiles from spike events, so preprocess_data
only validates shapes.  What the component owns is the interpolated basis `ibasis`
(uploaded to the device) and the map variables -> flat impulse weights w[n_pre*B+b].
"""
import numpy as np

from theano_pyglm_amd.components.component import Component
from theano_pyglm_amd.components.priors import create_prior, Gaussian, _rng
from theano_pyglm_amd.utils import basis as bs


def create_impulse_component(model, glm, latent):
    typ = model['impulse']['type'].lower()
    if typ == 'basis':
        return LinearBasisImpulses(model)
    if typ == 'dirichlet':
        return DirichletImpulses(model)
    raise Exception("Unsupported impulse model on the MI355X hot path: %s "
                    "(basis and dirichlet are implemented)" % typ)


class _ImpulseBase(Component):
    def preprocess_data(self, data):
        nT, Ns = data['S'].shape
        assert Ns == self.N, "ERROR: Spike train must be (TxN) dimensional where N=%d" % self.N

    def impulse(self, vars):
        """(N,R) impulse responses w . ibasis^T (impulse.py:65 / 329)."""
        return self.flat_weights(vars).reshape(self.N, self.B).dot(self.ibasis.T)

    def get_state(self, vars=None):
        st = {'basis': self.ibasis}
        if vars is not None:
            st['impulse'] = self.impulse(vars)
        return st


class LinearBasisImpulses(_ImpulseBase):
    """I_imp[t,n'] = sum_b fS[t,n',b] w_ir[n',b] (impulse.py:58); prior on w_ir (N,B)."""

    def __init__(self, model):
        self.model = model
        self.imp_model = model['impulse']
        if 'prior' in self.imp_model:
            self.prior = create_prior(self.imp_model['prior'])
        else:
            # reference templates without a 'prior' key cannot be built (impulse.py:25 KeyError,
            # SURVEY Appendix B #7); use the block's own mu/sigma (the legacy form, impulse.py:31-32)
            self.prior = Gaussian({'mu': self.imp_model['mu'], 'sigma': self.imp_model['sigma']})
        self.N = model['N']
        self.basis = bs.create_basis(self.imp_model['basis'])
        self.B = self.basis.shape[1]
        R = bs.n_taps(self.imp_model['dt_max'], model['dt'])
        ib = bs.interpolate_columns(self.basis, np.linspace(0, 1, R),
                                    np.linspace(0, 1, self.basis.shape[0]))    # impulse.py:99-103
        if self.imp_model['basis']['norm']:
            ib = ib / self.imp_model['dt_max']                                 # impulse.py:106-107
        self.ibasis = ib

    def get_variables(self):
        return {'w_ir': (self.N * self.B,)}

    def flat_weights(self, vars):
        return np.asarray(vars['w_ir'], dtype=float).reshape(-1)

    def chain_grad(self, vars, g_flat):
        return {'w_ir': np.array(g_flat, dtype=float)}

    def log_p(self, vars):
        return self.prior.log_p(self.flat_weights(vars).reshape(self.N, self.B))

    def log_p_all(self, vars_list):
        # both priors (priors.py:139 / 202) are sums over the (presynaptic neuron, basis) groups: all neurons' groups at once
        if len(vars_list) == 0:
            return 0.0
        W = np.stack([self.flat_weights(v) for v in vars_list])
        return float(self.prior.log_p(W.reshape(len(vars_list) * self.N, self.B)))

    def grad_log_p(self, vars):
        g = self.prior.grad_log_p(self.flat_weights(vars).reshape(self.N, self.B))
        return {'w_ir': np.asarray(g).reshape(-1)}

    def set_hyperparameters(self, model):
        if 'prior' in model:
            self.prior.set_hyperparameters(model['prior'])

    def sample(self, acc, rng=None):
        return {'w_ir': np.asarray(self.prior.sample(None, size=(self.N, self.B), rng=rng)).ravel()}


class DirichletImpulses(_ImpulseBase):
    """Normalised impulse responses: beta_n' = |g_n'| / sum|g_n'| (impulse.py:286-291),
    same current form (:308); prior sum_n' (alpha-1) sum log|g| - sum|g| (:320-322)."""

    def __init__(self, model):
        self.model = model
        self.imp_model = model['impulse']
        self.N = model['N']
        self.alpha = self.imp_model['alpha']
        self.basis = bs.create_basis(self.imp_model['basis'])
        self.B = self.basis.shape[1]
        dt, dt_max = model['dt'], self.imp_model['dt_max']
        t_int = np.arange(0.0, dt_max, step=dt)                                 # impulse.py:365
        t_bas = np.linspace(0.0, dt_max, self.basis.shape[0])
        ib = bs.interpolate_columns(self.basis, t_int, t_bas)
        if self.imp_model['basis']['norm']:
            trapz = getattr(np, 'trapezoid', None) or np.trapz
            ib = ib / trapz(ib, t_int, axis=0)                                  # impulse.py:373-374
        self.ibasis = ib

    def get_variables(self):
        return dict(('g_%d' % n, (self.B,)) for n in range(self.N))

    def _g(self, vars):
        return np.array([np.asarray(vars['g_%d' % n], dtype=float) for n in range(self.N)])

    def flat_weights(self, vars):
        ga = np.abs(self._g(vars))
        return (ga / ga.sum(axis=1, keepdims=True)).reshape(-1)

    def chain_grad(self, vars, g_flat):
        """d beta_b / d g_c = sign(g_c) (delta_bc s - |g_b|) / s^2, s = sum|g|."""
        g = self._g(vars)
        ga = np.abs(g)
        s = ga.sum(axis=1, keepdims=True)
        gb = np.asarray(g_flat, dtype=float).reshape(self.N, self.B)
        inner = np.sum(gb * ga, axis=1, keepdims=True)
        dg = np.sign(g) * (gb * s - inner) / s ** 2
        return dict(('g_%d' % n, dg[n]) for n in range(self.N))

    def log_p(self, vars):
        ga = np.abs(self._g(vars))
        return float(np.sum((self.alpha - 1.0) * np.sum(np.log(ga), axis=1) - np.sum(ga, axis=1)))

    def grad_log_p(self, vars):
        g = self._g(vars)
        dg = (self.alpha - 1.0) / g - np.sign(g)
        return dict(('g_%d' % n, dg[n]) for n in range(self.N))

    def sample(self, acc, rng=None):
        r = _rng(rng)
        return dict(('g_%d' % n, r.gamma(self.alpha, np.ones(self.B))) for n in range(self.N))
