"""
Population -- connected GLMs (counterpart of pyglm/population.py).

Same construction and method surface as the reference (`Population(model)`, `add_data`,
`set_data`, `sample`, `extract_vars`, `compute_log_p`, `compute_log_prior`, `compute_ll`,
`eval_state`, `simulate`) plus the gradient entry points the reference builds with
T.grad inside its inference code (coord_descent.py:27-30): `compute_ll_grad` (all
neurons in one fused device pass) and `compute_grad(vars, n)`.

Every data sequence is uploaded to the GPU once (spike counts as uint8 + the
spike-event index, the interpolated impulse basis, dense stimulus features) and stays
resident; `set_data` only switches the current handle -- the reference re-copies the
whole feature tensor into Theano shared variables on every evaluation
(coord_descent.py:52-57, glm.py:99-110).  There is no CPU fallback.
"""
import os

import numpy as np

from theano_pyglm_amd import _lib
from theano_pyglm_amd.components.latent import LatentVariables
from theano_pyglm_amd.components.network import Network
from theano_pyglm_amd.glm import Glm
from theano_pyglm_amd.utils.packvec import packdict, get_vars
from theano_pyglm_amd.utils.syms import from_shapes, differentiable, check_bound, compile_bound, compile_paths, check_paths


class Population(object):
    def __init__(self, model, device=None):
        """population.py:12-32."""
        self.model = model
        self.N = model['N']
        self.data_sequences = []
        self.latent = LatentVariables(model)
        self.network = Network(model, self.latent)
        self.glm = Glm(model, self.network, self.latent)
        self.device = int(os.environ.get('PYGLM_DEVICE', '0')) if device is None else int(device)
        self._handles = []        # (data dict, _lib.DeviceGlm) pairs: the pair keeps the dict alive, so a
                                  # recycled id() can never alias a stale handle; lookup is by identity
        self._current = None      # data dict conditioned on (set_data)
        self._time_shard = None   # (rank, world): every handle evaluates only its share of the bins

    # -- variables ------------------------------------------------------------
    def get_variables(self):
        """population.py:133-140."""
        return {'latent': {},
                'net': from_shapes(self.network.get_variables(), int_keys=('A',)),
                'glm': self.glm.get_variables()}

    def set_hyperparameters(self, model):
        """population.py:142-147."""
        self.network.set_hyperparameters(model)
        self.glm.set_hyperparameters(model)

    def sample(self, rng=None):
        """population.py:149-162: latent -> net -> glms[0..N-1]."""
        v = {'latent': {}}
        v['net'] = self.network.sample(v, rng=rng)
        v['glms'] = []
        for n in range(self.N):
            xn = self.glm.sample(v, rng=rng)
            xn['n'] = n
            v['glms'].append(xn)
        return v

    def shape_vars(self):
        """A value dictionary laid out like sample() with zeros of every symbol's shape -- for callers that need the
        packing shapes only (the reference draws a sample for that, coord_descent.py:24-26, 93-95; here the shapes come from
        the symbol table: 2 x 2.5 ms of Python per sweep at N = 128, and no random numbers consumed)."""
        def zeros(syms):
            out = {}
            for k, v in syms.items():
                out[k] = zeros(v) if isinstance(v, dict) else np.zeros(v.shape, dtype=v.dtype)
            return out
        syms = self.get_variables()
        v = {'latent': {}, 'net': zeros(syms['net']), 'glms': []}
        z = zeros(dict((k, s) for k, s in syms['glm'].items() if k != 'n'))
        for n in range(self.N):
            xn = dict(z)                                      # (the neurons share the zero arrays: shapes are all that is read)
            xn['n'] = n
            v['glms'].append(xn)
        return v

    def extract_vars(self, vals, n):
        """population.py:164-175."""
        out = {}
        for k, v in vals.items():
            if k == 'glms':
                out['glm'] = v[n]
            else:
                out[k] = v
        return out

    # -- data -------------------------------------------------------------------
    def preprocess_data(self, data):
        """population.py:187-196 (the impulse features fS are NOT materialised)."""
        assert isinstance(data, dict), 'Data must be a dictionary'
        self.glm.preprocess_data(data)
        data['preprocessed'] = True
        return data

    def add_data(self, data, set_as_current_data=True):
        """population.py:198-221."""
        assert isinstance(data, dict), 'Data must be a dictionary'
        assert 'S' in data, 'Data must contain an array of spike times'
        assert isinstance(data['S'], np.ndarray), 'Spike times must be a numpy array'
        if 'preprocessed' not in data or not data['preprocessed']:
            data = self.preprocess_data(data)
        self.data_sequences.append(data)
        if set_as_current_data:
            self.set_data(data)

    def _find_handle(self, data):
        for d, h in self._handles:
            if d is data:
                return h
        return None

    def _handle(self, data):
        h = self._find_handle(data)
        if h is None:
            S = np.asarray(data['S'])
            nT, N = S.shape
            assert N == self.N, "ERROR: Spike train must be (TxN) dimensional where N=%d" % self.N
            imp = self.glm.imp_model
            h = _lib.DeviceGlm(N, nT, imp.B, imp.ibasis.shape[0], self.glm.nlin_model.kind,
                               self.glm.dt, device=self.device)
            h.set_spikes(S)
            h.set_basis(imp.ibasis)
            # no per-evaluation HIP events on the product path: an event between two kernels of a stream idles the GPU
            # for ~6 us (pgl_last_timing is for the measurement tools, which drive DeviceGlm directly)
            h.set_option(_lib.OPT_TIMING, 0)
            if self.glm.Dstim > 0:
                if data.get('fstim', None) is not None:
                    if getattr(self.glm.bkgd_model, 'separable', False):
                        raise ValueError("data['fstim'] (dense stimulus features) cannot be used with the separable "
                                         "stimulus path: theta rows carry [w_t, w_x] there; drop 'fstim' (the features are "
                                         "built from data['stim'] on the device) or set model['bkgd']['separable'] = False")
                    h.set_stim_features(data['fstim'])      # caller-supplied dense features
                else:
                    self.glm.bkgd_model.upload(h, data)     # built on the device from data['stim']
            if self._time_shard is not None:
                self._apply_time_shard(h)
            self._handles.append((data, h))
        return h

    def _apply_time_shard(self, h):
        from theano_pyglm_amd.parallel import time_shard_bounds
        if self._time_shard is None:
            h.set_time_range(0, h.nT)
        else:
            rank, world = self._time_shard
            h.set_time_range(*time_shard_bounds(h.nT, rank, world))

    def set_time_shard(self, rank=None, world=None):
        """Restrict every data sequence to the bins of time shard `rank` of `world` (boundaries on the
        kernels' 16-bin tile grid; the likelihood is additive over time segments, population.py:41-43, so
        the partial (ll, grad) of the shards all-reduce to the full evaluation).  None: the whole
        recording.  Used by the time-sharded multi-GPU MAP (inference/parallel_coord_descent.py)."""
        self._time_shard = None if rank is None or world in (None, 1) else (int(rank), int(world))
        for _, h in self._handles:
            self._apply_time_shard(h)

    def set_data(self, data):
        """population.py:223-231: condition on `data` (switches the device-resident handle)."""
        assert 'preprocessed' in data and data['preprocessed'] == True, \
            'Data must be preprocessed before it can be set'
        self._handle(data)
        self._current = data

    def stim_features(self, data=None):
        """The dense stimulus feature columns `fstim` (nT, Dstim) of a data sequence (the reference
        keeps them in data['fstim'], bkgd.py:154 / 340); here they live on the device and are
        copied out on request."""
        data = self._current if data is None else data
        if self.glm.Dstim == 0:
            return None
        if getattr(self.glm.bkgd_model, 'separable', False):     # no dense matrix on the device: host twin
            return self.glm.bkgd_model.host_features(data, np.asarray(data['S']).shape[0])
        return self._handle(data).get_stim_features()

    def release_data(self, data=None):
        """Free the device buffers of one data sequence (or of all of them).

        Ownership: the population keeps a strong reference to every data dict it has been conditioned on
        (`add_data`, `set_data`, and internally every evaluation on a held-out dict) together with its device
        handle -- spikes, event index and up to three sets of resident feature tiles (3 GB at N=128, T=600 s) --
        until this method is called; the reference's shared variables are simply overwritten by the next
        set_data (glm.py:99-110).  Transient data sets (cross-validation splits, initialisation populations)
        should be released explicitly, as harness/synth_map_with_xv.py and inference/gibbs.py do."""
        keep = []
        for d, h in self._handles:
            if data is None or d is data:
                h.close()
                if self._current is d:
                    self._current = None
            else:
                keep.append((d, h))
        self._handles = keep

    # -- parameters -> device layout ----------------------------------------------
    def theta_matrix(self, vars, n_lo=0, n_hi=None):
        n_hi = self.N if n_hi is None else n_hi
        return self.glm.theta_rows([vars['glms'][n] for n in range(n_lo, n_hi)])

    def W_eff(self, vars):
        return self.network.W_eff(vars['net'])

    def _check_vars(self, vars, n=None):
        """Mirror of seval's binding check (theano_func_wrapper.py:69-105): every symbol of the network and of the listed
        neuron's GLM (n None: of every neuron) must have a value.  The symbol tables are those of the model and do not
        change: built once."""
        cache = getattr(self, '_bound_syms', None)
        if cache is None:
            syms = self.get_variables()
            net_tree = compile_bound({'net': syms['net']})
            glm_tree = compile_bound(dict((k, v) for k, v in syms['glm'].items() if k != 'n'))
            cache = (net_tree, compile_paths(net_tree), glm_tree, compile_paths(glm_tree))
            self._bound_syms = cache
        net_tree, net_paths, glm_tree, glm_paths = cache
        check_paths(net_paths, net_tree, vars)
        glms = vars['glms']
        for m in (range(self.N) if n is None else (n,)):
            check_paths(glm_paths, glm_tree, glms[m])

    # -- log probability ---------------------------------------------------------------
    def compute_log_prior(self, vars):
        """population.py:47-69: latent + network + sum_n glm.log_prior."""
        self._check_vars(vars)
        return self._log_prior(vars)

    def _log_prior(self, vars):
        lp = 0.0
        lp += self.latent.log_p(vars.get('latent', {}))
        lp += self.network.log_p(vars['net'])
        lp += self.glm.log_prior_all([vars['glms'][n] for n in range(self.N)])
        return lp

    def compute_ll_vector(self, vars, n_lo=0, n_hi=None):
        """Per-neuron ll of the CURRENT data sequence for neurons [n_lo, n_hi)."""
        n_hi = self.N if n_hi is None else n_hi
        if self._current is None:
            raise Exception("No data sequence has been set")
        h = self._handle(self._current)
        ll, _ = h.ll_grad(self.theta_matrix(vars, n_lo, n_hi), self.W_eff(vars), n_lo, n_hi,
                          want_grad=False)
        return ll

    def compute_ll(self, vars):
        """population.py:71-86: sum over neurons of glm.ll on the current data."""
        self._check_vars(vars)
        return float(np.sum(self.compute_ll_vector(vars)))

    def compute_log_p(self, vars):
        """population.py:34-45 (lkhd_scale is not applied here, like the reference).  The variables are checked once for
        the prior and all data sequences."""
        self._check_vars(vars)
        lp = 0.0
        lp += self._log_prior(vars)
        for data in self.data_sequences:
            self.set_data(data)
            lp += float(np.sum(self.compute_ll_vector(vars)))
        return lp

    # -- gradients (the reference: T.grad in coord_descent.py:27-30) --------------------
    def compute_ll_grad(self, vars, n_lo=0, n_hi=None):
        """ll_n and d ll_n / d(flat feature weights) for n in [n_lo,n_hi) on the current data:
        returns (ll (npost,), g_theta (npost,P)) straight from the fused device pass."""
        n_hi = self.N if n_hi is None else n_hi
        h = self._handle(self._current)
        return h.ll_grad(self.theta_matrix(vars, n_lo, n_hi), self.W_eff(vars), n_lo, n_hi)

    def glm_syms(self):
        """differentiable(syms['glm']) (coord_descent.py:24)."""
        return differentiable(self.get_variables()['glm'])

    def compute_lp_grad_packed(self, vars, n_lo=0, n_hi=None, include_prior=True):
        """For every neuron n in [n_lo,n_hi): log posterior of its GLM parameters
        (log_prior_n + sum_data ll_n) and the gradient w.r.t. the packed per-neuron vector
        (layout = packdict of the differentiable GLM variables).
        Returns (lp (npost,), grads (npost, P_packed))."""
        n_hi = self.N if n_hi is None else n_hi
        syms = self.glm_syms()
        lps = np.zeros(n_hi - n_lo)
        grads = None
        for data in self.data_sequences:
            self.set_data(data)
            ll, g_theta = self.compute_ll_grad(vars, n_lo, n_hi)
            lps += ll
            for i, n in enumerate(range(n_lo, n_hi)):
                xn = vars['glms'][n]
                gd = self.glm.chain_grad(xn, g_theta[i])
                gv, _ = packdict(get_vars(syms, gd))
                if grads is None:
                    grads = np.zeros((n_hi - n_lo, gv.size))
                grads[i] += gv
        if include_prior:
            for i, n in enumerate(range(n_lo, n_hi)):
                xn = vars['glms'][n]
                lps[i] += self.glm.log_prior(xn)
                gv, _ = packdict(get_vars(syms, self.glm.grad_log_prior(xn)))
                grads[i] += gv
        return lps, grads

    def compute_grad(self, vars, n):
        """Gradient of (glm.log_prior + sum_data glm.ll) of neuron n w.r.t. its packed
        parameter vector -- minus coord_descent.grad_nlp (coord_descent.py:61-80)."""
        return self.compute_lp_grad_packed(vars, n, n + 1)[1][0]

    # -- state ---------------------------------------------------------------------------
    def eval_state(self, vars):
        """population.py:88-120: rates, currents and component state of every neuron."""
        state = {'latent': {}, 'net': self.network.get_state(vars['net'])}
        h = self._handle(self._current)
        W = self.W_eff(vars)
        glm_states = []
        for n in range(self.N):
            xn = vars['glms'][n]
            st = self.glm.get_state(xn)
            lam, inet, istim = h.state(n, self.glm.theta_row(xn), W[:, n])
            st['lam'] = lam
            st['I_bias'] = self.glm.bias_model.I_bias(xn['bias'])
            st['I_bkgd'] = istim if self.glm.Dstim > 0 else 0.0
            st['I_net'] = inet
            glm_states.append(st)
        state['glms'] = glm_states
        state['logprior'] = self.compute_log_prior(vars)
        state['ll'] = self.compute_ll(vars)
        state['logp'] = state['ll'] + state['logprior']
        return state

    # -- simulation ------------------------------------------------------------------------
    def simulate(self, vars, T_range, dt, stim, dt_stim, rng=None, verbose=False, native=True):
        """population.py:233-389: integrate-and-fire thinning with exponential thresholds;
        every spike adds A*W*impulse to X[t+1 : t+R+1] (351-353); <= 10 spikes per bin.
        Seeded (the reference uses the global np.random).  native=True runs the time loop in the
        C++ library (pgl_simulate, ~100x the Python loop, same draw order); native=False is the
        line-by-line Python restatement kept for cross-checking."""
        from theano_pyglm_amd.components.priors import _rng
        r = _rng(rng)
        T_start, T_stop = T_range
        N = self.N
        nT = len(np.arange(T_start, T_stop, dt))
        X = np.zeros((nT, N))
        for n in range(N):
            X[:, n] = self.glm.bias_model.I_bias(vars['glms'][n]['bias'])
        if self.glm.Dstim > 0:
            tmp = {'S': np.zeros((nT, N)), 'stim': stim, 'dt_stim': dt_stim, 'T': float(T_stop - T_start),
                   'dt': dt}
            fst = self.glm.bkgd_model.host_features(tmp, nT)
            for n in range(N):
                X[:, n] += fst.dot(self.glm.bkgd_model.dense_weights(vars['glms'][n]['bkgd']))
        # imps[n_pre, n_post, :] (population.py:275-282)
        imps = np.array([self.glm.imp_model.impulse(vars['glms'][n]['imp']) for n in range(N)])
        imps = np.transpose(imps, axes=[1, 0, 2])
        T_imp = imps.shape[2]
        AW = self.W_eff(vars)[:, :, None] * imps              # (n_pre, n_post, R)
        if native:
            # uniforms in the reference's draw order come from `rng`; the library continues with
            # its own generator if the pre-drawn stream runs out
            n_draw = int(N + 4 * max(1.0, np.sum(self.glm.nlin_model.f_nlin(X)) * dt) + 1000)
            u = r.random_sample(min(n_draw, 50000000))
            seed = int(r.randint(0, 2 ** 31 - 1))
            S, X, n_exc = _lib.simulate(X, np.ascontiguousarray(np.transpose(AW, (0, 2, 1))),
                                        self.glm.nlin_model.kind, dt, uniforms=u, seed=seed)
            if verbose:
                print("Number of exceptions arising from multiple spikes per bin: %d" % n_exc)
            return S, X
        f_nlin = self.glm.nlin_model.f_nlin
        S = np.zeros((nT, N))
        acc = np.zeros(N)
        thr = -np.log(r.random_sample(N))
        n_exceptions = 0
        for t in range(nT):
            acc = acc + f_nlin(X[t, :]) * dt
            i_spk = acc > thr
            S[t, i_spk] += 1
            n_spk = int(np.sum(i_spk))
            t_imp = min(nT - t - 1, T_imp)
            while n_spk > 0:
                if np.any(S[t, :] >= 10):
                    n_exceptions += 1
                    break
                X[t + 1:t + t_imp + 1, :] += np.sum(AW[i_spk, :, :t_imp], 0).T
                acc -= thr * i_spk
                acc[acc < 0] = 0
                thr[i_spk] = -np.log(r.random_sample(n_spk))
                i_spk = acc > thr
                S[t, i_spk] += 1
                n_spk = int(np.sum(i_spk))
        if verbose:
            print("Number of exceptions arising from multiple spikes per bin: %d" % n_exceptions)
        return S, X
