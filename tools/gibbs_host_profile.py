"""cProfile of one batched collapsed-Gibbs sweep at the C4 shape: where the HOST time goes (dev tool)."""
import sys, cProfile, pstats, io
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.inference import gibbs as G
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
N, nT = 128, 600000
model = make_model('sparse_weighted_model', N=N, dt=0.001)
stabilize_sparsity(model)
popn = Population(model)
rng = np.random.default_rng(1238)
S = np.minimum(rng.poisson(20.0 * 0.001, size=(nT, N)), 10).astype(np.uint8)
popn.add_data({'S': S, 'N': N, 'dt': 0.001, 'T': nT * 0.001, 'stim': None, 'dt_stim': 0.1})
x = popn.sample(np.random.RandomState(4))
x['net']['weights']['W'] = 0.2 * np.asarray(x['net']['weights']['W'])
upd = G.CollapsedGibbsNetworkColumnUpdate(rng=np.random.RandomState(2))
upd.preprocess(popn)
upd.update_all(x)
pr = cProfile.Profile(); pr.enable()
upd.update_all(x)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
