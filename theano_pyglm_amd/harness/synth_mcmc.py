"""
Network-column Gibbs sweeps on synthetic data -- counterpart of test/synth_mcmc.py
restricted to the hot path: the CollapsedGibbsNetworkColumnUpdate sweep over all
columns (gibbs.py:2548-2551) with the model's other parameters held fixed (the HMC
updates of bias / impulse weights depend on the un-vendored `hips` package and are
out of scope).  Non-interactive (the reference prompts with raw_input).

    python -m theano_pyglm_amd.harness.synth_mcmc -d data.pkl [-n 10]
"""
import argparse
import pickle
import time

import numpy as np

from theano_pyglm_amd.inference.gibbs import CollapsedGibbsNetworkColumnUpdate
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population


def gibbs_network_sweeps(popn, x0, N_samples=10, rng=None, callback=None):
    upd = CollapsedGibbsNetworkColumnUpdate(rng=rng)
    upd.preprocess(popn)
    x = x0
    lps = []
    for smpl in range(N_samples):
        t0 = time.time()
        lp = popn.compute_log_p(x)
        lps.append(lp)
        for n in range(popn.N):
            upd.update(x, n)
        if callback is not None:
            callback(x)
        print("Gibbs iteration %d. Iter/s = %f. Log prob: %.3f" % (smpl, 1.0 / (time.time() - t0), lp))
    return x, lps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default='sparse_weighted_model')
    ap.add_argument('-d', '--dataFile', required=True)
    ap.add_argument('-n', '--N_samples', type=int, default=10)
    args = ap.parse_args()
    with open(args.dataFile, 'rb') as f:
        data = pickle.load(f)
    model = make_model(args.model, N=data['N'], dt=0.001)
    stabilize_sparsity(model)
    popn = Population(model)
    popn.add_data(data)
    rng = np.random.RandomState(0)
    # The reference initialises MCMC from a MAP fit converted to this model
    # (gibbs.py:2490-2507); a raw prior draw of W ~ N(0,1) can drive every quadrature node to
    # lam = 0 ("log_G not finie", gibbs.py:1024-1026).  Start from a prior draw with the weights
    # shrunk towards zero instead.
    x0 = popn.sample(rng)
    x0['net']['weights']['W'] = 0.05 * x0['net']['weights']['W']
    gibbs_network_sweeps(popn, x0, args.N_samples, rng)


if __name__ == '__main__':
    main()
