"""
MCMC on synthetic data -- counterpart of test/synth_mcmc.py (non-interactive: the reference
prompts with raw_input and calls gibbs_sample with a stale signature).

    python -m theano_pyglm_amd.harness.synth_mcmc -d data.pkl [-m sparse_weighted_model] [-n 10]
                                                  [-r out_dir] [--network-only]

Full sweeps (gibbs.py:2548-2556): HMC on biases / stimulus weights / impulse parameters in lock
step over neurons, then the collapsed Gibbs update of every network column.  With x0 absent the
chain starts from a standard_glm MAP fit converted to the model (gibbs.py:2490-2507).
--network-only keeps everything but A, W fixed (the "synth_mcmc inner ll" path alone).
"""
import argparse
import os
import pickle
import time

import numpy as np

from theano_pyglm_amd.inference.gibbs import CollapsedGibbsNetworkColumnUpdate, gibbs_sample
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population
from theano_pyglm_amd.utils.io import load_data


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
    ap.add_argument('-r', '--resultsDir', default=None)
    ap.add_argument('-n', '--N_samples', type=int, default=10)
    ap.add_argument('--network-only', action='store_true')
    args = ap.parse_args()
    data = load_data(args.dataFile)
    model = make_model(args.model, N=data['N'], dt=0.001)
    stabilize_sparsity(model)
    popn = Population(model)
    popn.add_data(data)
    rng = np.random.RandomState(0)
    if args.network_only:
        # A raw prior draw of W ~ N(0,1) can drive every quadrature node to lam = 0
        # ("log_G not finie", gibbs.py:1024-1026): start with the weights shrunk towards zero.
        x0 = popn.sample(rng)
        x0['net']['weights']['W'] = 0.05 * x0['net']['weights']['W']
        gibbs_network_sweeps(popn, x0, args.N_samples, rng)
        return
    smpls = gibbs_sample(popn, N_samples=args.N_samples, x0=None, init_from_mle=True, rng=rng)
    if args.resultsDir is not None:
        with open(os.path.join(args.resultsDir, 'results.pkl'), 'wb') as f:      # synth_mcmc.py:97-101
            pickle.dump(smpls, f, protocol=-1)


if __name__ == '__main__':
    main()
