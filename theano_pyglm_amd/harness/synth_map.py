"""
MAP fit on synthetic data -- counterpart of test/synth_map.py + test/synth_harness.py.

    python -m theano_pyglm_amd.harness.synth_map -d data.pkl -r out_dir [-m standard_glm] [--sequential]

The sweep over the neurons runs as the GPU lock-step optimizer by default (inference/batched_bfgs.py);
--sequential (batched=False) is the reference's loop of per-neuron scipy fits.
"""
import argparse
import os
import pickle
import time

from theano_pyglm_amd.inference.coord_descent import coord_descent
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.population import Population


def initialize_test_harness(model_name, data, data_dir=None):
    """test/synth_harness.py:9-59."""
    model = make_model(model_name, N=data['N'], dt=0.001)
    stabilize_sparsity(model)
    popn = Population(model)
    popn.add_data(data)
    popn_true, x_true = None, None
    if 'vars' in data and data_dir is not None and os.path.exists(os.path.join(data_dir, 'model.pkl')):
        x_true = data['vars']
        with open(os.path.join(data_dir, 'model.pkl'), 'rb') as f:
            model_true = pickle.load(f)
        popn_true = Population(model_true)
        popn_true.add_data(data)
        print("true LL: %f" % popn_true.compute_log_p(x_true))
    return popn, popn_true, x_true


def run_synth_test(model_name, data, results_dir, data_dir=None, batched=None, rng=None):
    """test/synth_map.py:10-32."""
    popn, popn_true, x_true = initialize_test_harness(model_name, data, data_dir)
    x0 = popn.sample(rng)
    print("LL0: %f" % popn.compute_log_p(x0))
    t0 = time.time()
    x_inf = coord_descent(popn, x0=x0, maxiter=1, batched=batched)
    wall = time.time() - t0
    ll_inf = popn.compute_log_p(x_inf)
    print("LL_inf: %f   (MAP wall-clock %.2f s)" % (ll_inf, wall))
    if results_dir is not None:
        with open(os.path.join(results_dir, 'results.pkl'), 'wb') as f:
            pickle.dump(x_inf, f, protocol=-1)
    return x_inf, ll_inf, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default='standard_glm')
    ap.add_argument('-d', '--dataFile', required=True)
    ap.add_argument('-r', '--resultsDir', default='.')
    ap.add_argument('--sequential', action='store_true', help='per-neuron scipy fits (the reference loop)')
    ap.add_argument('--batched', action='store_true', help='(deprecated, no-op: the GPU lock-step sweep is the default)')
    args = ap.parse_args()
    with open(args.dataFile, 'rb') as f:
        data = pickle.load(f)
    run_synth_test(args.model, data, args.resultsDir, os.path.dirname(args.dataFile),
                   False if args.sequential else None)


if __name__ == '__main__':
    main()
