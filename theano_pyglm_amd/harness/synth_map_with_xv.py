"""
MAP fit with cross-validation over prior hyper-parameters -- counterpart of
test/synth_map_with_xv.py:15-104 and synth_harness.get_xv_models (synth_harness.py:61-119).

    python -m theano_pyglm_amd.harness.synth_map_with_xv -d data.pkl -r out_dir [-m standard_glm]

The data are split 75 % / 25 % in time (segment_data); every candidate model (the group-lasso
`lam` grid of the reference) is fitted on the training split starting from the best optimum so
far and scored by the held-out log likelihood.  Training split, held-out split and the full data
each get their own device-resident handle, uploaded once; `set_hyperparameters` only changes
host-side prior scalars, so nothing moves between candidates.

Note on the reference script: its `nlp` sums over `population.data_sequences`, which at that point
holds the *full* data set only (the splits are preprocessed but never added, :26-31), so the
reference trains every candidate on all data.  Here `data_sequences` is switched to the training
split while fitting -- the evident intent of the script.
"""
import argparse
import copy
import itertools
import os
import pickle
import time

import numpy as np

from theano_pyglm_amd.harness.synth_map import initialize_test_harness
from theano_pyglm_amd.inference.coord_descent import coord_descent
from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity
from theano_pyglm_amd.utils.io import segment_data, load_data

XV_GRID = {('impulse', 'prior', 'lam'): (0.5, 1.0, 2.0, 3.0, 5.0, 7.5, 10.0)}     # synth_harness.py:76


def _has_path(d, path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return False
        d = d[k]
    return True


def _set_path(d, path, v):
    for k in path[:-1]:
        d = d[k]
    d[path[-1]] = v


def get_xv_models(model, grid=None):
    """synth_harness.py:61-119: one deep copy of `model` per point of the Cartesian product of the
    hyper-parameter grid, restricted to the settings the model actually has."""
    grid = XV_GRID if grid is None else grid
    keys = [k for k in grid if _has_path(model, k)]
    models = []
    for combo in itertools.product(*[grid[k] for k in keys]):
        m = copy.deepcopy(model)
        for k, v in zip(keys, combo):
            _set_path(m, k, v)
        models.append(m)
    return models


def run_xv(popn, data, models, train_frac=0.75, batched=None, rng=None, verbose=True):
    """test/synth_map_with_xv.py:23-90.  Returns (best_x, best_ind, train_lps, xv_lls, total_lls)."""
    T_split = data['T'] * train_frac
    train_data = popn.preprocess_data(segment_data(data, (0, T_split)))
    xv_data = popn.preprocess_data(segment_data(data, (T_split, data['T'])))
    full_sequences = popn.data_sequences
    best_x = popn.sample(rng)
    best_ind, best_xv_ll, best_model = -1, -np.inf, None
    train_lps, xv_lls, total_lls = (np.zeros(len(models)) for _ in range(3))
    try:
        for i, model in enumerate(models):
            x0 = copy.deepcopy(best_x)
            popn.set_hyperparameters(model)
            popn.data_sequences = [train_data]
            popn.set_data(train_data)
            x_inf = coord_descent(popn, x0=x0, maxiter=1, batched=batched)
            train_lps[i] = popn.compute_log_p(x_inf)
            popn.set_data(xv_data)
            popn.data_sequences = [xv_data]
            xv_lls[i] = popn.compute_ll(x_inf)
            popn.set_data(data)
            popn.data_sequences = [data]
            total_lls[i] = popn.compute_ll(x_inf)
            if verbose:
                print("Model %d:\tTrain LP: %.1f\tXV LL: %.1f\tTotal LL: %.1f"
                      % (i, train_lps[i], xv_lls[i], total_lls[i]))
            if xv_lls[i] > best_xv_ll:
                best_ind, best_xv_ll = i, xv_lls[i]
                best_x, best_model = copy.deepcopy(x_inf), copy.deepcopy(model)
    finally:
        popn.data_sequences = full_sequences
        popn.set_data(data)
        popn.release_data(train_data)         # the split handles (and their resident features) die here
        popn.release_data(xv_data)
    # refit the winner on all data, warm-started from its training optimum (:81-88)
    if best_model is not None:
        popn.set_hyperparameters(best_model)
        best_x = coord_descent(popn, x0=copy.deepcopy(best_x), maxiter=1, batched=batched)
    return best_x, best_ind, train_lps, xv_lls, total_lls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default='standard_glm')
    ap.add_argument('-d', '--dataFile', required=True)
    ap.add_argument('-r', '--resultsDir', default='.')
    args = ap.parse_args()
    data = load_data(args.dataFile)
    popn, popn_true, x_true = initialize_test_harness(args.model, data, os.path.dirname(args.dataFile))
    base = make_model(args.model, N=data['N'], dt=0.001)
    stabilize_sparsity(base)
    t0 = time.time()
    best_x, best_ind, tr, xv, tot = run_xv(popn, data, get_xv_models(base))
    print("Best model: %d   (XV sweep wall-clock %.2f s)" % (best_ind, time.time() - t0))
    print("Best Total LL: %f" % popn.compute_ll(best_x))
    if popn_true is not None:
        print("True LL: %f" % popn_true.compute_ll(x_true))
    with open(os.path.join(args.resultsDir, 'results.pkl'), 'wb') as f:
        pickle.dump(best_x, f, protocol=-1)


if __name__ == '__main__':
    main()
