"""
Synthetic data generation -- counterpart of test/generate_synth_data.py.

    python -m theano_pyglm_amd.harness.generate_synth_data -m standard_glm -N 4 -T 60 -r out_dir

Samples parameters from the model's prior, simulates spikes with Population.simulate
(seeded), checks the reference's consistency invariant
    allclose(state['glms'][n]['lam'], f_nlin(X[:,n]))      test/generate_synth_data.py:125-129
through the device path, and pickles data.pkl / model.pkl with the reference's schema
(S, X, N, dt, T, stim, dt_stim, vars).
"""
import argparse
import os
import pickle

import numpy as np

from theano_pyglm_amd.models.model_factory import make_model, stabilize_sparsity, check_stability
from theano_pyglm_amd.population import Population


def gen_synth_data(N, T_stop, popn, x_true, dt=0.001, dt_stim=0.1, stim=None, rng=None):
    """test/generate_synth_data.py:56-79."""
    S, X = popn.simulate(x_true, (0, T_stop), dt, stim, dt_stim, rng=rng)
    return {"S": S, "X": X, "N": N, "dt": dt, "T": float(T_stop), "stim": stim,
            'dt_stim': dt_stim, 'vars': x_true}


def make_dataset(model_name, N, T_stop, seed=0, dt=0.001, dt_stim=0.1, check=True, device=None,
                 adjust=None):
    rng = np.random.RandomState(seed)
    model = make_model(model_name, N=N, dt=dt)
    stabilize_sparsity(model)
    popn = Population(model, device=device)
    x_true = popn.sample(rng)
    if adjust is not None:
        adjust(x_true)        # e.g. tame prior draws that would saturate the 10-spikes-per-bin cap
    assert check_stability(model, x_true, N), "ERROR: Sampled network is unstable!"
    D = model['bkgd'].get('D_stim', 1)
    stim = rng.randn(int(round(T_stop / dt_stim)), D)
    data = gen_synth_data(N, T_stop, popn, x_true, dt, dt_stim, stim, rng)
    if check:
        popn.add_data(data)
        state = popn.eval_state(x_true)
        for n in range(N):
            lam_true = state['glms'][n]['lam']
            lam_sim = popn.glm.nlin_model.f_nlin(data['X'][:, n])
            assert np.allclose(lam_true, lam_sim), "rate consistency check failed for neuron %d" % n
    return model, popn, data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default='standard_glm')
    ap.add_argument('-r', '--resultsDir', default='.')
    ap.add_argument('-N', '--N', type=int, default=1)
    ap.add_argument('-T', '--T_stop', type=float, default=60.0)
    ap.add_argument('-s', '--seed', type=int, default=0)
    args = ap.parse_args()
    model, popn, data = make_dataset(args.model, args.N, args.T_stop, args.seed)
    out = dict((k, v) for k, v in data.items() if not k.startswith('_') and k not in ('fstim', 'preprocessed'))
    with open(os.path.join(args.resultsDir, 'model.pkl'), 'wb') as f:
        pickle.dump(model, f, protocol=-1)
    with open(os.path.join(args.resultsDir, 'data.pkl'), 'wb') as f:
        pickle.dump(out, f, protocol=-1)
    print("Sampled %s spikes." % str(np.sum(data['S'], 0)))


if __name__ == '__main__':
    main()
