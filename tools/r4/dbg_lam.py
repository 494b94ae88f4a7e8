import sys, time
import numpy as np
sys.path.insert(0, '.')
from theano_pyglm_amd.models import templates
from theano_pyglm_amd.harness.generate_synth_data import make_dataset
N = int(sys.argv[1]); T = float(sys.argv[2]); D = int(sys.argv[3])
tmpl = templates.spatiotemporal_glm()
tmpl['bkgd']['D_stim'] = D
tmpl['bkgd']['spatial_basis'] = {'type': 'identity', 'n_eye': D}
def tame(x):
    for g in x['glms']:
        g['bkgd']['w_x'] = np.asarray(g['bkgd']['w_x']) * (0.4 / np.sqrt(D))
        g['bkgd']['w_t'] = np.asarray(g['bkgd']['w_t']) * 0.5
        g['bias']['bias'] = 2.5 + 0.3 * np.asarray(g['bias']['bias'])
        g['imp']['w_ir'] = np.asarray(g['imp']['w_ir']) * 0.1
model, popn, data = make_dataset(tmpl, N, T, seed=1234 + 5, adjust=tame, check=False)
print("rates Hz:", np.round(data['S'].sum(0) / T, 1)[:8], "max count", data['S'].max(), flush=True)
popn.add_data(data)
h = popn._handle(popn._current)
print("stim_path", h.info()['stim_path'], "sep", popn.glm.bkgd_model.separable)
state = popn.eval_state(data['vars'])
bk = popn.glm.bkgd_model
fst = bk.host_features(data, data['S'].shape[0])
for n in (0, 1, N - 1):
    lam_true = state['glms'][n]['lam']
    lam_sim = popn.glm.nlin_model.f_nlin(data['X'][:, n])
    x_sim = data['X'][:, n]
    i_host = fst.dot(bk.dense_weights(data['vars']['glms'][n]['bkgd']))
    i_dev = state['glms'][n]['I_bkgd']
    bad = np.argmax(np.abs(i_host - i_dev))
    print(n, "lam rel err max", np.max(np.abs(lam_true - lam_sim) / np.maximum(np.abs(lam_sim), 1e-300)),
          "| I_stim host vs device max abs", np.max(np.abs(i_host - i_dev)), "at bin", bad, "of", len(i_host),
          "| max|I_stim|", np.abs(i_host).max(), "| x range", x_sim.min(), x_sim.max())
from oracle import c_oracle as CO
S8 = data['S'].astype(np.uint8)
fS = CO.features(S8, popn.glm.imp_model.ibasis)
for n in (0, 1):
    w = popn.glm.imp_model.flat_weights(data['vars']['glms'][n]['imp']).reshape(N, -1)
    I_or = np.einsum('tkb,kb->t', fS, w)
    I_dev = state['glms'][n]['I_net']
    I_sim = data['X'][:, n] - data['vars']['glms'][n]['bias']['bias'][0] - fst.dot(bk.dense_weights(data['vars']['glms'][n]['bkgd']))
    d1, d2 = np.abs(I_dev - I_or), np.abs(I_sim - I_or)
    print(n, "I_net device vs oracle max", d1.max(), "first bad bins", np.nonzero(d1 > 1e-9)[0][:5], "| simulate vs oracle max", d2.max(), "first bad", np.nonzero(d2 > 1e-9)[0][:5])
    b = np.nonzero(d2 > 1e-9)[0]
    if len(b):
        t = b[0]
        print("   around bin", t, "counts of all neurons in [t-3, t]:", data['S'][t - 3:t + 1].sum(axis=1), "max count so far", data['S'][:t + 1].max())
