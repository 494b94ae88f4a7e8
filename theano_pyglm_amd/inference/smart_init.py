"""Data-driven initialisation -- counterpart of pyglm/inference/smart_init.py:7-98.

The dense-graph initialisation is a host-side one-liner; the stimulus weights are warm-started
from the spike-triggered average, which runs on the device (utils/sta.py -> pgl_sta), followed by
the reference's tiny host-side factorisation / basis projection."""
import numpy as np

from theano_pyglm_amd.components.bkgd import BasisStimulus, SpatiotemporalStimulus
from theano_pyglm_amd.utils.basis import project_onto_basis
from theano_pyglm_amd.utils.sta import sta


def initialize_with_data(population, data, x0, Ns=None):
    """smart_init.py:7-12."""
    initialize_stim_with_sta(population, data, x0, Ns=Ns)
    initialize_with_dense_graph(population, data, x0)


def initialize_with_dense_graph(population, data, x0):
    """smart_init.py:14-18."""
    if 'A' in x0['net']['graph']:
        x0['net']['graph']['A'] = np.ones_like(x0['net']['graph']['A'])


def initialize_with_no_coupling(population, data, x0):
    """smart_init.py:20-26."""
    for glm in x0['glms']:
        if 'w_ir' in glm['imp']:
            glm['imp']['w_ir'] = np.zeros_like(glm['imp']['w_ir'])
        if 'bias' in glm['bias']:
            glm['bias']['bias'] = 0


def stim_weights_from_sta(bkgd, sn):
    """One neuron's stimulus weights from its (L, D) STA (smart_init.py:66-98).
    Spatiotemporal: best rank-1 factor pair f_t f_x^T of the STA (leading singular pair, each scaled
    by sqrt(sigma_0)), projected onto the temporal / spatial bases.  Basis: every stimulus dimension
    projected onto the temporal basis, stacked d-major."""
    sn = np.asarray(sn, dtype=float)
    if sn.ndim == 1:
        sn = sn.reshape(-1, 1)
    if isinstance(bkgd, SpatiotemporalStimulus):
        # (thin SVD: the leading pair is all that is used; the full one builds a D x D factor -- 0.3 s per neuron at D = 1024)
        U, Sig, Vt = np.linalg.svd(sn, full_matrices=False)
        f_t = U[:, 0] * np.sqrt(Sig[0])
        f_x = Vt[0, :] * np.sqrt(Sig[0])
        return {'w_x': np.ravel(project_onto_basis(f_x, bkgd.ibasis_x)),
                'w_t': np.ravel(project_onto_basis(f_t, bkgd.ibasis_t))}
    if isinstance(bkgd, BasisStimulus):
        w = [np.ravel(project_onto_basis(sn[:, d], bkgd.ibasis)) for d in range(sn.shape[1])]
        return {'w_stim': np.concatenate(w)}
    return {}


def initialize_stim_with_sta(population, data, x0, Ns=None):
    """smart_init.py:28-98; a no-op for models without a basis-function stimulus (:42-43)."""
    bkgd = population.glm.bkgd_model
    if isinstance(bkgd, BasisStimulus):
        L = bkgd.ibasis.shape[0]
    elif isinstance(bkgd, SpatiotemporalStimulus):
        L = bkgd.ibasis_t.shape[0]
    else:
        return
    if Ns is None:
        Ns = np.arange(population.N)
    if isinstance(Ns, (int, np.integer)):
        Ns = [int(Ns)]
    s = sta(data['stim'], data, L, Ns=Ns, handle=population._find_handle(data))
    for i, n in enumerate(Ns):
        x0['glms'][n]['bkgd'].update(stim_weights_from_sta(bkgd, s[i]))
