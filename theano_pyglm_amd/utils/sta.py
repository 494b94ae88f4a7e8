"""Spike-triggered average -- counterpart of pyglm/utils/sta.py:6-85, computed on the device
from the resident spike event lists (pgl_sta) instead of a dense (nT, L*D) lag matrix."""
import numpy as np

from theano_pyglm_amd import _lib


def sta(stim, data, L, Ns=None, handle=None, keep_on_device=False):
    """A[i,l,:] = sum_t S[t,Ns[i]] * istim[t-l,:] / sum_t S[t,Ns[i]]  (sta.py:43-80).

    stim : (Tstim, D) stimulus at sampling interval data['dt_stim']
    data : dict with 'S' (nT,N), 'dt', 'dt_stim'; `handle`: the resident DeviceGlm of a data set
           already added to a Population (Population._handle(data)); without it the spikes are
           uploaded for this call
    L    : number of lags in bins of data['dt'];  Ns: neuron indices (default all, int allowed)
    keep_on_device (needs `handle`): the averages stay on the device for handle.leading_singular_pairs(None, shape);
           returns their shape
    """
    stim = np.asarray(stim, dtype=float)
    if stim.ndim != 2:
        raise ValueError("stim must be (Tstim, D)")
    S = np.asarray(data['S'])
    nT, N = S.shape
    if Ns is None:
        Ns = np.arange(N)
    if isinstance(Ns, (int, np.integer)):
        Ns = [int(Ns)]
    h = handle
    own = h is None
    if own:
        h = _lib.DeviceGlm(N, nT, 1, 1, 'exp', float(data['dt']))
        h.set_spikes(S)
    try:
        return h.sta(stim, float(data['dt_stim']), int(L), Ns=Ns, keep_on_device=keep_on_device and not own)
    finally:
        if own:
            h.close()
