"""Data-dict I/O -- counterpart of pyglm/utils/io.py:82-149 (the schema of SURVEY §8a A0:
keys S (nT,N), N, dt, T, stim, dt_stim, optionally vars).  Command-line parsing and results
folders of the reference's io.py are outside the hot-path scope."""
import copy
import pickle

import numpy as np


def load_data(path):
    """io.py:82-124: '.pkl' -> the pickled data dict (files written by the Python-2 reference load
    with latin1 decoding); '.mat' -> scipy.io.loadmat(squeeze_me=True) with N / T converted back
    to python scalars (io.py:98-100)."""
    if path is None:
        raise Exception("Path to data file (.mat or .pkl) must be specified.")
    if path.endswith('.mat'):
        import scipy.io
        data = scipy.io.loadmat(path, squeeze_me=True)
        data['N'] = int(data['N'])
        data['T'] = float(data['T'])
        for k in ('dt', 'dt_stim'):
            if k in data:
                data[k] = float(data[k])
        if 'stim' in data and np.ndim(data['stim']) == 1:
            data['stim'] = np.reshape(data['stim'], (-1, 1))
        return data
    if path.endswith('.pkl'):
        with open(path, 'rb') as f:
            try:
                return pickle.load(f)
            except UnicodeDecodeError:
                f.seek(0)
                return pickle.load(f, encoding='latin1')
    raise Exception("Unrecognized file type: %s" % path)


def save_data(data, path):
    """generate_synth_data.py:131-140: pickle (protocol 2, readable by the reference) or .mat.
    Device handles and cached features attached by Population.add_data are not written."""
    out = {k: v for k, v in data.items() if k not in ('_device_handle', 'fS', 'fstim', 'preprocessed')}
    if path.endswith('.mat'):
        import scipy.io
        scipy.io.savemat(path, {k: v for k, v in out.items() if k != 'vars'}, oned_as='row')
    elif path.endswith('.pkl'):
        with open(path, 'wb') as f:
            pickle.dump(out, f, protocol=2)
    else:
        raise Exception("Unrecognized file type: %s" % path)


def segment_data(data, T_range):
    """io.py:126-149: the sub-interval [T_start, T_stop) seconds of a data dict (spikes cut on the
    dt grid, stimulus on the dt_stim grid).  The copy is not 'preprocessed' and owns no device
    handle, so adding it to a Population uploads the segment afresh."""
    T_start, T_stop = T_range
    assert 0 <= T_start <= data['T'] and 0 <= T_stop <= data['T'] and T_start < T_stop
    new = {k: copy.deepcopy(v) for k, v in data.items()
           if k not in ('_device_handle', 'fS', 'fstim', 'preprocessed')}
    new['T'] = T_stop - T_start
    i0, i1 = int(T_start // data['dt']), int(T_stop // data['dt'])
    new['S'] = new['S'][i0:i1, :]
    if 'stim' in data and data['stim'] is not None:
        j0, j1 = int(T_start // data['dt_stim']), int(T_stop // data['dt_stim'])
        new['stim'] = new['stim'][j0:j1, :]
    return new
