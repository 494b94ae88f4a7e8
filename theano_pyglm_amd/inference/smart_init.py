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


def leading_singular_pair(A):
    """(u_0, sigma_0, v_0) of A -- all the reference uses of np.linalg.svd(sn) (smart_init.py:68-72).  From the leading
    eigenvector of the smaller Gram matrix, polished by two steps of the alternating iteration on A itself: a full SVD of a
    300 x 1024 STA builds a 1024 x 1024 factor (0.3 s per neuron), the thin one still takes 0.17 s -- 11 s beside a 0.2 s MAP
    sweep at 64 neurons; this takes 17 ms on the GPU box's host (leading_singular_pairs: all neurons at once on the GPU).  The sign of the pair is LAPACK's business in the reference; here the component
    of u_0 of largest magnitude is positive (the rank-1 filter u_0 v_0^T does not depend on it)."""
    A = np.asarray(A, dtype=float)
    tall = A.shape[0] > A.shape[1]
    G = A.T.dot(A) if tall else A.dot(A.T)
    _, Q = np.linalg.eigh(G)
    x = Q[:, -1]
    for _ in range(2):
        y = (A.dot(x) if tall else A.T.dot(x))
        ny = np.linalg.norm(y)
        if ny == 0.0:
            break
        y /= ny
        x = (A.T.dot(y) if tall else A.dot(y))
        nx = np.linalg.norm(x)
        if nx == 0.0:
            break
        x /= nx
    u, v = ((A.dot(x), x) if tall else (x, A.T.dot(x)))
    sig = np.linalg.norm(u) if tall else np.linalg.norm(v)
    if sig > 0.0:
        if tall:
            u = u / sig
        else:
            v = v / sig
    sgn = np.sign(u[np.argmax(np.abs(u))]) or 1.0
    return u * sgn, float(sig), v * sgn


def leading_singular_pairs(S, device=None, handle=None):
    """leading_singular_pair for a batch S (n, L, D).  With a HIP device: pgl_leading_singular_pairs -- the library's own
    kernels (batched Gram matrices of the smaller side on the f64 MFMA, repeated squaring, two alternating steps on the
    matrices themselves; 64 STAs of 300 x 1024 in a few ms against 64 x 17 ms on the host) on `handle` (a DeviceGlm;
    None: a bare handle on `device` for the duration of the call).  Without the library or a device: the host routine,
    matrix by matrix.  Same sign convention.  Returns (U (n, L), sigma (n,), V (n, D))."""
    S = np.asarray(S, dtype=float)
    from theano_pyglm_amd import _lib
    use_gpu = S.ndim == 3 and min(S.shape[1:]) > 1
    if use_gpu and handle is None:
        try:
            use_gpu = _lib.device_count() > 0
        except Exception:
            use_gpu = False
    if not use_gpu:
        out = [leading_singular_pair(a) for a in S]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out])
    if handle is not None:
        return handle.leading_singular_pairs(S)
    tmp = _lib.DeviceGlm(1, 16, 1, 1, 'exp', 0.001, 0 if device is None else int(device))
    try:
        return tmp.leading_singular_pairs(S)
    finally:
        tmp.close()


def stim_weights_from_sta(bkgd, sn, pair=None):
    """One neuron's stimulus weights from its (L, D) STA (smart_init.py:66-98).
    Spatiotemporal: best rank-1 factor pair f_t f_x^T of the STA (leading singular pair, each scaled
    by sqrt(sigma_0)), projected onto the temporal / spatial bases.  Basis: every stimulus dimension
    projected onto the temporal basis, stacked d-major."""
    if sn is not None:
        sn = np.asarray(sn, dtype=float)
        if sn.ndim == 1:
            sn = sn.reshape(-1, 1)
    if isinstance(bkgd, SpatiotemporalStimulus):
        u, sig, v = leading_singular_pair(sn) if pair is None else pair
        f_t = u * np.sqrt(sig)
        f_x = v * np.sqrt(sig)
        # (identity spatial basis: the projection is f_x itself)
        return {'w_x': f_x.copy() if getattr(bkgd, 'identity_x', False) else np.ravel(project_onto_basis(f_x, bkgd.ibasis_x)),
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
    handle = population._find_handle(data)
    if isinstance(bkgd, SpatiotemporalStimulus) and np.ndim(data['stim']) == 2 and len(Ns) > 1 and handle is not None:
        # averages and their rank-1 factors stay on the device: only the factors (n x (L + D) numbers) come back
        shape = sta(data['stim'], data, L, Ns=Ns, handle=handle, keep_on_device=True)
        U, Sig, V = handle.leading_singular_pairs(None, shape)
        for i, n in enumerate(Ns):
            x0['glms'][n]['bkgd'].update(stim_weights_from_sta(bkgd, None, (U[i], float(Sig[i]), V[i])))
        return
    s = sta(data['stim'], data, L, Ns=Ns, handle=handle)
    pairs = None
    if isinstance(bkgd, SpatiotemporalStimulus) and np.ndim(s) == 3 and len(Ns) > 1:
        U, Sig, V = leading_singular_pairs(s, device=getattr(population, 'device', None), handle=handle)
        pairs = [(U[i], float(Sig[i]), V[i]) for i in range(len(Ns))]
    for i, n in enumerate(Ns):
        x0['glms'][n]['bkgd'].update(stim_weights_from_sta(bkgd, s[i], None if pairs is None else pairs[i]))
