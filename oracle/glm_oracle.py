"""
CPU oracle for the population-GLM log-likelihood / gradient hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``theano_pyglm_amd``) never imports anything from
``oracle/`` and fails loudly when the HIP library is missing.

It is a plain numpy/float64 restatement of the arithmetic of
slinderman/theano_pyglm (reference mounted at /root/reference while this was
written; nothing here reads it at run time).  Every function cites the
reference file:line it follows.

Parity pinning:
  * basis construction and the two causal convolutions are pinned against
    golden vectors produced by the reference's own ``pyglm/utils/basis.py``
    (tests/golden/make_golden.py -> tests/golden/*.npz);
  * the Theano graph (glm.py:31-63 and the component expressions) cannot be
    executed here (no Python 2 / Theano) and the reference ships no golden
    values for it.  For that part the status is "parity unpinned against
    reference output": it is pinned instead by the reference's own
    invariant ``allclose(lam_true, lam_sim)`` (test/generate_synth_data.py:125-129)
    re-stated in ``direct_currents`` below, by torch-float64 autograd and by
    central finite differences (tests/test_oracle.py).
"""
import numpy as np

# ----------------------------------------------------------------------------
# A1. bases (pyglm/utils/basis.py:9-199) and their interpolation
# ----------------------------------------------------------------------------

def create_cosine_basis(prms, orth_fn=None):
    """pyglm/utils/basis.py:56-106.  100-point raised cosines in log time.

    ``orth_fn`` lets a caller inject scipy.linalg.orth; the orthonormalised
    basis is SVD-sign/platform dependent so product
    code uses committed tables instead of recomputing."""
    n_pts = 100
    n_cos = prms['n_cos']
    n_eye = prms['n_eye']
    n_bas = n_eye + n_cos
    basis = np.zeros((n_pts, n_bas))
    basis[:n_eye, :n_eye] = np.eye(n_eye)
    a = prms['a']
    b = prms['b']
    u_ir = np.log(a * np.arange(n_pts) + b)                              # basis.py:82-83
    ctrs = u_ir[np.floor(np.linspace(n_eye, n_pts / 2.0, n_cos)).astype(int)]  # :84
    if len(ctrs) == 1:
        w = ctrs / 2
    else:
        w = (ctrs[-1] - ctrs[0]) / (n_cos - 1)                           # :88
    for i in range(n_cos):                                               # :91-93
        arg = np.maximum(-np.pi, np.minimum(np.pi, (u_ir - ctrs[i]) * np.pi / w / 2.0))
        basis[:, n_eye + i] = (np.cos(arg) + 1) / 2.0
    if prms['orth']:                                                     # :97-98
        if orth_fn is None:
            import scipy.linalg
            orth_fn = scipy.linalg.orth
        basis = orth_fn(basis)
    if prms['norm']:                                                     # :99-104
        if np.any(basis < 0):
            raise Exception("We can only normalize nonnegative impulse responses!")
        basis = basis / np.tile(np.sum(basis, axis=0), [n_pts, 1]) / (1.0 / n_pts)
    return basis


def create_identity_basis(prms):
    """pyglm/utils/basis.py:187-199."""
    return np.eye(prms['n_eye'])


def create_basis(prms, orth_fn=None):
    """pyglm/utils/basis.py:9-26 (cosine and identity only; the other types
    are not used by the named configs)."""
    typ = prms['type'].lower()
    if typ == 'cosine':
        return create_cosine_basis(prms, orth_fn)
    if typ in ('identity', 'eye'):
        return create_identity_basis(prms)
    raise Exception("Unrecognized basis type: %s" % typ)


def interp_basis_unit(basis, n_int):
    """The interpolation shared by LinearBasisImpulses.initialize_basis
    (impulse.py:92-103), BasisStimulus (bkgd.py:101-112) and the temporal part
    of SpatiotemporalStimulus (bkgd.py:274-284): both grids are linspace(0,1,.)."""
    L, B = basis.shape
    t_int = np.linspace(0, 1, n_int)
    t_bas = np.linspace(0, 1, L)
    ib = np.zeros((n_int, B))
    for b in range(B):
        ib[:, b] = np.interp(t_int, t_bas, basis[:, b])
    return ib


def linear_impulse_ibasis(basis, dt, dt_max, norm):
    """impulse.py:92-112: interpolate to R=dt_max/dt taps; '/dt_max' if norm."""
    R = int(round(dt_max / dt))
    ib = interp_basis_unit(basis, R)
    if norm:
        ib = ib / dt_max
    return ib


def dirichlet_impulse_ibasis(basis, dt, dt_max, norm):
    """impulse.py:359-376: t_int=arange(0,dt_max,dt), t_bas=linspace(0,dt_max,L),
    '/trapz' if norm."""
    L, B = basis.shape
    t_int = np.arange(0.0, dt_max, step=dt)
    t_bas = np.linspace(0.0, dt_max, L)
    ib = np.zeros((len(t_int), B))
    for b in range(B):
        ib[:, b] = np.interp(t_int, t_bas, basis[:, b])
    if norm:
        _trapz = getattr(np, "trapezoid", None) or np.trapz
        ib = ib / _trapz(ib, t_int, axis=0)
    return ib


def stim_temporal_ibasis(basis, dt, dt_max, norm):
    """bkgd.py:101-118 and 274-296: interpolate, then unit L1 norm per column."""
    R = int(round(dt_max / dt))
    ib = interp_basis_unit(basis, R)
    if norm:
        ib = ib / np.tile(np.sum(ib, 0), [R, 1])
    return ib


def stim_spatial_ibasis(basis, D_stim):
    """bkgd.py:286-292 (no normalisation of the spatial basis)."""
    return interp_basis_unit(basis, D_stim)


# ----------------------------------------------------------------------------
# A2/A3. causal convolutions (pyglm/utils/basis.py:201-273)
# ----------------------------------------------------------------------------

def convolve_with_basis(stim, basis):
    """basis.py:201-236.  fstim[t,d,b] = sum_{tau=1..R} stim[t-tau,d]*basis[tau-1,b]
    (a zero row is prepended to the basis, 217-220, so the filter is strictly
    causal; 'full' convolution truncated to the first T rows, 232-234).
    Direct time-domain form (the reference uses an FFT; same numbers to ~1e-15)."""
    T, D = stim.shape
    R, B = basis.shape
    out = np.zeros((T, D, B))
    for b in range(B):
        k = np.concatenate(([0.0], basis[:, b]))
        for d in range(D):
            out[:, d, b] = np.convolve(stim[:, d], k)[:T]
    return out


def convolve_with_basis_fft(stim, basis):
    """Same as convolve_with_basis via scipy's FFT convolution (what the
    reference calls, basis.py:232); used for large cases."""
    import scipy.signal as sig
    T, D = stim.shape
    R, B = basis.shape
    bz = np.vstack((np.zeros((1, B)), basis))
    out = np.empty((T, D, B))
    for b in range(B):
        out[:, :, b] = sig.fftconvolve(stim, bz[:, b].reshape(R + 1, 1), 'full')[:T, :]
    return out


def convolve_with_low_rank_2d_basis(stim, basis_x, basis_t):
    """basis.py:238-273: fstimx = stim @ basis_x (256), then the causal
    convolution of every column with every temporal basis -> (T,Bx,Bt)."""
    T, D = stim.shape
    Rx, Bx = basis_x.shape
    assert Rx == D
    fx = np.dot(stim, basis_x)
    return convolve_with_basis(fx, basis_t)          # (T,Bx,Bt), same convention


def interp_stim(stim, dt_stim, dt, nT):
    """bkgd.py:132-142 / 303-313: linear interpolation of each stimulus
    column from the dt_stim grid to the dt grid (np.interp clamps at the ends)."""
    t = dt * np.arange(nT)
    t_stim = dt_stim * np.arange(stim.shape[0])
    out = np.zeros((nT, stim.shape[1]))
    for d in range(stim.shape[1]):
        out[:, d] = np.interp(t, t_stim, stim[:, d])
    return out


def basis_stim_features(stim, dt_stim, dt, nT, ibasis):
    """BasisStimulus.preprocess_data, bkgd.py:122-154: (nT, D*B), column d*B+b."""
    s = interp_stim(stim, dt_stim, dt, nT)
    c = convolve_with_basis(s, ibasis)               # (nT,D,B)
    nT_, D, B = c.shape
    return c.reshape(nT_, D * B)


def spatiotemporal_stim_features(stim, dt_stim, dt, nT, ibasis_x, ibasis_t):
    """SpatiotemporalStimulus.preprocess_data, bkgd.py:303-340: (nT, Bt*Bx),
    column bt*Bx+bx (transpose to (T,Bt,Bx) then row-major reshape, 337-340)."""
    s = interp_stim(stim, dt_stim, dt, nT)
    f = convolve_with_low_rank_2d_basis(s, ibasis_x, ibasis_t)   # (nT,Bx,Bt)
    f = np.transpose(f, axes=[0, 2, 1])
    return f.reshape(nT, -1)


def frame_rate_table(ibasis_t, q):
    """The identity behind the device's frame-rate stimulus kernels (k_sepf_*), restated in numpy so that it can be
    checked on the CPU against the reference's own convolution: with the stimulus interpolated linearly between
    frames q bins apart (bkgd.py:303-313) and filtered causally with Rt taps (basis.py:238-273, 201-236), bin
    t = q F + o sees only the J = ceil(Rt / q) + 2 frame values base(F) .. base(F) + J - 1, base(F) = max(F - M, 0),
    M = ceil(Rt / q):
        conv(interp(z), ibasis_t)[t, bt] = sum_j C[row(t), j, bt] z[base(F) + j]
    with row(t) = t for t < q M (bins t - tau < 0 dropped), q M + o after.  Returns (C, M, J)."""
    Rt, Bt = ibasis_t.shape
    M = -(-Rt // q)
    J = M + 2
    C = np.zeros((q * (M + 1), J, Bt))
    for t in range(q * (M + 1)):
        base = max(t // q - M, 0)
        for tau in range(1, min(Rt, t) + 1):
            s = t - tau
            f, a = s // q, (s % q) / float(q)
            C[t, f - base] += ibasis_t[tau - 1] * (1.0 - a)
            if a != 0.0:
                C[t, f + 1 - base] += ibasis_t[tau - 1] * a
    return C, M, J


def frame_rate_features(z, ibasis_t, q, nT):
    """conv(interp(z), ibasis_t) (nT, Bx, Bt) through frame_rate_table; frames past the last one repeat it
    (np.interp holds the last value)."""
    C, M, J = frame_rate_table(ibasis_t, q)
    Tz = z.shape[0]
    out = np.zeros((nT,) + z.shape[1:] + (ibasis_t.shape[1],))
    for t in range(nT):
        F = t // q
        base = max(F - M, 0)
        row = t if t < q * M else q * M + t % q
        idx = np.minimum(base + np.arange(J), Tz - 1)
        out[t] = np.einsum('jb,j...->...b', C[row], z[idx])
    return out


def project_onto_basis(f, basis, lam=0.0):
    """basis.py:416-436: beta = inv(basis^T basis + lam I) basis^T f, shape (B,1)."""
    R, B = basis.shape
    f = np.asarray(f, dtype=float).reshape(R, -1)
    return np.dot(np.dot(np.linalg.inv(np.dot(basis.T, basis) + lam * np.eye(B)), basis.T), f)


def sta(stim, S, dt, dt_stim, L, Ns):
    """pyglm/utils/sta.py:6-85 in its dense form: interpolate the stimulus to the bin grid and
    divide by dt_stim/dt (27-41), pad L zero rows in front (44-45), build the lag matrix
    stim_lag[t, l*D+d] = istim_padded[L+t-l, d] (62-66), A[i] = S[:,n]^T stim_lag reshaped (L,D)
    (68-76), divided by the neuron's spike count (79-80)."""
    nT = S.shape[0]
    D = stim.shape[1]
    istim = interp_stim(stim, dt_stim, dt, nT) / (dt_stim / dt)
    pad = np.vstack((np.zeros((L, D)), istim))
    A = np.zeros((len(Ns), L, D))
    for l in range(L):
        lagged = pad[L - l:L + nT - l, :]                   # (nT, D): istim[t-l], zero for t<l
        for i, n in enumerate(Ns):
            A[i, l, :] = np.dot(S[:, n], lagged)
    with np.errstate(invalid='ignore', divide='ignore'):
        for i, n in enumerate(Ns):
            A[i] /= np.sum(S[:, n])
    return A


def sta_stim_weights(sn, kind, ibasis_t, ibasis_x=None):
    """smart_init.py:66-98: 'spatiotemporal' -> leading singular pair of the (L,D) STA scaled by
    sqrt(sigma_0), projected on the temporal/spatial bases; 'basis' -> per-dimension projection
    on the temporal basis, stacked d*B+b."""
    sn = np.asarray(sn, dtype=float)
    if sn.ndim == 1:
        sn = sn.reshape(-1, 1)
    if kind == 'spatiotemporal':
        U, Sig, V = np.linalg.svd(sn)
        f_t = U[:, 0] * np.sqrt(Sig[0])
        f_x = V[0, :] * np.sqrt(Sig[0])
        return {'w_t': np.ravel(project_onto_basis(f_t, ibasis_t)),
                'w_x': np.ravel(project_onto_basis(f_x, ibasis_x))}
    B = ibasis_t.shape[1]
    w = np.zeros(B * sn.shape[1])
    for d in range(sn.shape[1]):
        w[d * B:(d + 1) * B] = np.ravel(project_onto_basis(sn[:, d], ibasis_t))
    return {'w_stim': w}


# ----------------------------------------------------------------------------
# A5. the GLM log likelihood (glm.py:31-52 and the component expressions)
# ----------------------------------------------------------------------------

def softplus(x):
    """nlin.py:43 `log(1+exp(x))` in the overflow-safe form Theano rewrites it
    to.  Parity is defined on finite inputs."""
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


def nlin(x, kind):
    """nlin.py:25 (exp) / nlin.py:43 (explinear)."""
    if kind == 'exp':
        return np.exp(x)
    if kind == 'explinear':
        return softplus(x)
    raise Exception("Unrecognized nonlinearity model: %s" % kind)


def dirichlet_beta(g):
    """impulse.py:286-291: beta = |g| / sum|g| for one presynaptic neuron."""
    ga = np.abs(g)
    return ga / np.sum(ga)


def impulse_currents(fS, w):
    """impulse.py:58 / 308: I_imp[t,n'] = sum_b fS[t,n',b] * w[n',b]."""
    return np.sum(fS * w[None, :, :], axis=2)


def glm_currents(n, fS, w_imp, W_eff_col, bias, fstim=None, w_stim=None):
    """glm.py:31-45.  Returns (x, I_net, I_stim) with
    x = I_bias + I_stim + I_net,  I_net = I_imp . (A[:,n]*W[:,n])  (glm.py:33-39)."""
    I_imp = impulse_currents(fS, w_imp)
    I_net = I_imp.dot(W_eff_col)
    if fstim is not None:
        I_stim = fstim.dot(w_stim)                   # bkgd.py:81 / 227
    else:
        I_stim = 0.0                                 # bkgd.py:43
    x = bias + I_stim + I_net
    return x, I_net, I_stim


def glm_ll_from_x(x, Sn, dt, kind):
    """glm.py:52: ll = sum(-dt*lam + log(lam)*S[:,n])  (no log S!, no dt in the log)."""
    lam = nlin(x, kind)
    if kind == 'exp':
        loglam = x
    else:
        loglam = np.log(lam)
    return np.sum(-dt * lam + loglam * Sn)


def glm_ll(n, S, fS, w_imp, W_eff_col, bias, dt, kind, fstim=None, w_stim=None):
    x, _, _ = glm_currents(n, fS, w_imp, W_eff_col, bias, fstim, w_stim)
    return glm_ll_from_x(x, S[:, n], dt, kind)


def glm_resid(x, Sn, dt, kind):
    """r_t = d ll / d x_t: explinear (-dt + S/lam)*sigmoid(x);
    exp: -dt*lam + S."""
    if kind == 'exp':
        return -dt * np.exp(x) + Sn
    lam = softplus(x)
    return (-dt + Sn / lam) * sigmoid(x)


def glm_ll_grad(n, S, fS, w_imp, W_eff_col, bias, dt, kind, fstim=None, w_stim=None):
    """ll and its gradient w.r.t. (bias, w_stim-as-flat-feature-weights, w_imp)
    -- the quantity T.grad(glm.ll, ...) produces in coord_descent.py:27-30.
    Returns ll, g_bias, g_wstim (None if no stimulus), g_wimp (N,B)."""
    x, _, _ = glm_currents(n, fS, w_imp, W_eff_col, bias, fstim, w_stim)
    Sn = S[:, n]
    ll = glm_ll_from_x(x, Sn, dt, kind)
    r = glm_resid(x, Sn, dt, kind)
    g_bias = np.sum(r)
    # d x_t / d w[n',b] = fS[t,n',b] * W_eff[n']
    g_w = np.tensordot(r, fS, axes=(0, 0)) * W_eff_col[:, None]
    g_ws = fstim.T.dot(r) if fstim is not None else None
    return ll, g_bias, g_ws, g_w


def spatiotemporal_w_stim(w_t, w_x):
    """bkgd.py:214-220: w_stim = vec(w_t (x) w_x), index bt*Bx+bx."""
    return np.outer(w_t, w_x).reshape(-1)


def spatiotemporal_chain(g_wstim, w_t, w_x):
    """Chain rule of g_wstim (Bt*Bx) through w_stim = vec(w_t (x) w_x)."""
    G = g_wstim.reshape(len(w_t), len(w_x))
    return G.dot(w_x), G.T.dot(w_t)          # g_w_t, g_w_x


def dirichlet_chain(g_beta, g):
    """d beta_b / d g_c = sign(g_c) (delta_bc*sum|g| - |g_b|) / (sum|g|)^2
    (SURVEY Appendix A)."""
    ga = np.abs(g)
    s = np.sum(ga)
    J = (np.eye(len(g)) * s - ga[:, None]) / s ** 2 * np.sign(g)[None, :]   # J[b,c]
    return J.T.dot(g_beta)


# ----------------------------------------------------------------------------
# A9. priors
# ----------------------------------------------------------------------------

def bias_log_p(bias, mu, sigma):
    """bias.py:33."""
    return -0.5 / sigma ** 2 * (bias - mu) ** 2


def gaussian_log_p(value, mu, sigma):
    """priors.py:139."""
    return -0.5 / sigma ** 2 * np.sum((value - mu) ** 2)


def group_lasso_log_p(value, lam, mu, sigma):
    """priors.py:202: value is (groups, B)."""
    return -1.0 * lam * np.sum(np.sqrt(np.sum(((value - mu) / sigma) ** 2, axis=1)))


def group_lasso_grad(value, lam, mu, sigma):
    """Gradient of group_lasso_log_p.  A zero group gives 0/0 = NaN exactly like
    T.grad of the sqrt in the reference (callers zero NaN gradients,
    coord_descent.py:179-180)."""
    z = (value - mu) / sigma
    nrm = np.sqrt(np.sum(z ** 2, axis=1, keepdims=True))
    with np.errstate(invalid='ignore', divide='ignore'):
        return -lam * z / nrm / sigma


def basis_stim_log_p(w_stim):
    """bkgd.py:76 (sigma hard-coded to 0.01)."""
    return np.sum(-0.5 / (0.01 ** 2) * (w_stim - 0.0) ** 2)


def spatiotemporal_log_p(w_x, w_t, mu, sigma):
    """bkgd.py:223-224."""
    return (-0.5 / sigma ** 2 * np.sum((w_x - mu) ** 2)
            - 0.5 / sigma ** 2 * np.sum((w_t - mu) ** 2))


def dirichlet_log_p(gs, alpha):
    """impulse.py:320-322: sum_n (alpha-1)*sum log|g_n| - sum|g_n|."""
    lp = 0.0
    for g in gs:
        lp += (alpha - 1.0) * np.sum(np.log(np.abs(g))) - np.sum(np.abs(g))
    return lp


def gaussian_weight_log_p(W, mu, sigma, mu_ref=None, sigma_ref=None):
    """weights.py:64-71: off-diagonal prior + refractory prior on the diagonal."""
    N = W.shape[0]
    if mu_ref is None:
        return gaussian_log_p(W, mu, sigma)
    off = ~np.eye(N, dtype=bool)
    return gaussian_log_p(W[off], mu, sigma) + gaussian_log_p(np.diag(W), mu_ref, sigma_ref)


def erdos_renyi_log_p(A, rho):
    """graph.py:68-71 (lkhd_scale = 1)."""
    return np.sum(A * np.log(np.minimum(1.0 - 1e-8, rho)) +
                  (1 - A) * np.log(np.maximum(1e-8, 1.0 - rho)))


# ----------------------------------------------------------------------------
# A10. collapsed-Gibbs inner ll (gibbs.py:812-864, 910-937, 1002-1032)
# ----------------------------------------------------------------------------

def other_current(I_imp, A, W, n_pre, n_post):
    """gibbs.py:835-864: I_net with A[n_pre,n_post] forced to 0."""
    col = (A[:, n_post] * W[:, n_post]).astype(float).copy()
    col[n_pre] = 0.0
    return I_imp.dot(col)


def mcmc_inner_ll(ws, I_bias, I_stim, I_other, I_col, Sn, dt, kind):
    """gibbs.py:910-937: ll(w) with I_net = I_other + w * I_imp[:, n_pre] for
    every w in ws (the 10 Gauss-Hermite nodes + w=0 + ARS probes)."""
    out = np.zeros(len(ws))
    for i, w in enumerate(ws):
        x = I_bias + I_stim + I_other + w * I_col
        out[i] = glm_ll_from_x(x, Sn, dt, kind)
    return out


def gauss_hermite_nodes(mu_w, sigma_w, deg=10):
    """gibbs.py:787-789, 1004: W_nns = sqrt(2)*sigma*x_i + mu, weights omega_i."""
    x, w = np.polynomial.hermite.hermgauss(deg)
    return np.sqrt(2) * sigma_w * x + mu_w, w


# ----------------------------------------------------------------------------
# known-answer: time-domain superposition (population.py:275-282, 351-353)
# ----------------------------------------------------------------------------

def impulse_responses(ibasis, w_imp):
    """impulse.py:65 / 329: impulse[n',:] = w[n',:] . ibasis^T  -> (N,R)."""
    return w_imp.dot(ibasis.T)


def direct_currents(S, imps_post, W_eff_col, bias, I_stim=0.0):
    """What Population.simulate accumulates (population.py:351-353): every spike
    of n' at bin s adds A*W*imp[n',:] to X[s+1 : s+R+1, n].  `imps_post` is the
    (N,R) impulse matrix of the post-synaptic neuron.  This is the independent
    side of the reference's allclose(lam_true, lam_sim) check
    (generate_synth_data.py:125-129)."""
    nT, N = S.shape
    R = imps_post.shape[1]
    x = np.zeros(nT) + bias + I_stim
    for npre in range(N):
        h = W_eff_col[npre] * imps_post[npre]
        for s in np.nonzero(S[:, npre])[0]:
            e = min(nT, s + 1 + R)
            x[s + 1:e] += S[s, npre] * h[:e - s - 1]
    return x


# ----------------------------------------------------------------------------
# packing order (packvec.py:17-83, theano_func_wrapper.py:53-67)
# ----------------------------------------------------------------------------

def packdict(d):
    """packvec.py:17-45: sorted-key DFS concatenation; returns (vec, shapes)."""
    vec = np.zeros((0,))
    shapes = {}
    for k in sorted(d.keys()):
        v = d[k]
        if isinstance(v, dict):
            sv, ss = packdict(v)
            vec = np.concatenate((vec, sv))
            shapes[k] = ss
        elif isinstance(v, list) and v == []:
            continue
        else:
            v = np.asarray(v, dtype=float)
            shapes[k] = v.shape
            vec = np.concatenate((vec, v.reshape(-1)))
    return vec, shapes
