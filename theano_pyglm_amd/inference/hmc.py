"""
Hamiltonian Monte Carlo -- in-repo replacement for `hips.inference.hmc.hmc`, which the reference
imports (pyglm/inference/gibbs.py:15) from the un-vendored `hips` package.  The
algorithm is the published one (Neal 2011, "MCMC using Hamiltonian dynamics", fig. 2): Gaussian
momentum, a leapfrog trajectory of `n_steps` steps of size `step_sz`, Metropolis accept/reject on
the change of total energy.  The step-size controller follows the call signature the reference
uses (gibbs.py:306-316): an exponential moving average of the acceptance indicator, and a 2 %
multiplicative step change towards a target acceptance rate.

Two forms:
  * `hmc`          one chain, callbacks U(q) / grad_U(q) like the reference's nll / grad_nll;
  * `hmc_lockstep` M independent chains advanced together through one callback
                   U_and_grad(Q) -> (U (M,), G (M,d)).  The per-neuron conditional posteriors of the
                   GLM parameters are independent given the network, so one leapfrog step of all
                   N neurons is ONE batched ll+grad evaluation on the device.
Sample streams are not comparable with the reference (different random streams; "parity
unpinned" for sampler trajectories) -- tests pin the invariant distribution.
"""
import numpy as np

TGT_ACCEPT_RATE = 0.9
ACCEPT_TIME_CONST = 0.95
MIN_STEP_SZ, MAX_STEP_SZ = 1e-3, 1.0


def adapt_step_size(step_sz, avg_accept_rate, accepted, tgt_accept_rate=TGT_ACCEPT_RATE,
                    time_const=ACCEPT_TIME_CONST, min_step_sz=MIN_STEP_SZ, max_step_sz=MAX_STEP_SZ):
    """One controller update: returns (new_step_sz, new_avg_accept_rate)."""
    new_rate = time_const * avg_accept_rate + (1.0 - time_const) * float(accepted)
    factor = 1.02 if avg_accept_rate > tgt_accept_rate else 0.98
    return float(np.clip(step_sz * factor, min_step_sz, max_step_sz)), new_rate


def hmc(U, grad_U, step_sz, n_steps, q_curr, adaptive_step_sz=False, avg_accept_rate=0.9,
        tgt_accept_rate=TGT_ACCEPT_RATE, rng=None):
    """One HMC transition of a single chain.  U is the negative log density, grad_U its gradient.
    Returns q_next, or (q_next, new_step_sz, new_avg_accept_rate) with adaptive_step_sz."""
    rng = np.random if rng is None else rng
    q0 = np.array(q_curr, dtype=float)
    q = q0.copy()
    p = rng.standard_normal(q.shape)
    H0 = float(U(q0)) + 0.5 * np.sum(p * p)
    p = p - 0.5 * step_sz * np.asarray(grad_U(q))
    for i in range(n_steps):
        q = q + step_sz * p
        scale = 1.0 if i < n_steps - 1 else 0.5
        p = p - scale * step_sz * np.asarray(grad_U(q))
    H1 = float(U(q)) + 0.5 * np.sum(p * p)
    dH = H0 - H1
    accepted = bool(np.isfinite(H1) and np.log(rng.random_sample()) < dH)
    q_next = q if accepted else q0
    if not adaptive_step_sz:
        return q_next
    new_step, new_rate = adapt_step_size(step_sz, avg_accept_rate, accepted, tgt_accept_rate)
    return q_next, new_step, new_rate


def hmc_lockstep(U_and_grad, step_sz, n_steps, Q_curr, active=None, rng=None, UG_curr=None):
    """One HMC transition of M independent chains sharing a step size.

    U_and_grad(Q (M,d)) -> (U (M,), G (M,d)), the negative log densities and gradients of all
    chains in one call.  `active` (M,) bool freezes chains that have nothing to sample this round
    (zero momentum, never accepted).  Returns (Q_next (M,d), accepted (M,) bool, n_evals)."""
    rng = np.random if rng is None else rng
    Q0 = np.array(Q_curr, dtype=float)
    M = Q0.shape[0]
    active = np.ones(M, dtype=bool) if active is None else np.asarray(active, dtype=bool)
    P = rng.standard_normal(Q0.shape) * active[:, None]
    U0, G = U_and_grad(Q0) if UG_curr is None else UG_curr
    n_evals = 1 if UG_curr is None else 0
    H0 = np.asarray(U0, dtype=float) + 0.5 * np.sum(P * P, axis=1)
    Q = Q0.copy()
    P = P - 0.5 * step_sz * np.where(active[:, None], G, 0.0)
    U1 = None
    for i in range(n_steps):
        Q = Q + step_sz * P
        U1, G = U_and_grad(Q)
        n_evals += 1
        scale = 1.0 if i < n_steps - 1 else 0.5
        P = P - scale * step_sz * np.where(active[:, None], G, 0.0)
    H1 = np.asarray(U1, dtype=float) + 0.5 * np.sum(P * P, axis=1)
    with np.errstate(invalid='ignore'):
        accepted = active & np.isfinite(H1) & (np.log(rng.random_sample(M)) < H0 - H1)
    return np.where(accepted[:, None], Q, Q0), accepted, n_evals
