"""
Derivative-free adaptive rejection sampling -- in-repo replacement for
`hips.inference.ars.adaptive_rejection_sample`, which the reference imports (gibbs.py:14) from the
un-vendored `hips` package and uses to draw a synaptic weight from its log-concave conditional
posterior (gibbs.py:1087-1126).  The algorithm is the published one (Gilks 1992, "Derivative-free
adaptive rejection sampling for Gibbs sampling"): for a concave log density h known at abscissae
x_0 < ... < x_k the chords through neighbouring pairs, extended, bound h from above between the
*other* points, and the chords themselves bound it from below.  Samples are drawn from the
piecewise-exponential upper hull and accepted against the lower hull (squeeze, no evaluation) or
against h itself; a rejected point refines both hulls.
"""
import numpy as np


def _log_piece_mass(y_a, m, length):
    """log of int_0^length exp(y_a + m t) dt, length may be inf when m < 0."""
    if np.isinf(length):
        return y_a - np.log(-m)
    z = m * length
    if abs(z) < 1e-8:
        return y_a + np.log(length) + 0.5 * z
    if z > 0:
        return y_a + z + np.log(-np.expm1(-z)) - np.log(m)
    return y_a + np.log(-np.expm1(z)) - np.log(-m)


def _log_piece_masses(y_a, m, length):
    """_log_piece_mass for arrays of pieces (same formulas, element by element)."""
    y_a, m, length = np.asarray(y_a, dtype=float), np.asarray(m, dtype=float), np.asarray(length, dtype=float)
    out = np.empty_like(y_a)
    inf = np.isinf(length)
    with np.errstate(all='ignore'):
        z = m * length
        small = ~inf & (np.abs(z) < 1e-8)
        pos = ~inf & ~small & (z > 0)
        neg = ~inf & ~small & ~pos
        out[inf] = y_a[inf] - np.log(-m[inf])
        out[small] = y_a[small] + np.log(length[small]) + 0.5 * z[small]
        out[pos] = y_a[pos] + z[pos] + np.log(-np.expm1(-z[pos])) - np.log(m[pos])
        out[neg] = y_a[neg] + np.log(-np.expm1(z[neg])) - np.log(-m[neg])
    return out


def _sample_piece(y_a, m, length, u):
    """inverse CDF on one exponential piece, offset from its left end."""
    if np.isinf(length):
        return np.log1p(-u) / m                       # m < 0
    z = m * length
    if abs(z) < 1e-8:
        return u * length
    if z > 0:
        # stable for large z: t = length + log(u + (1-u) e^{-z}) / m
        return length + np.log(u + (1.0 - u) * np.exp(-z)) / m
    return np.log1p(u * np.expm1(z)) / m


class _Hull(object):
    def __init__(self, xs, hs, lb, ub):
        order = np.argsort(xs)
        self.x = list(np.asarray(xs, dtype=float)[order])
        self.h = list(np.asarray(hs, dtype=float)[order])
        self.lb, self.ub = lb, ub

    def insert(self, x, h):
        i = int(np.searchsorted(self.x, x))
        if (i < len(self.x) and self.x[i] == x) or (i > 0 and self.x[i - 1] == x):
            return
        self.x.insert(i, x)
        self.h.insert(i, h)

    def slopes(self):
        x, h = np.asarray(self.x), np.asarray(self.h)
        return np.diff(h) / np.diff(x)

    def pieces(self):
        """Upper hull as a list of (left, length, value at left, slope); left tail pieces are
        stored mirrored as (right end, length, value, -slope, mirrored=True)."""
        x, h, m = self.x, self.h, self.slopes()
        k = len(x) - 1                                # number of chords
        out = []
        # left of x_0: chord 0 extended
        if np.isinf(self.lb):
            out.append((x[0], np.inf, h[0], -m[0], True))
        elif self.lb < x[0]:
            out.append((x[0], x[0] - self.lb, h[0], -m[0], True))
        for j in range(k):
            a, b = x[j], x[j + 1]
            left = m[j - 1] if j > 0 else None        # chord j-1 extended to the right of x_j
            right = m[j + 1] if j < k - 1 else None   # chord j+1 extended to the left of x_{j+1}
            if left is None and right is None:
                out.append((a, b - a, h[j], m[j], False))            # two points only: the chord
            elif left is None:
                out.append((b, b - a, h[j + 1], -right, True))
            elif right is None:
                out.append((a, b - a, h[j], left, False))
            else:
                # intersection of  h_j + left (t - a)  and  h_{j+1} + right (t - b)
                if left - right > 1e-300:
                    z = (h[j + 1] - h[j] + left * a - right * b) / (left - right)
                    z = min(max(z, a), b)
                else:                                  # parallel within rounding: locally linear
                    z = b
                if z > a:
                    out.append((a, z - a, h[j], left, False))
                if z < b:
                    out.append((b, b - z, h[j + 1], -right, True))
        if np.isinf(self.ub):
            out.append((x[k], np.inf, h[k], m[k - 1], False))
        elif self.ub > x[k]:
            out.append((x[k], self.ub - x[k], h[k], m[k - 1], False))
        return out

    def lower(self, t):
        x, h = self.x, self.h
        if t < x[0] or t > x[-1]:
            return -np.inf
        j = min(max(int(np.searchsorted(x, t)) - 1, 0), len(x) - 2)
        return h[j] + (h[j + 1] - h[j]) * (t - x[j]) / (x[j + 1] - x[j])


def adaptive_rejection_sample(func, xs, v_xs, domain=(-np.inf, np.inf), stepsz=1.0, rng=None,
                              max_evals=200, return_evals=False):
    """One draw from the density proportional to exp(func(x)) on `domain`, func concave.

    xs, v_xs : initial abscissae and func values there (non-finite entries are dropped);
    stepsz   : stride used to extend the abscissae until the hull is integrable on an unbounded
               domain (leftmost chord must rise, rightmost chord must fall)."""
    rng = np.random if rng is None else rng
    lb, ub = domain
    xs = np.asarray(xs, dtype=float)
    v_xs = np.asarray(v_xs, dtype=float)
    ok = np.isfinite(v_xs) & np.isfinite(xs)
    hull = _Hull(xs[ok], v_xs[ok], lb, ub)
    n_evals = 0

    def evaluate(t):
        return float(func(float(t)))

    if len(hull.x) == 0:
        raise ValueError("adaptive_rejection_sample needs at least one finite starting point")
    step = stepsz
    while len(hull.x) < 3:
        t = hull.x[-1] + step if hull.x[-1] + step < ub else 0.5 * (hull.x[-1] + ub)
        hull.insert(t, evaluate(t))
        n_evals += 1
    # make the tails integrable
    step = stepsz
    while np.isinf(lb) and not hull.slopes()[0] > 0:
        t = hull.x[0] - step
        hull.insert(t, evaluate(t))
        n_evals += 1
        step *= 2.0
        if n_evals > max_evals:
            raise RuntimeError("ARS: could not bracket the mode on the left")
    step = stepsz
    while np.isinf(ub) and not hull.slopes()[-1] < 0:
        t = hull.x[-1] + step
        hull.insert(t, evaluate(t))
        n_evals += 1
        step *= 2.0
        if n_evals > max_evals:
            raise RuntimeError("ARS: could not bracket the mode on the right")
    # drop -inf abscissae produced while bracketing (outside the support)
    keep = [i for i, v in enumerate(hull.h) if np.isfinite(v)]
    if len(keep) < 3:
        raise RuntimeError("ARS: fewer than three finite abscissae")
    hull.x = [hull.x[i] for i in keep]
    hull.h = [hull.h[i] for i in keep]

    while True:
        pieces = hull.pieces()
        logm = _log_piece_masses([p[2] for p in pieces], [p[3] for p in pieces], [p[1] for p in pieces])
        w = np.exp(logm - np.max(logm))
        i = int(np.searchsorted(np.cumsum(w), rng.random_sample() * np.sum(w)))
        i = min(i, len(pieces) - 1)
        anchor, length, y_a, m, mirrored = pieces[i]
        off = _sample_piece(y_a, m, length, rng.random_sample())
        t = anchor - off if mirrored else anchor + off
        upper = y_a + m * off
        log_u = np.log(rng.random_sample())
        if log_u <= hull.lower(t) - upper:
            break
        ht = evaluate(t)
        n_evals += 1
        if log_u <= ht - upper:
            break
        if np.isfinite(ht):
            hull.insert(t, ht)
        if n_evals > max_evals:
            raise RuntimeError("ARS: too many rejections (is the density log-concave?)")
    return (t, n_evals) if return_evals else t
