"""Table and prototype of pgl_softplus_tail_tab (csrc/pglm_gibbs.hip.h): log1p(exp(-a)) for a in [0, 12] in one table
step.  `python3 tools/ubench/softplus_tail_table.py` prints the table rows ({L0, s} per interval, rounded from 60-digit
values) and the error of the device algorithm (numpy f64, no fused multiply-adds) against the 60-digit function."""
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 60
NJ = 97                                          # a0 = j / 8, j = 0 .. 96


def table():
    rows = []
    for j in range(NJ):
        e0 = (-(Decimal(j) / 8)).exp()
        rows.append((float((1 + e0).ln()), float(e0 / (1 + e0))))
    return np.array(rows)


def tail(a, T):
    """The device algorithm: j = rint(8 a), v = j/8 - a, m = expm1(v) to v^8/8!, w = s m, L0 + log1p(w) to w^10/10."""
    jd = np.rint(a * 8.0)
    v = jd * 0.125 - a
    L0, s = T[jd.astype(int), 0], T[jd.astype(int), 1]
    q = v / 40320 + 1 / 5040
    for c in (1 / 720, 1 / 120, 1 / 24, 1 / 6, 0.5, 1.0):
        q = q * v + c
    w = (s * v) * q
    p = w * (-0.1) + 1 / 9
    for c in (-1 / 8, 1 / 7, -1 / 6, 1 / 5, -1 / 4, 1 / 3, -0.5, 1.0):
        p = p * w + c
    return w * p + L0


def reference(a):
    return np.array([float((1 + (-Decimal(float(x))).exp()).ln()) for x in a])


if __name__ == '__main__':
    T = table()
    for L0, s in T:
        print("    {%s, %s}," % (float(L0).hex(), float(s).hex()))
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(0, 12.0000005, 50000), np.arange(NJ) / 8.0, np.arange(NJ - 1) / 8.0 + 0.0625])
    err = np.abs(tail(a, T) - reference(a))
    print("max abs error %.2e, max relative %.2e" % (err.max(), (err / reference(a)).max()))
