"""A recording ten times the length of C3 (N = 128, nT = 6 000 000: 31.5 GB of resident feature tiles, element offsets
beyond 2^31): internal consistency of the population ll+grad where no oracle can follow in reasonable time --
(1) the first and the LAST 600 000 bins through set_time_range equal fresh 600 000-bin handles built from those bins
    (plus the R bins of history), (2) ten sub-ranges add up to the whole, (3) the oracle on 4 096 bins near the end.
Dev tool:  python3 tools/r4/long_recording.py [nT]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
from oracle import glm_oracle as O

N = 128
nT = int(sys.argv[1]) if len(sys.argv) > 1 else 6000000
ib = H.std_ibasis()
R, B = ib.shape
P = 1 + N * B
rng = np.random.default_rng(77)
S = np.empty((nT, N), dtype=np.uint8)
for i in range(0, nT, 1000000):                       # (chunks: rng.poisson returns int64)
    S[i:i + 1000000] = np.minimum(rng.poisson(0.02, size=(min(1000000, nT - i), N)), 10)
theta = np.zeros((N, P)); theta[:, 0] = 20.0 + 0.1 * rng.standard_normal(N); theta[:, 1:] = 0.5 * rng.standard_normal((N, N * B))
Weff = np.ones((N, N))


def handle(Sx):
    d = _lib.DeviceGlm(N, Sx.shape[0], B, R, 'explinear', 0.001)
    d.set_spikes(Sx); d.set_basis(ib)
    return d


t0 = time.perf_counter()
big = handle(S)
ll, g = big.ll_grad(theta, Weff)
print("nT = %d: first evaluation (tiles built) %.2f s; info %s" % (nT, time.perf_counter() - t0, {k: big.info()[k] for k in ('kernel_version', 'bytes_tiles') if k in big.info()}))
t0 = time.perf_counter()
for _ in range(3):
    ll, g = big.ll_grad(theta, Weff)
print("whole recording: %.2f ms per ll+grad (host-pointer API)" % ((time.perf_counter() - t0) / 3 * 1e3))
L = 600000
parts_ll, parts_g = np.zeros(N), np.zeros((N, P))
for i in range(0, nT, L):
    big.set_time_range(i, min(nT, i + L))
    a, b = big.ll_grad(theta, Weff)
    parts_ll += a; parts_g += b
    if i == 0:
        first = (a, b)
    last = (a, b, i)
print("sub-ranges add up: ll %.2e, grad %.2e (relative to max)" % (np.max(np.abs(parts_ll - ll) / np.abs(ll)), np.max(np.abs(parts_g - g)) / np.max(np.abs(g))))
small = handle(S[:L])
a, b = small.ll_grad(theta, Weff)
print("first %d bins vs a fresh handle: ll %.2e grad %.2e" % (L, np.max(np.abs(a - first[0]) / np.abs(a)), np.max(np.abs(b - first[1])) / np.max(np.abs(b))))
small.close()
i0 = last[2]
H0 = 208                                   # history: >= R bins, a multiple of 16 (range starts are tile-aligned)
tail = handle(S[i0 - H0:])
tail.set_time_range(H0, nT - i0 + H0)
a, b = tail.ll_grad(theta, Weff)
print("last %d bins (from bin %d) vs a fresh handle with %d bins of history: ll %.2e grad %.2e" % (nT - i0, i0, H0, np.max(np.abs(a - last[0]) / np.abs(a)), np.max(np.abs(b - last[1])) / np.max(np.abs(b))))
tail.close()
# the numpy oracle on a short range near the end
t_hi = nT - 992; t_lo = t_hi - 4096
big.set_time_range(t_lo, t_hi)
a, b = big.ll_grad(theta, Weff)
Ssub = S[t_lo - R:t_hi].astype(float)
fS = O.convolve_with_basis(Ssub, ib)[R:]                     # (bins, N, B)
lo, go = np.zeros(N), np.zeros((N, P))
for n in range(N):
    l, gb, _, gw = O.glm_ll_grad(n, Ssub[R:], fS, theta[n, 1:].reshape(N, B), Weff[:, n], theta[n, 0], 0.001, 'explinear')
    lo[n] = l; go[n, 0] = gb; go[n, 1:] = gw.ravel()
print("oracle on bins [%d, %d): ll %.2e grad %.2e" % (t_lo, t_hi, np.max(np.abs(a - lo) / np.abs(lo)), np.max(np.abs(b - go)) / np.max(np.abs(go))))
# the batched Gibbs inner ll on the long recording: halves add up, regime-split == all-f64 kernel
big.set_time_range(0, nT)
Wg = 0.01 * rng.standard_normal((N, N))
big.gibbs_prepare_all(theta, Wg)
cols = np.arange(N); pre = np.full(N, 11)
ws = np.tile(np.concatenate((np.sqrt(2) * np.polynomial.hermite.hermgauss(10)[0], [0.0])), (N, 1))
whole = big.gibbs_ll_cols(cols, pre, Wg[pre, cols], ws)
t0 = time.perf_counter()
for _ in range(3):
    whole = big.gibbs_ll_cols(cols, pre, Wg[pre, cols], ws)
print("Gibbs inner ll, 128 pairs x 11 weights on %d bins: %.2f ms per launch" % (nT, (time.perf_counter() - t0) / 3 * 1e3))
big.set_option(_lib.OPT_GIBBS_KERNEL, 1); f64 = big.gibbs_ll_cols(cols, pre, Wg[pre, cols], ws); big.set_option(_lib.OPT_GIBBS_KERNEL, 0)
fin = np.isfinite(whole) & np.isfinite(f64)
print("   regime-split vs all-f64 kernel: %.2e (finite %.3f)" % (np.max(np.abs(whole[fin] - f64[fin]) / np.abs(f64[fin])), fin.mean()))
half = (nT // 32) * 16
acc = 0.0
for lo, hi in ((0, half), (half, nT)):
    big.set_time_range(lo, hi); big.gibbs_prepare_all(theta, Wg)
    acc = acc + big.gibbs_ll_cols(cols, pre, Wg[pre, cols], ws)
print("   two halves add up: %.2e" % np.max(np.abs(acc[fin] - whole[fin]) / np.abs(whole[fin])))
big.close()
