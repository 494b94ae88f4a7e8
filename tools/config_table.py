"""Timing of every BASELINE.json configuration on one GPU (dev tool; numbers quoted in DESIGN.md)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib


def run(name, N, T, ibasis, kind, Dstim=0):
    nT = int(round(T / 0.001))
    p = H.Problem(N, nT, ibasis, kind=kind, Dstim=Dstim, seed=1234, w_scale=0.5 if kind == 'explinear' else 0.02)
    dev = p.device()
    for i in range(4):
        ll, g = dev.ll_grad(p.theta, p.Weff)
    fused, total = dev.last_timing()
    _, _ = dev.ll_grad(p.theta, p.Weff, want_grad=False)
    f2, t2 = dev.last_timing()
    info = dev.info()
    print("| %s | %d | %d | %d | %.3f | %.3f | %.1f | %.3f |" % (name, N, nT, info['ktiles'] * 16, total, fused,
                                                              info['flops'] / fused / 1e9, t2))
    dev.close()


print("| config | N | nT | padded K | ll+grad total ms | fused kernel ms | TFLOP/s (alg.) | ll only ms |")
print("|---|---|---|---|---|---|---|---|")
run("C1 standard_glm", 4, 60.0, H.std_ibasis(), 'explinear')
run("C2 standard_glm", 32, 300.0, H.std_ibasis(), 'explinear')
run("C3 standard_glm", 128, 600.0, H.std_ibasis(), 'explinear')
run("C5 spatiotemporal (D_stim=3)", 64, 300.0, H.st_ibasis(), 'exp', Dstim=9)
