"""Timing of every BASELINE.json configuration on one GPU (profiles/rNN_config_table.md).
    python tools/config_table.py [out.json]"""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

rows = []


import os
ONLY = [t for t in os.environ.get('CFG_ONLY', '').split(',') if t]    # e.g. CFG_ONLY="C2 standard_glm,C5 spatio"


def run(name, N, T, ibasis, kind, Dstim=0, kernel=0, n_hi=None, sep_D=0, tap_rate=False):
    if ONLY and not any(name.startswith(t) for t in ONLY):
        return
    nT = int(round(T / 0.001))
    p = H.Problem(N, nT, ibasis, kind=kind, Dstim=Dstim, seed=1234, w_scale=0.5 if kind == 'explinear' else 0.02)
    dev = p.device()
    if kernel:
        dev.set_option(_lib.OPT_KERNEL, kernel)
    n_hi = N if n_hi is None else n_hi
    if sep_D:             # separable stimulus (stress variant of C5): theta rows [bias, w_t(3), w_x(D), w_ir]
        rng = np.random.default_rng(1234 + 5)
        stim = rng.standard_normal((int(round(T / 0.1)), sep_D))
        dev.set_stimulus_separable(stim, 0.1, np.ascontiguousarray(H.golden()['lr2d_ibasis_t']), None)
        if tap_rate:
            dev.set_option(94, 2)
        p.theta = np.concatenate((p.theta[:, :1], 0.3 * rng.standard_normal((N, 3)), 0.05 * rng.standard_normal((N, sep_D)),
                                  p.theta[:, 1:]), axis=1)
    th = p.theta[:n_hi]
    for i in range(6):
        ll, g = dev.ll_grad(th, p.Weff, 0, n_hi)
    fused, total = dev.last_timing()
    _, _ = dev.ll_grad(th, p.Weff, 0, n_hi, want_grad=False)
    f2, t2 = dev.last_timing()
    info = dev.info(0, n_hi)
    tf = info['flops'] / fused / 1e9
    print("| %s | %d | %d | %d | %d | %.3f | %.3f | %.1f | %.2f | %.3f |"
          % (name, n_hi, nT, info['ktiles'] * 16, info['kernel_version'], total, fused, tf, tf / 78.6, t2))
    rows.append({'config': name, 'post_neurons': n_hi, 'N': N, 'nT': nT, 'kernel_version': int(info['kernel_version']),
                 'total_ms': total, 'fused_ms': fused, 'tflops': tf, 'frac_f64_mfma_peak': tf / 78.6, 'll_only_ms': t2})
    dev.close()


print("| config | post neurons | nT | padded K | kernel | ll+grad total ms | fused kernel ms | TFLOP/s (alg.) | frac of 78.6 | ll only ms |")
print("|---|---|---|---|---|---|---|---|---|---|")
run("C1 standard_glm", 4, 60.0, H.std_ibasis(), 'explinear')
run("C2 standard_glm", 32, 300.0, H.std_ibasis(), 'explinear')
run("C2 standard_glm (resident K-split)", 32, 300.0, H.std_ibasis(), 'explinear', kernel=6)
run("C2 standard_glm (in-kernel features)", 32, 300.0, H.std_ibasis(), 'explinear', kernel=2)
run("C3 standard_glm", 128, 600.0, H.std_ibasis(), 'explinear')
run("C5 spatiotemporal (D_stim=3)", 64, 300.0, H.st_ibasis(), 'exp', Dstim=9)
run("C5 (resident K-split)", 64, 300.0, H.st_ibasis(), 'exp', Dstim=9, kernel=6)
run("C5 (in-kernel features)", 64, 300.0, H.st_ibasis(), 'exp', Dstim=9, kernel=2)
run("C5 stress (D_stim=1024, separable, frame-rate kernels)", 64, 300.0, H.st_ibasis(), 'exp', sep_D=1024)
run("C5 stress (D_stim=1024, separable, tap-rate kernels)", 64, 300.0, H.st_ibasis(), 'exp', sep_D=1024, tap_rate=True)
for nh in (64, 32, 16):
    run("C3 neuron shard", 128, 600.0, H.std_ibasis(), 'explinear', n_hi=nh)
run("C3 neuron shard 64 (two-pass resident)", 128, 600.0, H.std_ibasis(), 'explinear', kernel=4, n_hi=64)
run("C3 neuron shard 48 (two-pass resident)", 128, 600.0, H.std_ibasis(), 'explinear', kernel=4, n_hi=48)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
