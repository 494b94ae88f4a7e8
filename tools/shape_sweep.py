"""Shape sweep of the population ll+grad (standard_glm, B = 5, explinear, Poisson 20 Hz): for every population size the
kernel the dispatcher picks (make_plan) and the forced alternatives, fused-kernel and whole-evaluation time, fraction
of the f64 MFMA peak.  Writes a markdown table (profiles/r05_shape_sweep.md via tools/r5_profiles.sh).

    python tools/shape_sweep.py [T seconds = 300] [--alts] [--no-helpers] [--chunk-major] [N ...]
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

PEAK = 78.6
args = [a for a in sys.argv[1:] if not a.startswith('--')]
alts = '--alts' in sys.argv
ptw = [int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('--ptw=')]
slc = [int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('--slice=')]
nohlp = '--no-helpers' in sys.argv                                  # dev option 92: two-pass kernel without helper waves (A/B)
cmajor = '--chunk-major' in sys.argv                              # dev option 91: wide populations on the chunk-major grid (A/B)
T = float(args[0]) if args else 300.0
Ns = [int(a) for a in args[1:]] or [16, 32, 48, 64, 80, 96, 128, 160, 256]
nT = int(T * 1000)
NAMES = {1: 'k_fused (4-wave, round 1)', 2: 'k_fused2 (K split, in-kernel features)', 3: 'k_fused2 f32', 4: 'k_fused3 (two-pass, in-kernel features)',
         5: 'k_fused5 (two-pass, resident tiles)', 6: 'k_fused6 (K split, resident tiles)', 7: 'k_fused7 (no K split, resident tiles)'}
print("| N | K = 5N | selection | kernel | k-tiles | blocks | fused ms | evaluation ms | TFLOP/s | fraction of f64 MFMA peak |")
print("|---|---|---|---|---|---|---|---|---|---|")
for N in Ns:
    p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5)
    for sel, opt in [('auto', 0)] + ([('force 7', 7), ('force 6', 6), ('force 5', 4), ('force 3', 3), ('force 2', 2)] if alts else []):
        dev = p.device()
        try:
            dev.set_option(_lib.OPT_KERNEL, opt)
            if nohlp:
                dev.set_option(92, 1)
            if cmajor:
                dev.set_option(91, 1)
            if slc:
                dev.set_option(93, slc[0])                      # dev: feature columns per slice of the 3-phase path
            if ptw:
                dev.set_option(98, ptw[0])                      # dev: post tiles per workgroup of the K-split kernels
            info = dev.info()
            if opt and {7: 7, 6: 6, 4: 5, 3: 4, 2: 2}[opt] != int(info['kernel_version']):
                continue                                        # the forced form does not exist for this shape
            for i in range(6):
                dev.ll_grad(p.theta, p.Weff)
            ts = []
            for i in range(5):
                dev.ll_grad(p.theta, p.Weff)
                ts.append(dev.last_timing())
            fused, total = np.median([t[0] for t in ts]), np.median([t[1] for t in ts])
            flops = 4.0 * nT * N * N * 5
            print("| %d | %d | %s | %s | %d | %d | %.3f | %.3f | %.1f | %.2f |"
                  % (N, 5 * N, sel, NAMES[int(info['kernel_version'])], info['ktiles'], info['blocks'], fused, total,
                     flops / fused / 1e9, flops / fused / 1e9 / PEAK), flush=True)
        except Exception as e:
            print("| %d | %d | %s | failed: %s | | | | | | |" % (N, 5 * N, sel, str(e)[:60]), flush=True)
        finally:
            dev.close()
