"""Quick single-GPU timing of the fused kernel at a given shape (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
f32 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dbgs = [int(x) for x in sys.argv[5].split(',')] if len(sys.argv) > 5 else [0]
nT = int(round(T / 0.001))
t0 = time.time()
p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5)
print("gen %.1fs" % (time.time() - t0)); t0 = time.time()
dev = p.device(f32=bool(f32), nchunks=nch)
import os
if os.environ.get('PTW'): dev.set_option(98, int(os.environ['PTW']))
print("upload %.1fs" % (time.time() - t0), dev.info())
for dbg in dbgs:
  dev.set_option(99, dbg)
  print("dbg", dbg)
  for i in range(3):
    t0 = time.time()
    ll, g = dev.ll_grad(p.theta, p.Weff)
    wall = time.time() - t0
    fused, total = dev.last_timing()
    info = dev.info()
    if i == 2: print("iter %d wall %.2f ms fused %.3f ms total %.3f ms -> %.1f TFLOP/s (f64 alg), ll0 %.6f"
          % (i, wall * 1e3, fused, total, info['flops'] / fused / 1e9, ll[0]))
t0 = time.time()
ll, _ = dev.ll_grad(p.theta, p.Weff, want_grad=False)
print("ll-only wall %.2f ms fused %.3f" % ((time.time() - t0) * 1e3, dev.last_timing()[0]))
