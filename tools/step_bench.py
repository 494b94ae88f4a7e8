"""Steady-state cost of one ll+grad evaluation of a BASELINE configuration when evaluations are queued back to back
on one stream (device pointers, no host synchronisation in between) -- what an optimizer loop pays per evaluation,
as opposed to the HIP-event span of a single isolated call (which includes the host's submission latency of every
launch because the stream runs empty).  Dev tool.   python tools/step_bench.py C1|C2|C3|C5 [n_hi]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import helpers as H
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C1'
N, T, ib, kind, Ds, ws = {'C1': (4, 60.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C2': (32, 300.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C3': (128, 600.0, H.std_ibasis(), 'explinear', 0, 0.5),
                          'C5': (64, 300.0, H.st_ibasis(), 'exp', 9, 0.02)}[cfg]
import os
BIAS = os.environ.get('BIAS')          # e.g. BIAS=5 RATE=5: a low-rate population (currents inside the |x| < 12 band)
p = H.Problem(N, int(round(T / 0.001)), ib, kind=kind, Dstim=Ds, seed=1234, w_scale=ws,
              bias_mu=float(BIAS) if BIAS else None, rate_hz=float(os.environ.get('RATE', '20')))
dev = p.device()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
dev.set_stream(st.cuda_stream)
import os
from theano_pyglm_amd import _lib
TIMING = int(os.environ.get('TIMING', '1'))
dev.set_option(_lib.OPT_TIMING, TIMING)      # HIP events around every TIMING-th evaluation (0: none)
d_theta = torch.from_numpy(p.theta).cuda(); d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
d_ll = torch.zeros(N, dtype=torch.float64, device='cuda'); d_g = torch.zeros((N, p.P), dtype=torch.float64, device='cuda')
for _ in range(20):
    dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
torch.cuda.synchronize(); dev.timing_summary(reset=True)
K = 200
t0 = time.perf_counter()
for _ in range(K):
    dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e3
n, fused, total = dev.timing_summary(reset=True) if TIMING else (0, float('nan'), float('nan'))
info = dev.info()
print("%s (events every %d calls): %.4f ms per evaluation back to back (fused kernel %.4f ms, event span of one call %.4f ms); %.1f TFLOP/s (alg.) = %.3f of 78.6 on the whole evaluation"
      % (cfg, TIMING, wall, fused, total, info['flops'] / wall / 1e9, info['flops'] / wall / 1e9 / 78.6))
