"""How the fused kernel's phases scale with the spike rate (events per window): dev tool.
    python tools/rate_sweep.py [rates...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H

rates = [float(x) for x in sys.argv[1:]] or [2.0, 10.0, 20.0, 40.0]
N, nT = 128, 300000
for r in rates:
    p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5, rate_hz=r)
    dev = p.device()
    out = []
    for dbg in (0, 1):
        dev.set_option(99, dbg)
        for i in range(3):
            dev.ll_grad(p.theta, p.Weff)
        out.append(dev.last_timing()[0])
    print("rate %5.1f Hz  events %8d  full %.3f ms  no-gen %.3f ms  gen %.3f ms"
          % (r, int(dev.info()['events']), out[0], out[1], out[0] - out[1]))
    dev.close()
