import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
N, nT = 128, 600000
for ws, rate in ((0.5, 20.0), (0.0, 20.0), (0.5, 0.0)):
    p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=ws, rate_hz=rate)
    if ws == 0.0: p.theta[:] = 0.0
    dev = p.device()
    for i in range(6):
        ll, g = dev.ll_grad(p.theta, p.Weff)
    lib = _lib.load()
    buf = np.zeros((2, 4096, 8, 12), dtype=np.int64)
    lib.pgl_debug_prof.argtypes = [C.c_void_p, C.c_int]
    lib.pgl_debug_prof(buf.ctypes.data_as(C.c_void_p), buf.size)
    out = []
    for ps in (0, 1):
        cyc, rt = buf[ps, :255, :, 10].astype(float), buf[ps, :255, :, 11].astype(float)
        out.append("pass%d %.3f GHz %.3f ms %.0f cyc/tile" % (ps + 1, (cyc / rt).mean() * 0.1, rt.mean() / 1e5, cyc.mean() / 146.5))
    print("w_scale %.1f rate %.0f: fused %.3f ms | %s" % (ws, rate, dev.last_timing()[0], " | ".join(out)))
    dev.close()
