"""Phase timeline of k_fused5 (dev tool).  Needs a -DPGL_PROF build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPGL_PROF pglm_capi.hip -o ../libpyglm_hip_prof.so
    PYGLM_HIP_LIB=$PWD/theano_pyglm_amd/libpyglm_hip_prof.so python tools/phase_profile.py [N] [T]
Prints, per pass, the mean shader cycles per tile a wave spends between the phase marks."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
nT = int(round(T / 0.001))
p = H.Problem(N, nT, H.std_ibasis(), seed=1234, w_scale=0.5)
dev = p.device()
info = dev.info()
print(info)
for i in range(4):
    ll, g = dev.ll_grad(p.theta, p.Weff)
    print("fused %.3f ms" % dev.last_timing()[0])
lib = _lib.load()
buf = np.zeros((2, 4096, 8, 12), dtype=np.int64)
lib.pgl_debug_prof.argtypes = [C.c_void_p, C.c_int]
rc = lib.pgl_debug_prof(buf.ctypes.data_as(C.c_void_p), buf.size)
assert rc == 0
nblk = int(info['blocks'])
tiles = (nT + 15) // 16 / info['chunks']
names1 = ['fwd', 'epi tail', 'barrier1', 'barrier1b+counts', 'r store+bwd', 'vmcnt(0)', 'barrier2', 'epi exp', 'epi series', 'epi spikes']
names2 = ['vmcnt(0)', 'barrier', 'dma issue', 'bwd']
for ps, names in ((0, names1), (1, names2)):
    d = buf[ps, :nblk - 1].astype(float) / tiles          # last chunk is short
    tot = d[:, :, :10].sum(axis=2)
    cyc, rt = buf[ps, :nblk - 1, :, 10].astype(float), buf[ps, :nblk - 1, :, 11].astype(float)
    print("pass %d: shader clock over the tile loop %.3f GHz (s_memtime / s_memrealtime at 100 MHz), loop %.3f ms"
          % (ps + 1, (cyc / rt).mean() * 0.1, rt.mean() / 1e5))
    print("pass %d: cycles per tile and wave (mean over %d workgroups; total %.0f)" % (ps + 1, nblk - 1, tot.mean()))
    for i, nm in enumerate(names):
        print("  %-12s mean %8.0f   waves0-3 %8.0f  waves4-7 %8.0f   min %8.0f max %8.0f"
              % (nm, d[:, :, i].mean(), d[:, :4, i].mean(), d[:, 4:, i].mean(), d[:, :, i].min(), d[:, :, i].max()))

ts = np.zeros((4096, 5), dtype=np.int64)
if hasattr(lib, 'pgl_debug_prof_ts'):
    lib.pgl_debug_prof_ts.argtypes = [C.c_void_p, C.c_int]
    if lib.pgl_debug_prof_ts(ts.ctypes.data_as(C.c_void_p), ts.size) == 0:
        # (the table holds the LAST launch that wrote it: pass 2 when the gradient was requested)
        ts = ts[:min(nblk, 4096), :4].astype(float) / 100.0
        t0 = ts[:, 0].min()
        print("last launch, workgroup timeline (us): entry mean %.1f max %.1f | entry->loop mean %.1f max %.1f | loop mean %.1f | "
              "loop end->exit mean %.1f max %.1f | last exit %.1f"
              % ((ts[:, 0] - t0).mean(), (ts[:, 0] - t0).max(), (ts[:, 1] - ts[:, 0]).mean(), (ts[:, 1] - ts[:, 0]).max(),
                 (ts[:, 2] - ts[:, 1]).mean(), (ts[:, 3] - ts[:, 2]).mean(), (ts[:, 3] - ts[:, 2]).max(), ts[:, 3].max() - t0))
