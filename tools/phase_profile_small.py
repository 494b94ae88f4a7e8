"""Phase timeline of the resident-tile kernels for short feature rows, k_fused6 / k_fused7 (dev tool).
Needs a -DPGL_PROF build (tools/build_variant.sh prof -DPGL_PROF):
    PYGLM_HIP_LIB=$PWD/theano_pyglm_amd/libpyglm_hip_prof.so python tools/phase_profile_small.py C2|C5|C1|N<size> [n_hi] [PGL_OPT_KERNEL]
Prints the mean shader cycles per step (one 16-bin tile, or MT tiles for k_fused6 with MT = 2) a wave spends
between the phase marks."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
if cfg == 'C2':
    N, T, ib, kind, Ds, ws = 32, 300.0, H.std_ibasis(), 'explinear', 0, 0.5
elif cfg == 'C1':
    N, T, ib, kind, Ds, ws = 4, 60.0, H.std_ibasis(), 'explinear', 0, 0.5
elif cfg.startswith('N'):                                     # N16, N48, ...: standard_glm of that size, T = 300 s
    N, T, ib, kind, Ds, ws = int(cfg[1:]), 300.0, H.std_ibasis(), 'explinear', 0, 0.5
elif cfg == 'C3':
    N, T, ib, kind, Ds, ws = 128, 600.0, H.std_ibasis(), 'explinear', 0, 0.5
else:
    N, T, ib, kind, Ds, ws = 64, 300.0, H.st_ibasis(), 'exp', 9, 0.02
n_hi = int(sys.argv[2]) if len(sys.argv) > 2 else N
force = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # PGL_OPT_KERNEL (7: the sub-phases of the rate epilogue are marked there)
nT = int(round(T / 0.001))
p = H.Problem(N, nT, ib, kind=kind, Dstim=Ds, seed=1234, w_scale=ws)
dev = p.device()
if force:
    dev.set_option(_lib.OPT_KERNEL, force)
info = dev.info(0, n_hi)
print(cfg, info)
for i in range(4):
    ll, g = dev.ll_grad(p.theta[:n_hi], p.Weff, 0, n_hi)
    print("fused %.3f ms total %.3f ms" % dev.last_timing())
lib = _lib.load()
buf = np.zeros((2, 4096, 8, 12), dtype=np.int64)
lib.pgl_debug_prof.argtypes = [C.c_void_p, C.c_int]
assert lib.pgl_debug_prof(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
nblk = min(int(info['blocks']), 4096)
nw = int(info['threads']) // 64
ver = int(info['kernel_version'])
tiles = (nT + 15) // 16 / info['chunks']
names = {6: ['vmcnt(0)', 'barrier landed', 'issue dma+counts', 'fwd + X store', 'barrier partials', 'epilogue',
             'barrier resid', 'bwd'],
         7: ['vmcnt(0)', 'barrier landed', 'issue dma+counts', 'fwd', 'epi tail', 'bwd', '-', 'epi exp', 'epi series',
             'epi spikes']}[ver]
d = buf[0, :nblk - 1, :nw].astype(float) / tiles
cyc, rt = buf[0, :nblk - 1, :nw, 10].astype(float), buf[0, :nblk - 1, :nw, 11].astype(float)
print("kernel %d: %d waves/WG, %.1f tiles per chunk; shader clock over the loop %.3f GHz, loop %.3f ms"
      % (ver, nw, tiles, (cyc / rt).mean() * 0.1, rt.mean() / 1e5))
tot = d[:, :, :10].sum(axis=2)
print("cycles per tile and wave: total %.0f" % tot.mean())
for i, nm in enumerate(names):
    print("  %-18s mean %8.0f  (%4.1f %%)  min %8.0f max %8.0f" % (nm, d[:, :, i].mean(), 100 * d[:, :, i].mean() / tot.mean(),
                                                               d[:, :, i].min(), d[:, :, i].max()))

ts = np.zeros((4096, 5), dtype=np.int64)
lib.pgl_debug_prof_ts.argtypes = [C.c_void_p, C.c_int]
if lib.pgl_debug_prof_ts(ts.ctypes.data_as(C.c_void_p), ts.size) == 0:
    hw = ts[:nblk, 4].copy()
    ts = ts[:nblk, :4].astype(float) / 100.0                  # us
    t0 = ts[:, 0].min()
    print("workgroup timeline (us, relative to the first entry): entry mean %.1f max %.1f | entry->loop mean %.1f max %.1f | "
          "loop mean %.1f | loop end->exit mean %.1f max %.1f | last exit %.1f"
          % ((ts[:, 0] - t0).mean(), (ts[:, 0] - t0).max(), (ts[:, 1] - ts[:, 0]).mean(), (ts[:, 1] - ts[:, 0]).max(),
             (ts[:, 2] - ts[:, 1]).mean(), (ts[:, 3] - ts[:, 2]).mean(), (ts[:, 3] - ts[:, 2]).max(), ts[:, 3].max() - t0))
    order = np.argsort(ts[:, 0])
    q = [0, nblk // 4, nblk // 2, 3 * nblk // 4, nblk - 1]
    print("entry time quantiles (us):", ["%.1f" % (ts[order[i], 0] - t0) for i in q],
          " exit time quantiles:", ["%.1f" % (np.sort(ts[:, 3])[i] - t0) for i in q])
    # where did the slow workgroups run?  XCC id (HW_REG_XCC_ID) and SE / CU id (HW_REG_HW_ID bits 15:13 / 11:8)
    xcc, hwid = hw >> 16, hw & 0xffff
    se, cu = (hwid >> 13) & 7, (hwid >> 8) & 15
    dur = ts[:, 3] - ts[:, 0]
    print("exit-entry per XCC: " + "  ".join("x%d: n=%d mean %.1f max %.1f" % (x, (xcc == x).sum(), dur[xcc == x].mean(), dur[xcc == x].max())
                                               for x in np.unique(xcc)))
    key = xcc * 1000 + se * 16 + cu
    per_cu = {}
    for k, d_ in zip(key, dur):
        per_cu.setdefault(int(k), []).append(d_)
    cnt = np.array([len(v) for v in per_cu.values()])
    mx = np.array([max(v) for v in per_cu.values()])
    print("distinct CUs %d; workgroups per CU: %s; mean of the per-CU max duration by count: %s"
          % (len(per_cu), dict(zip(*np.unique(cnt, return_counts=True))),
             {int(c): round(float(mx[cnt == c].mean()), 1) for c in np.unique(cnt)}))
