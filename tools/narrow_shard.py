"""The neuron-sharded step of an 8-GPU run on ONE GPU: ll+grad of a 16-neuron shard (one post tile) of C3 against the
whole 640-column feature row (dev tool).   python tools/narrow_shard.py [--f32] [neurons_per_shard ...]
k_fused8 (per-wave block rings); compared, on a 60 s recording, with the whole-population evaluation.  (The A/B against
k_fused6<5,1,1,8,0>, one image buffer, that this tool ran while both existed: profiles/r05_narrow_shard_ab.txt.)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from theano_pyglm_amd import _lib
import bench

shards = [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [16]
F32 = 2 if '--f32' in sys.argv else 0                                # PGL_OPT_FEATURE_F32 = 2: f32 resident blocks (opt-in)
N, dt = 128, 0.001
ib = bench.standard_ibasis()
R, B = ib.shape
P = 1 + N * B
rng = np.random.default_rng(99)
theta = np.zeros((N, P)); theta[:, 0] = 20.0 + 0.1 * rng.standard_normal(N)
theta[:, 1:] = 0.5 * rng.standard_normal((N, N * B))


def run(T, K, check_full):
    S = bench.make_workload(N, T, dt, seed=1234 + 3)
    nT = S.shape[0]
    dev = _lib.DeviceGlm(N, nT, B, R, 'explinear', dt)
    dev.set_spikes(S); dev.set_basis(ib)
    dev.set_option(_lib.OPT_TIMING, 1)
    if F32:
        dev.set_option(_lib.OPT_FEATURE_F32, F32)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    dev.set_stream(st.cuda_stream)
    d_W = torch.ones((N, N), dtype=torch.float64, device='cuda')
    full = None
    if check_full:
        d_theta = torch.from_numpy(theta).cuda()
        d_out = torch.zeros(N * (1 + P), dtype=torch.float64, device='cuda')
        dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_out[:N].data_ptr(), d_out[N:].data_ptr())
        torch.cuda.synchronize()
        full = (d_out[:N].cpu().numpy(), d_out[N:].view(N, P).cpu().numpy())
    for M in shards:
        n_lo = 32
        d_theta = torch.from_numpy(theta[n_lo:n_lo + M]).cuda()
        res = {}
        for name in (" + ".join(_lib.plan_kernels(N, B=B, R=R, nT=nT, n_lo=n_lo, count=M, opt_f32=F32)),):
            d_out = torch.zeros(M * (1 + P), dtype=torch.float64, device='cuda')
            info = dev.info(n_lo, n_lo + M)
            for _ in range(5):
                dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_out[:M].data_ptr(), d_out[M:].data_ptr(),
                                n_lo, n_lo + M)
            torch.cuda.synchronize()
            dev.timing_summary(reset=True)
            t0 = time.perf_counter()
            for _ in range(K):
                dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_out[:M].data_ptr(), d_out[M:].data_ptr(),
                                n_lo, n_lo + M)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / K * 1e3
            n, fused, total = dev.timing_summary(reset=True)
            ll, g = d_out[:M].cpu().numpy(), d_out[M:].view(M, P).cpu().numpy()
            res[name] = (ll, g)
            gb = info.get('resident_feature_bytes', float('nan')) / 1e9
            print("T=%g s, %d neurons, %s: step %.3f ms (fused %.3f, prep+fused+finalize %.3f), kernel version %s, "
                  "%.2f GB of tiles -> %.2f TB/s" % (T, M, name, wall, fused, total, info.get('kernel_version'), gb,
                                                      gb / fused if fused else float('nan')), flush=True)
            if full is not None:
                print("    vs the whole-population evaluation: ll %.1e, grad %.1e (relative to max |g|)"
                      % (np.max(np.abs(ll - full[0][n_lo:n_lo + M]) / np.abs(ll)),
                         np.max(np.abs(g - full[1][n_lo:n_lo + M])) / np.max(np.abs(g))))
    dev.close()


# (timing ablation of the tile loop: build variants with tools/build_variant.sh abl<bits> -DPGL_F8_ABL=<bits> -- 1 no MFMAs,
#  2 no rate epilogue, 4 no barriers, 8 no fragment reads, 16 no requests inside the loop, 32 no spike-count loads -- and run
#  this tool with PYGLM_HIP_LIB=theano_pyglm_amd/libpyglm_hip_abl<bits>.so; results: profiles/r05_narrow_shard_ab.txt)
run(60.0, 20, True)
run(600.0, 100, False)
