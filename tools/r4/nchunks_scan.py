"""Headline workload (C3) against the number of time chunks (= workgroups) of the two-pass kernel: does a stream of
shorter workgroups run the same tiles faster than one long-lived workgroup per CU?  (dev tool)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
N = 128
p = H.Problem(N, 600000, H.std_ibasis(), kind='explinear', seed=1234, w_scale=0.5)
dev = p.device()
st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
dev.set_option(_lib.OPT_TIMING, 1)
d_theta = torch.from_numpy(p.theta).cuda(); d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
d_ll = torch.zeros(N, dtype=torch.float64, device='cuda'); d_g = torch.zeros((N, p.P), dtype=torch.float64, device='cuda')
for nch in (0, 256, 384, 512, 768, 1024, 1536, 2048, 255, 250, 240):
    dev.set_option(_lib.OPT_NCHUNKS, nch)
    for _ in range(5):
        dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize(); dev.timing_summary(reset=True)
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    n, fused, total = dev.timing_summary(reset=True)
    print("nchunks %4d: fused kernels %.3f ms, whole evaluation %.3f ms back to back" % (nch, fused, wall))
