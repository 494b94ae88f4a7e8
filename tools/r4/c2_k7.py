"""C2 through k_fused7 (forced) against the default k_fused6 (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from tests import helpers as H
from theano_pyglm_amd import _lib
p = H.Problem(32, 300000, H.std_ibasis(), seed=1234, w_scale=0.5)
for kern in (0, 7, 8, 0, 7, 8):
    dev = p.device()
    dev.set_option(_lib.OPT_KERNEL, kern)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
    dev.set_option(_lib.OPT_TIMING, 1)
    d_theta = torch.from_numpy(p.theta).cuda(); d_W = torch.from_numpy(np.ascontiguousarray(p.Weff)).cuda()
    d_ll = torch.zeros(32, dtype=torch.float64, device='cuda'); d_g = torch.zeros((32, p.P), dtype=torch.float64, device='cuda')
    for _ in range(20): dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize(); dev.timing_summary(reset=True)
    for _ in range(100): dev.ll_grad_dev(d_theta.data_ptr(), d_W.data_ptr(), d_ll.data_ptr(), d_g.data_ptr())
    torch.cuda.synchronize()
    n, fused, total = dev.timing_summary(reset=True)
    i = dev.info()
    print("kernel option %d -> version %d, blocks %d threads %d lds %d: fused %.4f ms, evaluation %.4f ms" % (kern, i['kernel_version'], i['blocks'], i['threads'], i['lds_bytes'], fused, total))
    dev.close()
