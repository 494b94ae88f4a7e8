"""Host and device cost of the optimizer's two rocBLAS calls per launch at the C3 and the C5-stress row length (dev)."""
import time, torch
for M, P in ((128, 641), (64, 1220), (64, 199)):
    dev = 'cuda'
    H = torch.eye(P, dtype=torch.float64, device=dev).repeat(M, 1, 1)
    g = torch.randn(M, P, dtype=torch.float64, device=dev)
    t = torch.empty(M, P, dtype=torch.float64, device=dev)
    U = torch.randn(M, P, 3, dtype=torch.float64, device=dev) * 1e-3
    V = torch.randn(M, P, 3, dtype=torch.float64, device=dev) * 1e-3
    for name, fn in (("bmm(H, g)", lambda: torch.bmm(H, g.unsqueeze(2), out=t.unsqueeze(2))),
                     ("H.baddbmm_(U, V^T)", lambda: H.baddbmm_(U, V.transpose(1, 2))),
                     ("einsum H g", lambda: torch.einsum('mij,mj->mi', H, g)),
                     ("matmul (H @ g[...,None])", lambda: torch.matmul(H, g.unsqueeze(2)))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): fn()
        host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 50
        print("M=%d P=%d %-28s host %.3f ms per call, with device %.3f ms" % (M, P, name, host * 1e3, tot * 1e3))
