import torch, time
x = torch.empty(3_072_000_000 // 8, dtype=torch.float64, device='cuda').normal_()
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
s = t(lambda: x.sum()); print("sum  (read 3.07 GB): %.3f ms -> %.2f TB/s" % (s*1e3, 3.072e9/s/1e12))
s = t(lambda: torch.amax(x)); print("amax (read 3.07 GB): %.3f ms -> %.2f TB/s" % (s*1e3, 3.072e9/s/1e12))
s = t(lambda: y.copy_(x)); print("copy (read+write 6.1 GB): %.3f ms -> %.2f TB/s" % (s*1e3, 6.144e9/s/1e12))
xi = x.view(torch.int64)
s = t(lambda: xi.sum()); print("int sum: %.3f ms -> %.2f TB/s" % (s*1e3, 3.072e9/s/1e12))
