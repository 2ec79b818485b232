"""Developer aid: bhattacharyya forward + backward through the native Gaussian pair kernels for several (C, K, dtype)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd import distances
dev = torch.device("cuda:0")
for C, K, dtype in ((1000, 16, torch.float32), (1000, 12, torch.float32), (1000, 8, torch.float32), (1000, 4, torch.float32), (1000, 15, torch.float32),
                    (1000, 8, torch.float64), (1000, 4, torch.float64), (1000, 16, torch.float64), (100, 16, torch.float32), (1000, 32, torch.float32)):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(C, 4 * K, K, generator=g, dtype=torch.float64)
    cov = (X.transpose(1, 2) @ X / (4 * K) + 0.05 * torch.eye(K, dtype=torch.float64)).to(dev, dtype).requires_grad_(True)
    mu = (0.3 * torch.randn(C, K, generator=g, dtype=torch.float64)).to(dev, dtype).requires_grad_(True)
    st = {"means": mu, "covariances": cov}
    best = None
    for phase in range(6):   # batch 0 warms up; the minimum over the batches is reported: the loop is host-bound at small K and
        n = 2 if phase == 0 else 6   # the host stalls for milliseconds now and then
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            D = distances.bhattacharyya(st, st)
            torch.autograd.grad(D.sum(), [cov, mu])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        if phase:
            best = dt if best is None else min(best, dt)
    print(f"C={C} K={K} {str(dtype)[6:]}: bhattacharyya fwd+bwd {best*1e3:.3f} ms (best of 5 batches of 6)", flush=True)
