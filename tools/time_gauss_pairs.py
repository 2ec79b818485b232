"""Developer aid: the native Gaussian pair kernel (sqfa_gauss_pair_terms, behind bhattacharyya /
mahalanobis / hellinger / fisher_rao_same_cov) against the torch expression of the reference it replaces
(src/sqfa/distances.py:240-432: batched solve + logdet on the (C,C,K,K) mean-covariance tensor), forward + backward of
bhattacharyya at C classes, K filters, on the GPU.    python tools/time_gauss_pairs.py [C K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd import distances

C, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 16)
dev = torch.device("cuda:0")

def stats(dtype):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(C, 4 * K, K, generator=g, dtype=torch.float64)
    cov = (X.transpose(1, 2) @ X / (4 * K) + 0.05 * torch.eye(K, dtype=torch.float64)).to(dev, dtype).requires_grad_(True)
    mu = (0.3 * torch.randn(C, K, generator=g, dtype=torch.float64)).to(dev, dtype).requires_grad_(True)
    return {"means": mu, "covariances": cov}

def torch_expression(a, b):
    covA, covB, muA, muB = a["covariances"], b["covariances"], a["means"], b["means"]
    mid = 0.5 * (covA[:, None] + covB[None])
    delta = muA[:, None] - muB[None]
    sol = torch.linalg.solve(mid, delta.unsqueeze(-1)).squeeze(-1)
    Q, LD = (delta * sol).sum(-1), torch.logdet(mid)
    return Q / 8 + 0.5 * (LD - 0.5 * (torch.logdet(covA)[:, None] + torch.logdet(covB)[None]))

def timed(fn, st, reps):
    outs = None
    for phase in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps if phase else 2):
            D = fn(st, st)
            g = torch.autograd.grad(D.sum(), [st["covariances"], st["means"]])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (reps if phase else 2)
        outs = (D.detach(), g[0].detach())
    return dt * 1e3, outs

for dtype in (torch.float32, torch.float64):
    st = stats(dtype)
    t_nat, (Dn, gn) = timed(distances.bhattacharyya, st, 10)
    try:
        t_ref, (Dr, gr) = timed(torch_expression, st, 3)
        err = ((Dn - Dr).norm() / Dr.norm()).item(), ((gn - gr).norm() / gr.norm()).item()
        print(f"C={C} K={K} {str(dtype)[6:]}: native fwd+bwd {t_nat:.2f} ms; torch expression (batched solve/logdet on the "
              f"(C,C,K,K) tensor) {t_ref:.1f} ms = {t_ref / t_nat:.0f}x; values agree to {err[0]:.1e}, gradients to {err[1]:.1e}", flush=True)
    except RuntimeError as e:
        print(f"C={C} K={K} {str(dtype)[6:]}: native fwd+bwd {t_nat:.2f} ms; torch expression failed: {str(e)[:80]}", flush=True)
