"""Developer aid: which pairs differ between the float32 and float64 m=17 kernels for one random draw."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, numpy as np
import model_cases as mc
from sqfa_amd import distances
DEV = "cuda:0"
stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=300, D=64).items()}
for seed in (36, 346):
    g = torch.Generator(device="cpu").manual_seed(seed)
    F = torch.randn(16, 64, generator=g, dtype=torch.float64).to(DEV)
    F = F / F.norm(dim=1, keepdim=True)
    S = torch.einsum("kd,cde,le->ckl", F, stats["covariances"].double(), F) + 0.01 * torch.eye(16, device=DEV, dtype=torch.float64)
    mu = stats["means"].double() @ F.T
    E = torch.zeros(300, 17, 17, dtype=torch.float64, device=DEV)
    E[:, :16, :16] = S + mu[:, :, None] * mu[:, None, :]; E[:, :16, 16] = mu; E[:, 16, :16] = mu; E[:, 16, 16] = 1
    d64 = distances.affine_invariant(E, E)
    d32 = distances.affine_invariant(E.float(), E.float()).double()
    err = (d32 - d64).abs()
    idx = torch.nonzero(err > 1e-4)
    print("seed", seed, "pairs off by > 1e-4:", idx.shape[0], "max", err.max().item())
    for (i, j) in idx[:12].tolist():
        print("   i", i, "j", j, "i%16", i % 16, "j%8", j % 8, "d64", d64[i, j].item(), "d32", d32[i, j].item())
    lam = torch.linalg.eigvalsh(E)
    print("   cond range", (lam[:, -1] / lam[:, 0]).min().item(), (lam[:, -1] / lam[:, 0]).max().item())
