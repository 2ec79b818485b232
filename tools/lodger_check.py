"""Developer aid: float32 m=17 pair kernel against the float64 kernel over many random filter draws (C=300, D=64, K=16
SQFA embedding), for kernel variants.  python tools/lodger_check.py lib1.so lib2.so"""
import sys, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, numpy as np
if sys.argv[1] == "child":
    import model_cases as mc
    from sqfa_amd import _native
    DEV = "cuda:0"
    stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=300, D=64).items()}
    P = 300 * 299 // 2
    worst = (0, -1)
    for seed in range(400):
        g = torch.Generator(device="cpu").manual_seed(seed)
        F = torch.randn(16, 64, generator=g, dtype=torch.float64).to(DEV)
        F = F / F.norm(dim=1, keepdim=True)
        S = torch.einsum("kd,cde,le->ckl", F, stats["covariances"].double(), F) + 0.01 * torch.eye(16, device=DEV, dtype=torch.float64)
        mu = stats["means"].double() @ F.T
        E = torch.zeros(300, 17, 17, dtype=torch.float64, device=DEV)
        E[:, :16, :16] = S + mu[:, :, None] * mu[:, None, :]; E[:, :16, 16] = mu; E[:, 16, :16] = mu; E[:, 16, 16] = 1
        l64, _ = _native.PairwiseLoss.apply(E, 0.5, 1e-6, True, -1.0 / P, (0, 1), None)
        l32, _ = _native.PairwiseLoss.apply(E.float(), 0.5, 1e-6, True, -1.0 / P, (0, 1), None)
        rel = abs(l32.item() - l64.item()) / abs(l64.item())
        if rel > worst[0]:
            worst = (rel, seed)
        if rel > 2e-6:
            print("  seed", seed, "rel", rel, flush=True)
    print("worst", worst, flush=True)
else:
    for lib in sys.argv[1:]:
        print(lib, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SQFA_HIP_LIBRARY=os.path.abspath(lib)))
