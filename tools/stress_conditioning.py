"""Ad-hoc stress: float32 and float64 kernels on badly conditioned pencils against the float64 /
LAPACK references.  python tools/stress_conditioning.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sqfa_amd import _native

def spd_batch(C, m, cond, rng):
    out = np.empty((C, m, m))
    for c in range(C):
        Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
        ev = np.exp(rng.uniform(-0.5, 0.5, m) * np.log(cond))
        out[c] = (Q * ev) @ Q.T
    return out

def fused(S):
    S = S.clone().requires_grad_(True); C = S.shape[0]; P = C * (C - 1) // 2
    l, fl = _native.PairwiseLoss.apply(S, 1.0, 1e-6, True, -1.0 / P, (0, 1), None); l.backward()
    return l.item(), S.grad, fl.tolist()

rng = np.random.default_rng(0)
for m in (5, 16, 17, 20, 33, 40, 48, 64):
    for cond in (1e2, 1e4, 1e6, 1e8):
        S = torch.tensor(spd_batch(40, m, cond, rng), device="cuda")
        l64, g64, f64 = fused(S)
        l32, g32, f32 = fused(S.float())
        # LAPACK reference for the loss (float64)
        Sc = S.cpu().numpy(); tot = 0.0; C = Sc.shape[0]
        import scipy.linalg as sl
        for i in range(C):
            for j in range(i):
                lam = sl.eigh(Sc[i], Sc[j], eigvals_only=True)
                tot += np.sqrt((np.log(lam) ** 2).sum() + 1e-6)
        ref = -tot / (C * (C - 1) // 2)
        gerr = ((g32.double() - g64).norm() / g64.norm()).item()
        print(f"m={m:2d} cond={cond:.0e}: f64 vs LAPACK loss {abs(l64-ref)/abs(ref):.1e} flags {f64} | f32 vs f64 loss {abs(l32-l64)/abs(l64):.1e} grad {gerr:.1e} flags {f32}", flush=True)
