"""Developer aid: time sqfa_project_scatters alone (HBM GB/s) and the fused forward+backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd import _native

def run(C, D, K, reps=20):
    Psi = torch.randn(C, D, D, device="cuda")
    F = torch.randn(K, D, device="cuda", requires_grad=True)
    G = torch.randn(C, K, K, device="cuda")
    for _ in range(3):
        S = _native.ProjectScatters.apply(F, Psi)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for _ in range(reps):
        e0.record(); S = _native.ProjectScatters.apply(F, Psi); e1.record()
        (g,) = torch.autograd.grad(S, F, G); e2.record(); torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    tf /= reps; tb /= reps
    byts = 4.0 * C * D * D
    print(f"C={C} D={D} K={K}: forward {tf:.3f} ms = {byts/tf/1e6:.0f} GB/s ({byts/tf/1e6/8000*100:.0f}% of 8 TB/s), backward {tb:.3f} ms", flush=True)

if __name__ == "__main__":
    run(1000, 784, 16); run(100, 784, 8); run(300, 2048, 32); run(100, 3072, 16); run(1000, 784, 4); run(400, 1024, 64); run(10, 784, 4); run(50, 784, 8); run(200, 784, 16); run(30, 3072, 16)
