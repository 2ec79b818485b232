"""Host LAPACK eigh of fit_pca (kept on the host for sign/degenerate-subspace parity with the
reference) with all hardware threads vs the process' CPU quota, and rocSOLVER for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd.statistics import usable_cpus
for D in (784, 2048, 3072):
    A = torch.randn(D, D); S = A @ A.T / D
    for n in (torch.get_num_threads(), usable_cpus()):
        saved = torch.get_num_threads(); torch.set_num_threads(n)
        t0 = time.perf_counter(); torch.linalg.eigh(S); dt = time.perf_counter() - t0
        torch.set_num_threads(saved)
        print(f"D={D}: host eigh f32, {n} threads: {dt:.3f} s", flush=True)
    if torch.cuda.is_available():
        Sg = S.cuda(); torch.linalg.eigh(Sg); torch.cuda.synchronize()
        t0 = time.perf_counter(); torch.linalg.eigh(Sg); torch.cuda.synchronize()
        print(f"D={D}: rocSOLVER eigh: {time.perf_counter() - t0:.3f} s", flush=True)
