"""Minimal driver for rocprofv3: 30 fused evaluations of tile shard 0/N of c3 (N from argv, default 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sqfa_amd import _native
from jacobi_emulation import baseline_like
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = torch.tensor(baseline_like(1000, 784, 16), dtype=torch.float32, device="cuda")
for _ in range(130):
    _native.PairwiseLoss.apply(S, 1.0, 1e-6, True, -1.0 / 499500, (0, n), None)
torch.cuda.synchronize()
