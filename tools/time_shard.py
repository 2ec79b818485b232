"""Per-rank cost of the sharded evaluation without the collective: times the fused loss+grad launch
for tile shard (0, N) of the c3 workload on one GPU, N = 1, 2, 4, 8 (what a rank of an N-GPU job does
between two all-reduces).  python tools/time_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sqfa_amd import _native
from jacobi_emulation import baseline_like

S = torch.tensor(baseline_like(1000, 784, 16), dtype=torch.float32, device="cuda")
P = 1000 * 999 // 2
for n in (1, 2, 4, 8):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            _native.PairwiseLoss.apply(S, 1.0, 1e-6, True, -1.0 / P, (0, n), None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"shard 0/{n}: {dt*1e3:.3f} ms per evaluation ({1.0/dt:.0f}/s per rank)", flush=True)
