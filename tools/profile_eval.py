"""cProfile of the host side of one fused loss+grad evaluation (what bounds an 8-GPU rank once its
GPU work is down to ~0.12 ms).  python tools/profile_eval.py [shard_count]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sqfa_amd import _native
from jacobi_emulation import baseline_like

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = torch.tensor(baseline_like(1000, 784, 16), dtype=torch.float32, device="cuda")
P = 1000 * 999 // 2
for _ in range(50):
    _native.PairwiseLoss.apply(S, 1.0, 1e-6, True, -1.0 / P, (0, n), None)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(1000):
    _native.PairwiseLoss.apply(S, 1.0, 1e-6, True, -1.0 / P, (0, n), None)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18); print(out.getvalue())
