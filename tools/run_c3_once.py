"""Developer aid for profiling: a few c3 evaluations, nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_feature_scatters, WORKLOADS
from sqfa_amd import _native
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C, D, K, model = WORKLOADS[wl]
S, scale = make_feature_scatters(C, D, K, model, torch.device("cuda:0"))
P = C * (C - 1) // 2
for _ in range(n):
    out = _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                   shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
torch.cuda.synchronize()
print(out["loss"].item())
