"""Developer aid: c3 with and without gradient, for per-kernel timing under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_feature_scatters, WORKLOADS
from sqfa_amd import _native
C, D, K, model = WORKLOADS["c3"]
S, scale = make_feature_scatters(C, D, K, model, torch.device("cuda:0"))
P = C * (C - 1) // 2
for grad in (True, False, True, False, True, False):
    out = _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                   shard=(0, 1), want_loss=True, want_grad=grad, want_dist=False, want_eig=False)
    torch.cuda.synchronize()
print(out["loss"].item())
