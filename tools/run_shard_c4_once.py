"""Minimal driver for rocprofv3 --kernel-trace --stats: 12 evaluations of tile shard 0/N of the c4 pair stage
(C=1000, m=32; N from argv, default 8) -- the per-kernel split (K0 cholesky, K0b class factors, K1 pair tiles, K2 finalize)
behind profiles/r4_shard_timings_c4.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqfa_amd import _native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S, scale = bench.make_feature_scatters(1000, 2048, 32, "smsqfa", torch.device("cuda"), torch.float32)
for _ in range(12):
    _native.PairwiseLoss.apply(S, scale, 1e-6, True, -1.0 / 499500, (0, n), None)
torch.cuda.synchronize()
