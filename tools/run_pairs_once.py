"""Developer aid for rocprofv3 passes: a few fused loss+grad evaluations at the sizes given as
C:K:model[:dtype] arguments (model = smsqfa | sqfa), bench.py's synthetic feature scatters.

    rocprofv3 --kernel-trace --pmc SQ_WAVES ... -d gpurun_out/x -o x --output-format csv -- python3 tools/run_pairs_once.py 1000:16:smsqfa 1000:16:sqfa
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sqfa_amd import _native  # noqa: E402

REPS = int(os.environ.get("SQFA_REPS", "6"))
for spec in sys.argv[1:] or ["1000:16:smsqfa"]:
    parts = spec.split(":")
    C, K, model = int(parts[0]), int(parts[1]), parts[2]
    dtype = torch.float64 if (len(parts) > 3 and parts[3] == "f64") else torch.float32
    D = 784 if K <= 16 else 2048
    S, scale = bench.make_feature_scatters(C, D, K, model, torch.device("cuda"), dtype)
    P = C * (C - 1) // 2
    for _ in range(REPS):
        out = _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None,
                                       uniform_weight=-1.0 / P, shard=(0, 1), want_loss=True, want_grad=True,
                                       want_dist=False, want_eig=False)
    torch.cuda.synchronize()
    print(spec, "loss", out["loss"].item(), "flags", out["nonfinite"].tolist(), flush=True)
