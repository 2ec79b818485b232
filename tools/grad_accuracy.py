"""Developer aid: float32 loss / gradient of the fused evaluation against the float64 kernels on the same input
(bench.py's synthetic scatters), for A/B of stop thresholds and factor variants (SQFA_HIP_LIBRARY selects the library).
    python tools/grad_accuracy.py 300:16:smsqfa 300:16:sqfa 300:32:smsqfa"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sqfa_amd import _native  # noqa: E402

for spec in sys.argv[1:] or ["300:16:smsqfa"]:
    parts = spec.split(":")
    C, K, model = int(parts[0]), int(parts[1]), parts[2]
    D = 784 if K <= 16 else 2048
    S64, scale = bench.make_feature_scatters(C, D, K, model, torch.device("cuda"), torch.float64)
    S32 = S64.float()
    P = C * (C - 1) // 2
    outs = []
    for S in (S32.double(), S32):      # the float64 kernels on the float32-rounded input: only the arithmetic differs
        outs.append(_native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                             shard=(0, 1), want_loss=True, want_grad=True, want_dist=True, want_eig=False))
    ref, got = outs
    g64, g32 = ref["gradA"].double(), got["gradA"].double()
    d64, d32 = ref["dist"].double(), got["dist"].double()
    mask = ~torch.eye(C, dtype=torch.bool, device=d64.device)
    print(f"{spec}: loss rel err {abs(got['loss'].item() - ref['loss'].item()) / abs(ref['loss'].item()):.2e}  "
          f"distances max rel err {((d32 - d64).abs() / d64.abs())[mask].max().item():.2e}  "
          f"gradient |.|_F rel err {((g32 - g64).norm() / g64.norm()).item():.2e}  "
          f"worst class {((g32 - g64).flatten(1).norm(dim=1) / g64.flatten(1).norm(dim=1)).max().item():.2e}", flush=True)
