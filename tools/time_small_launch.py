"""Developer aid: where does the small-launch lane geometry stop paying?  Pair-kernel time (HIP events inside the library)
of the regular row and of the small-launch row of a padded size, over a range of class counts.
    python tools/time_small_launch.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sqfa_amd import _lib, _native  # noqa: E402

lib = _lib.load()


def kernel_us(S, scale, mode, reps=20):
    _native.POLICY["geometry"] = mode
    P = S.shape[0] * (S.shape[0] - 1) // 2
    f = lambda: _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                         shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    lib.sqfa_airm_profile(1)
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.sqfa_airm_profile_read(ctypes.byref(ms), ctypes.byref(n))
    lib.sqfa_airm_profile(0)
    _native.POLICY["geometry"] = 0
    return ms.value / n.value * 1e3


DTYPE = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
CASES = ((8, "smsqfa"), (16, "smsqfa"), (16, "sqfa")) if DTYPE == torch.float64 else ((4, "sqfa"), (8, "smsqfa"), (8, "sqfa"), (16, "smsqfa"), (16, "sqfa"))
for K, model in CASES:
    for C in (50, 100, 150, 200, 300, 450, 600, 800, 1000):
        S, scale = bench.make_feature_scatters(C, 784, K, model, torch.device("cuda"), DTYPE)
        a, b = kernel_us(S, scale, -1), kernel_us(S, scale, 1)
        print(f"m={S.shape[1]:2d} C={C:4d} ({C*(C-1)//2:7d} pairs): regular row {a:8.1f} us, small-launch row {b:8.1f} us  -> {'small' if b < a else 'regular'}", flush=True)
