"""T = Psi F^T from the full (C,D,D) tensor (sqfa_project_scatters) vs from the block-triangular packed statistics
(sqfa_project_scatters_packed): ms per launch (HIP events around 30 launches, alternating), GB/s against each kernel's own
algorithmic bytes, and the fraction of the 8 TB/s HBM roofline.    python tools/time_projection_packed.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sqfa_amd import _lib, _native

lib = _lib.load()


def run(C, D, K, reps=30):
    Psi = torch.randn(C, D, D, device="cuda")
    Psi = (Psi + Psi.transpose(1, 2)).contiguous()
    F = torch.randn(K, D, device="cuda")
    T = torch.empty(C, D, K, device="cuda")
    Tp = torch.empty_like(T)
    packed = _native.pack_scatters(Psi)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = lambda: lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    pk = lambda: lib.sqfa_project_scatters_packed(F.data_ptr(), K, D, packed.data_ptr(), C, 0, Tp.data_ptr(), st)
    res = {"full": [], "packed": []}
    for rnd in range(3):
        for name, fn in (("full", full), ("packed", pk)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / reps)
    a, b = min(res["full"]), min(res["packed"])
    diff = ((Tp - T).norm() / T.norm()).item()
    bf, bp = 4.0 * C * D * D, 4.0 * C * packed.shape[1]
    print(f"C={C:4d} D={D:4d} K={K:2d}: full {a:.3f} ms = {bf / a / 1e6:5.0f} GB/s ({bf / a / 1e6 / 8000:.2f} of 8 TB/s) | packed {b:.3f} ms = "
          f"{bp / b / 1e6:5.0f} GB/s ({bp / b / 1e6 / 8000:.2f}) of {bp / bf:.3f} x the bytes | speed-up {a / b:.2f}x  (packed vs full {diff:.1e})", flush=True)


run(1000, 784, 16); run(1000, 2048, 32); run(100, 3072, 16); run(1000, 784, 8); run(1000, 1024, 16); run(1000, 512, 8)
run(300, 1008, 20); run(1000, 2048, 16); run(500, 3072, 16); run(256, 784, 16)
