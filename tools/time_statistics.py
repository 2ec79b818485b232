"""Class statistics on the GPU: wall-clock and achieved f32 GEMM rate of
sqfa_amd.statistics.class_statistics for dataset-shaped inputs.  python tools/time_statistics.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sqfa_amd

SHAPES = {"mnist-like": (60000, 784, 10), "cifar100-like": (50000, 3072, 100), "c3-like": (100000, 784, 1000)}

dev = torch.device("cuda:0")
for name, (N, D, C) in SHAPES.items():
    g = torch.Generator(device="cpu").manual_seed(0)
    X = torch.randn(N, D, generator=g).to(dev)
    y = (torch.arange(N) % C)[torch.randperm(N, generator=g)].to(dev)
    for estimator in ("empirical", "oas"):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = sqfa_amd.statistics.class_statistics(X, y, estimator=estimator)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        flops = 2.0 * N * D * D
        print(f"{name} N={N} D={D} C={C} {estimator}: {dt*1e3:.1f} ms, {flops/dt/1e12:.1f} TFLOP/s of covariance GEMMs", flush=True)
    del X, y, st
