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
    # the same launches replayed from a captured HIP graph (what a rank of a sharded fit / of bench.py --gpus N does):
    # takes the host out of the measurement
    fused = torch.empty(S.numel() + 3, dtype=S.dtype, device="cuda")
    def launch():
        _native.hip_pair_backend(S, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                 shard=(0, n), want_loss=True, want_grad=True, want_dist=False, want_eig=False,
                                 out_loss=fused[0], out_gradA=fused[3:].view(S.shape))
    launch(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / 200
    print(f"shard 0/{n}: {dt*1e3:.3f} ms per evaluation eager ({1.0/dt:.0f}/s per rank), {dg*1e3:.3f} ms replayed from a graph", flush=True)
