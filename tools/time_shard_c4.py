"""What ONE rank of an N-GPU job does for BASELINE config 4 (C=1000, n_dim=2048, n_filters=32) between its collectives,
measured on one GPU for shard 0 of N = 1, 2, 4, 8 (no xGMI involved):

  pair stage alone (scaling_c4_pairs of bench.py): K0 + K0b + K1 + K2 of tile shard (0, N) on the replicated S (C,32,32)
  class-sharded closure (scaling_c4_closure):      stage A1  sphere + projection of the rank's C/N classes (16.8 GB / N)
                                                   stage A2  pair-tile shard (0, N) on the gathered S
                                                   stage B   backward of the local classes to the raw filters
Every stage is replayed from its own captured HIP graph (as sqfa_amd._optim.ShardedClosure runs it); the collectives
between the stages (all-gather of S slices 4.1 MB, all-reduce of [loss, flags, dL/dS] 4.1 MB, all-reduce of dL/dF 262 KB)
are NOT in these numbers.   python tools/time_shard_c4.py [C]      (C: class count, default 1000)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sqfa_amd import _native

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D, K = 2048, 32
dev = torch.device("cuda")
P = C * (C - 1) // 2
REPS = 20


def graph_ms(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


# ---- pair stage alone -----------------------------------------------------------------------------------------
S, scale = bench.make_feature_scatters(C, D, K, "smsqfa", dev, torch.float32)
print(f"c4 pair stage, C={C}, m={S.shape[1]}: per-rank K0 + K0b + K1 + K2 of tile shard (0, N), replayed from a graph")
base = None
for n in (1, 2, 4, 8):
    fused = torch.empty(S.numel() + 3, dtype=S.dtype, device=dev)

    def launch():
        _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                 shard=(0, n), want_loss=True, want_grad=True, want_dist=False, want_eig=False,
                                 out_loss=fused[0], out_gradA=fused[3:].view(S.shape))
    ms = graph_ms(launch)
    base = base or ms
    print(f"  shard 0/{n}: {ms:8.3f} ms  -> {base / ms:5.2f}x of one GPU before the all-reduce of {fused.numel() * 4 / 1e6:.1f} MB", flush=True)

# ---- class-sharded closure ------------------------------------------------------------------------------------
torch.manual_seed(7)
raw = torch.randn(K, D, device=dev)
noise = 0.01
print(f"c4 closure, class-sharded: stages of rank 0 of N (its {C}/N classes of the (C,{D},{D}) statistics), each replayed from a graph")
S_full = torch.empty((C, K, K), device=dev)
for lo in range(0, C, 125):                 # the gathered S of all classes, once (what the all-gather delivers)
    hi = min(C, lo + 125)
    st = _native.closure_stage_project(raw, bench.make_class_shard_statistics(C, D, lo, hi, dev), None, noise, True)
    S_full[lo:hi].copy_(st["S"])
    del st
torch.cuda.synchronize()
tot1 = None
for n in (1, 2, 4, 8):
    hi = C // n
    local = bench.make_class_shard_statistics(C, D, 0, hi, dev)
    fused = torch.empty(C * K * K + 3, dtype=torch.float32, device=dev)
    box = {}

    def a1():
        box["st"] = _native.closure_stage_project(raw, local, None, noise, True)

    def a2():
        _native.closure_stage_pairs(S_full, 1.0, True, -1.0 / P, (0, n), fused)

    def b():
        box["grad"] = _native.closure_stage_backward(box["st"], fused[3:].view(C, K, K)[:hi], None)

    t_a1 = graph_ms(a1, 10)
    t_a2 = graph_ms(a2)
    t_b = graph_ms(b)
    tot = t_a1 + t_a2 + t_b
    tot1 = tot1 or tot
    gbs = 4.0 * hi * D * D / (t_a1 * 1e-3) / 1e9
    print(f"  rank 0/{n}: projection of {hi} classes {t_a1:7.3f} ms ({gbs:6.0f} GB/s incl. sphere + S = F T), pair shard {t_a2:7.3f} ms, "
          f"backward {t_b:6.3f} ms, sum {tot:7.3f} ms -> {tot1 / tot:5.2f}x of one GPU before the three collectives", flush=True)
    del local, box
    torch.cuda.empty_cache()
