"""Why is the c4 pair kernel 5-6 % slower inside the closure than alone (bench.py: scaling_c4_closure.pair_kernel_ms 8.0 vs
scaling_c4_pairs.pair_kernel_ms 7.6)?  Same session, HIP events around the pair kernel (sqfa_airm_profile), sweeps per wave:
  A  bench.py's pair-stage input (make_feature_scatters), launches back to back
  B  the closure's own input (the S that the projection of the class-sharded generator's statistics produces), back to back
  C  input B, each launch preceded by the 16.8 GB projection launch (what the closure does)
  D  input B, each launch preceded by 5 ms of idle GPU
  E  input B, each launch preceded by a 4.2 GB device-to-device copy (HBM traffic without MFMA work)
    python tools/c4_pair_in_closure_probe.py [C]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sqfa_amd import _lib, _native

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D, K = 2048, 32
dev = torch.device("cuda")
lib = _lib.load()
P = C * (C - 1) // 2

S_bench, scale = bench.make_feature_scatters(C, D, K, "smsqfa", dev, torch.float32)
stats = bench.make_class_shard_statistics(C, D, 0, C, dev)
torch.manual_seed(7)
import sqfa_amd
model = sqfa_amd.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.01).to(dev)
raw = model.parametrizations.filters.original.detach()
st = _native.closure_stage_project(raw, stats, None, 0.01, True)
S_clos = st["S"].clone()
big = torch.empty(1 << 30, dtype=torch.float32, device=dev)   # 4.3 GB copy source
big2 = torch.empty_like(big)


def pair(S):
    return _native.hip_pair_backend(S, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                    shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)


def sweeps(S):
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    with _native.policies(sweep_counter=cnt):
        pair(S); torch.cuda.synchronize()
    c = cnt.tolist()
    return c[0] / max(c[1], 1)


def timed(S, before, reps=12):
    for _ in range(3):
        before(); pair(S)
    torch.cuda.synchronize()
    lib.sqfa_airm_profile(1)
    for _ in range(reps):
        before(); pair(S)
    torch.cuda.synchronize()
    lib.sqfa_airm_profile(0)
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_airm_profile_read(ctypes.byref(ms), ctypes.byref(n))
    pm, pn = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_project_profile_read(ctypes.byref(pm), ctypes.byref(pn))
    return ms.value / max(n.value, 1)


def idle():
    torch.cuda.synchronize(); time.sleep(0.005)


nothing = lambda: None
project = lambda: _native.closure_stage_project(raw, stats, None, 0.01, True)
copy = lambda: big2.copy_(big)
print(f"C={C}: sweeps per wave round  A (bench input) {sweeps(S_bench):.3f}   B (closure input) {sweeps(S_clos):.3f}")
for rnd in range(2):
    print(f"round {rnd}:  A alone {timed(S_bench, nothing):.3f} ms | B alone {timed(S_clos, nothing):.3f} | C after the projection {timed(S_clos, project):.3f} | "
          f"D after 5 ms idle {timed(S_clos, idle):.3f} | E after a 4.3 GB copy {timed(S_clos, copy):.3f} | A after the projection {timed(S_bench, project):.3f}", flush=True)
