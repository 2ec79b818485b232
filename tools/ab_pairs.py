"""A/B of pre-built library variants on bench.py's own synthetic inputs (SURVEY.md 8d generator), one process per
variant:  python tools/ab_pairs.py C:K:model[:dtype] lib1.so lib2.so ...   (model = smsqfa | sqfa; "-" = installed library)
Prints, per variant: pair kernel ms per launch (HIP events around the kernel, sqfa_airm_profile; median of the
runs' averages), whole evaluation ms (K0 + K0b + K1 + K2, events around the call), sweeps per wave round, loss."""
import os
import subprocess
import sys

CHILD = r'''
import ctypes, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
import bench
from sqfa_amd import _lib, _native
spec = sys.argv[1].split(":")
C, K, model = int(spec[0]), int(spec[1]), spec[2]
dtype = torch.float64 if (len(spec) > 3 and spec[3] == "f64") else torch.float32
D = 784 if K <= 16 else 2048
S, scale = bench.make_feature_scatters(C, D, K, model, torch.device("cuda"), dtype)
P = C * (C - 1) // 2
lib = _lib.load()
def f():
    return _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                    shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
with _native.policies(sweep_counter=cnt):
    out = f(); torch.cuda.synchronize(); c = cnt.tolist()
reps = int(os.environ.get("SQFA_REPS", "30" if S.shape[1] <= 20 else "8"))
for _ in range(max(3, reps // 2)): f()
torch.cuda.synchronize()
ks, es = [], []
for rnd in range(5):
    lib.sqfa_airm_profile(1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): out = f()
    b.record(); torch.cuda.synchronize()
    lib.sqfa_airm_profile(0)
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_airm_profile_read(ctypes.byref(ms), ctypes.byref(n))
    ks.append(ms.value / max(n.value, 1)); es.append(a.elapsed_time(b) / reps)
m = S.shape[1]
k = statistics.median(ks)
print(f"m={m} {str(dtype)[6:]} kernel {k:.4f} ms (min {min(ks):.4f})  eval {statistics.median(es):.4f} ms  nominal {8*C*(C-1)*m**3/k/1e9/157.3:.3f}  "
      f"sweeps {c[0]/max(c[1],1):.2f}  loss {out['loss'].item():.7f} flags {out['nonfinite'].tolist()}")
'''

spec = sys.argv[1]
for lib in sys.argv[2:]:
    env = dict(os.environ)
    if lib != "-":
        env["SQFA_HIP_LIBRARY"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CHILD, spec], capture_output=True, text=True, env=env)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr[-400:].replace("\n", " | ")
    print(f"{os.path.basename(lib):28s} {line}", flush=True)
