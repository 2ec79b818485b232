"""Developer aid: can the HBM-bound projection kernel hide behind the VALU-bound pair kernel?
Both are launched on their own data, back to back on one stream and concurrently on two streams; if the
concurrent time is about the sum, the kernels only time-slice (the pair kernel's workgroups fill the LDS and
the register file of every CU) and a class-chunked projection/pair pipeline inside one closure cannot gain.
    python tools/overlap_probe.py [c3|c4]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqfa_amd import _lib, _native

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
C, D, K, model = bench.WORKLOADS[which]
dev = torch.device("cuda:0")
lib = _lib.load()
S, scale = bench.make_feature_scatters(C, D, K, model, dev)
Psi = torch.randn(C, D, D, device=dev)
F = torch.randn(K, D, device=dev)
T = torch.empty(C, D, K, device=dev)
P = C * (C - 1) // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def pairs():
    return _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                    shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)

def project():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st), "project")

def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def sequential():
    project(); pairs()

def concurrent(first_pairs):
    def run():
        order = [(s1, pairs), (s2, project)] if first_pairs else [(s2, project), (s1, pairs)]
        for st, fn in order:
            with torch.cuda.stream(st):
                fn()
    return run

tp, tq = timed(project), timed(pairs)
ts = timed(sequential)
tc1, tc2 = timed(concurrent(True)), timed(concurrent(False))
print(f"{which}: projection alone {tp:.3f} ms, pair evaluation alone {tq:.3f} ms, back to back on one stream {ts:.3f} ms, "
      f"two streams (pairs launched first) {tc1:.3f} ms, two streams (projection launched first) {tc2:.3f} ms", flush=True)
