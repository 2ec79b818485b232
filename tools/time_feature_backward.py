"""Developer aid: time sqfa_feature_scatters_backward_ex + the class-group reduction that follows it
(sqfa_sphere_backward) for several numbers of class groups.  python tools/time_feature_backward.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd import _lib

lib = _lib.load()
ptr = lambda t: ctypes.c_void_p(t.data_ptr())


def run(C, D, K, groups_list=(16, 32, 64), reps=30, ldg=None):
    T = torch.randn(C, D, K, device="cuda")
    ldg = ldg or K
    G = torch.randn(C, ldg, ldg, device="cuda"); G = G + G.transpose(1, 2)
    X = torch.randn(K, D, device="cuda"); norms = X.norm(dim=1)
    gout = torch.empty(K, D, device="cuda")
    one = torch.ones(1, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for groups, sym in [(g, s) for g in groups_list for s in (0, 1)]:
        partial = torch.empty(groups, K, D, device="cuda")
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tb = ts = 0.0
        for it in range(reps + 3):
            e[0].record()
            _lib.check(lib.sqfa_feature_scatters_backward_ex(ptr(G), ldg, ptr(T), C, D, K, _lib.SQFA_F32, groups, sym, ptr(partial), stream), "bwd")
            e[1].record()
            _lib.check(lib.sqfa_sphere_backward(ptr(X), ptr(norms), K, D, _lib.SQFA_F32, ptr(partial), groups, None, ptr(one), ptr(gout), stream), "sph")
            e[2].record(); torch.cuda.synchronize()
            if it >= 3:
                tb += e[0].elapsed_time(e[1]); ts += e[1].elapsed_time(e[2])
        print(f"C={C} D={D} K={K} ldg={ldg} groups={groups} symmetric={sym}: backward {tb/reps*1e3:.1f} us  group reduction {ts/reps*1e3:.1f} us", flush=True)


if __name__ == "__main__":
    run(1000, 784, 16); run(1000, 784, 16, ldg=17); run(1000, 2048, 32); run(100, 3072, 16, ldg=17)
