"""Phase ablation of the pair kernel at c3 size: forward only vs forward+backward, for the library
given in argv[1] (build one with -DSQFA_MAX_SWEEPS=0 to take the sweeps out).  Prints ms per launch
of the whole evaluation (K0 + K1 + K2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["SQFA_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])   # read by sqfa_amd/_lib.py at import
import torch
from sqfa_amd import _native
from jacobi_emulation import baseline_like
S = torch.tensor(baseline_like(1000, 784, 16), dtype=torch.float32, device="cuda")
for want_grad in (False, True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            _native.hip_pair_backend(S, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-2e-6,
                                     shard=(0, 1), want_loss=True, want_grad=want_grad, want_dist=False, want_eig=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'installed'}: want_grad={want_grad}: {dt*1e3:.3f} ms", flush=True)
