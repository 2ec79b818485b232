"""Developer aid: time the fused loss+grad launch for a few (C, m, dtype) and print sweep counts."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from sqfa_amd import _native, _lib, distances
sys.path.insert(0, "tools")
from jacobi_emulation import baseline_like

def run(C, K, sqfa, dtype, reps=5):
    D = 784 if K <= 16 else 2048
    S = torch.tensor(baseline_like(min(C, 200), D, K, sqfa=sqfa), dtype=dtype)
    if C > S.shape[0]:
        S = S.repeat((C + S.shape[0] - 1) // S.shape[0], 1, 1)[:C]
        S = S * (1 + 0.3 * torch.rand(C, 1, 1, dtype=dtype))  # make the copies distinct
        # perturb
        N = torch.randn(C, S.shape[1], S.shape[1], dtype=dtype) * 0.02
        S = S + N @ N.transpose(1, 2)
    S = S.cuda()
    P = C * (C - 1) // 2
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    lib = _lib.load()
    _native.POLICY["sweep_counter"] = cnt
    f = lambda: _native.hip_pair_backend(S, None, scale=0.5 if sqfa else 1.0, eps=1e-6, sqrt_mode=True, weights=None,
                                         uniform_weight=-1.0 / P, shard=(0, 1), want_loss=True, want_grad=True,
                                         want_dist=False, want_eig=False)
    out = f(); torch.cuda.synchronize()
    c = cnt.tolist()
    _native.POLICY["sweep_counter"] = None
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    m = S.shape[1]
    t = min(ts)
    print(f"C={C} m={m} {str(dtype)[6:]}: {t*1e3:.3f} ms/eval  {1/t:.1f} evals/s  {P/t/1e6:.1f} Mpairs/s  "
          f"nominal {8*C*(C-1)*m**3/t/1e12:.2f} TF ({8*C*(C-1)*m**3/t/157.3e12*100:.1f}% of fp32 peak)  "
          f"avg sweeps {c[0]/max(c[1],1):.2f}  loss {out['loss'].item():.6f} flags {out['nonfinite'].tolist()}", flush=True)

if __name__ == "__main__":
    run(1000, 16, False, torch.float32)
    run(1000, 16, True, torch.float32)
    run(100, 8, False, torch.float32)
    run(1000, 8, False, torch.float32)
    run(1000, 4, False, torch.float32)
    run(1000, 32, False, torch.float32)
    run(1000, 32, True, torch.float32)
    run(1000, 16, False, torch.float64)
    run(300, 32, False, torch.float64)
