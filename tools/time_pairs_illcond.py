"""Developer aid: the fused evaluation on UNRELATED ill-conditioned SPD classes (random orthogonal eigenvectors, eigenvalues
log-uniform over a condition number) -- the other end of the input range from bench.py's similar classes: ms per
evaluation and sweeps per wave.      python tools/time_pairs_illcond.py [C] [cond]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sqfa_amd import _native, _lib

C = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cond = float(sys.argv[2]) if len(sys.argv) > 2 else 1e3
lib = _lib.load()
for m in (12, 16, 17, 24, 32, 33):
    g = torch.Generator().manual_seed(m)
    Q, _ = torch.linalg.qr(torch.randn(C, m, m, generator=g, dtype=torch.float64))
    ev = torch.exp((torch.rand(C, m, generator=g, dtype=torch.float64) - 0.5) * torch.log(torch.tensor(cond)))
    S = ((Q * ev[:, None, :]) @ Q.transpose(1, 2)).float().cuda()
    S = 0.5 * (S + S.transpose(1, 2))
    P = C * (C - 1) // 2
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    f = lambda: _native.hip_pair_backend(S, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                         shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
    with _native.policies(sweep_counter=cnt):
        out = f(); torch.cuda.synchronize(); c = cnt.tolist()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"C={C} m={m} cond {cond:g}: {min(ts)*1e3:.3f} ms/eval  avg sweeps {c[0]/max(c[1],1):.2f}  loss {out['loss'].item():.6f} flags {out['nonfinite'].tolist()}", flush=True)
