"""A/B in one session: closure time with the native S = F T / dL/dF kernels vs torch's batched GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import time_closure as tc
from sqfa_amd import _native
import sqfa_amd

def closure_time(C, D, K, model_name, native, reps=60):
    _native.ProjectScatters.NATIVE_PRODUCTS_MIN_CLASSES = 1 if native else 10 ** 9
    mu, cov = tc.make_stats(C, D)
    torch.manual_seed(7)
    if model_name == "sqfa":
        model = sqfa_amd.model.SQFA(n_dim=D, n_filters=K, feature_noise=0.01).cuda(); stats = {"means": mu, "covariances": cov}
    else:
        model = sqfa_amd.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.01).cuda(); stats = cov + mu[:, :, None] * mu[:, None, :]
    prepared = model._prepare_statistics(stats)
    def closure():
        model.zero_grad(); loss, flags = model._fused_closure_loss(prepared); loss.backward(); return loss
    for _ in range(10): closure()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): closure()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best

for cfg in ((1000, 784, 16, "smsqfa"), (1000, 784, 16, "sqfa"), (100, 3072, 16, "sqfa"), (100, 784, 8, "smsqfa")):
    a = closure_time(*cfg, native=False); b = closure_time(*cfg, native=True); a2 = closure_time(*cfg, native=False)
    print(f"{cfg}: torch products {a*1e3:.3f} / {a2*1e3:.3f} ms, native {b*1e3:.3f} ms", flush=True)
