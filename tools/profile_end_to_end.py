"""Wall-clock of the user-visible steps of a fit on CIFAR-100-shaped statistics (C=100, D=3072,
K=16, SQFA): model construction, fit_pca, fit, transform.  python tools/profile_end_to_end.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_benchmark as fb
import torch
import sqfa_amd

dev = torch.device("cuda:0")
def tick(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print(f"{label:28s} {t1 - t0:7.3f} s", flush=True); return t1

for rep in range(2):
    print(f"--- pass {rep + 1} ({'cold process' if rep == 0 else 'warm'})")
    t = time.perf_counter()
    st = fb.stats(100, 3072, dev); t = tick("synthetic statistics", t)
    model = sqfa_amd.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01).to(dev); t = tick("model construction", t)
    model.fit_pca(data_statistics=st); t = tick("fit_pca", t)
    loss, _ = model.fit(data_statistics=st, max_epochs=10, show_progress=False, return_loss=True); t = tick("fit, 10 epochs", t)
    X = torch.randn(50000, 3072, device=dev)
    Z = model.transform(X); t = tick("transform 50000 points", t)
    ts = model.transform_scatters(st["covariances"]); t = tick("transform_scatters", t)
