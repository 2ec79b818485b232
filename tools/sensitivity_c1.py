"""Sensitivity of the reference's fixed-step LBFGS on the c1 shape (C=10, D=784, K=4, SQFA, float32):
final loss and epoch count for rounding-level perturbations of the PCA initialisation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_benchmark as fb
import torch, sqfa_amd
dev = torch.device("cuda:0")
fb.warm_up()
st = fb.stats(10, 784, dev)
for trial in range(8):
    model = sqfa_amd.model.SQFA(n_dim=784, n_filters=4, feature_noise=0.01).to(dev)
    model.fit_pca(data_statistics=st)
    if trial:
        g = torch.Generator(device="cpu").manual_seed(trial)
        with torch.no_grad():
            F = model.filters.detach().clone()
            F = F * (1 + 1e-6 * torch.randn(F.shape, generator=g).to(dev))
            model._replace_filters(F)
    loss, t = model.fit(data_statistics=st, max_epochs=300, show_progress=False, return_loss=True)
    print(f"trial {trial}: {len(loss)} epochs, final loss {loss[-1].item():.6f}", flush=True)
