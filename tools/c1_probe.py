"""Developer aid: does the c1 (K=4) fit follow the reference when started from the reference's exact initial filters?"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, model_cases as mc
from conftest import load_golden, rel_err
G = load_golden("g6b_fit_c1.npz")
DEV = torch.device("cuda:0")
stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=10, D=784).items()}
for exact_init in (False, True):
    model = mc.make_model("sqfa", 784, 4, 0.01, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    print("init rel err", rel_err(model.filters.detach().cpu(), G["sqfa_init"]), "max abs", np.abs(model.filters.detach().cpu().numpy() - G["sqfa_init"]).max())
    if exact_init:
        with torch.no_grad():
            model.parametrizations.filters.original.copy_(torch.tensor(G["sqfa_init"], device=DEV))
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    ref = G["sqfa_loss"]
    n = min(len(loss), len(ref))
    d = np.abs(loss.numpy()[:n].astype(np.float64) - ref[:n])
    print(f"exact_init={exact_init}: epochs {len(loss)} filters err {rel_err(model.filters.detach().cpu(), G['sqfa_filters']):.2e} max |dloss| {d.max():.2e}", flush=True)
