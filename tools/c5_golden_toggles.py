"""Developer aid: the float64 c5 golden fit (tests/test_gpu_model.py::test_c5_config_full_fit_matches_reference_f64)
under L-BFGS feature toggles: distance of the final filters to the reference's and per-epoch loss differences."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import model_cases as mc
from conftest import load_golden
import sqfa_amd._optim as o, sqfa_amd._lbfgs as l
G7 = load_golden("g7_fit_c5.npz")
DEV = "cuda:0"
stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=100, D=3072).items()}
def run(tag):
    model = mc.make_model("sqfa", 3072, 16, 0.01, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    with torch.no_grad():
        model.parametrizations.filters.original.copy_(torch.as_tensor(G7["sqfa_init"], dtype=torch.float64, device=DEV))
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    F = model.filters.detach().cpu().numpy(); R = G7["sqfa_filters"]
    err = np.linalg.norm(F - R) / np.linalg.norm(R)
    ref = G7["sqfa_loss"]; n = min(len(ref), len(loss))
    print(f"{tag}: {len(loss)} epochs, filters vs reference {err:.2e}, |dloss| {np.abs(loss.numpy()[:n] - ref[:n]).round(7)}", flush=True)
run("default")
l.CompactLBFGS.speculate_descent_test = False; run("no speculation")
o.DEFERRED_CLOSURE = False; run("no speculation, no deferred closure")
l._History.native = False; run("... torch compact form")
l._History.native = True; o.COMPACT_LBFGS = False; run("torch.optim.LBFGS on the device")
