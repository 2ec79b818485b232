"""Developer aid: closure-by-closure trace of the float64 c5 golden fit (loss, |g|max per closure, closures per epoch)."""
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
o.COMPACT_LBFGS = False   # torch.optim.LBFGS itself
o.GRAPH_CLOSURE = False
model = mc.make_model("sqfa", 3072, 16, 0.01, "sphere", torch.float64, DEV)
model.fit_pca(data_statistics=stats)
with torch.no_grad():
    model.parametrizations.filters.original.copy_(torch.as_tensor(G7["sqfa_init"], dtype=torch.float64, device=DEV))
trace = []
orig = model._fused_closure_loss
def traced(p):
    out = orig(p)
    trace.append(float(out[0].detach()))
    return out
model._fused_closure_loss = traced
loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
print("epoch losses", [round(float(v), 6) for v in loss])
print("reference   ", [round(float(v), 6) for v in G7["sqfa_loss"]])
print("closures", len(trace))
for i in range(0, len(trace), 10):
    print(i, [round(v, 5) for v in trace[i:i + 10]])
