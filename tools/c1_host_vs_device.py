"""Developer aid: loss history of the c1-shaped synthetic fit with host-side and device-side optimizer state."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fit_benchmark as fb
import sqfa_amd, sqfa_amd._optim as o
dev = torch.device("cuda:0")
C, D, K, _ = fb.CFG["c1"]
st = fb.stats(C, D, dev)
for dtype in (torch.float32, torch.float64):
    std = {k: v.to(dtype) for k, v in st.items()}
    for limit in (8192, 0):
        o.HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT = limit
        torch.manual_seed(0)
        model = sqfa_amd.model.SQFA(n_dim=D, n_filters=K, feature_noise=0.01).to(dev).to(dtype)
        model.fit_pca(data_statistics=std)
        f0 = model.filters.detach().clone()
        loss, t = model.fit(data_statistics=std, max_epochs=300, show_progress=False, return_loss=True)
        d = model.get_class_distances(std)
        tri = torch.tril_indices(C, C, -1)
        print(str(dtype)[6:], "host" if limit else "device", len(loss), [round(float(v), 5) for v in loss[:6]], "... final", float(loss[-1]),
              "recomputed", float(-d[tri[0], tri[1]].mean()), "init norm", float(f0.norm()), flush=True)
