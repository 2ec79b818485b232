"""cProfile of fit() on a small configuration: where the per-closure host time goes.
python tools/profile_fit.py [c1|c2|c2s|c5]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_benchmark as fb
import torch
import sqfa_amd

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
C, D, K, model_name = fb.CFG[name]
dev = torch.device("cuda:0")
st = fb.stats(C, D, dev)
cls = sqfa_amd.model.SQFA if model_name == "sqfa" else sqfa_amd.model.SecondMomentsSQFA
for rep in range(2):  # the first fit warms every library path up
    model = cls(n_dim=D, n_filters=K, feature_noise=0.01).to(dev)
    model.fit_pca(data_statistics=st)
    pr = cProfile.Profile()
    pr.enable()
    model.fit(data_statistics=st, max_epochs=30, show_progress=False)
    torch.cuda.synchronize()
    pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(35)
print(out.getvalue())
