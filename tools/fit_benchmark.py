"""Metric M3: fit() wall-clock to the reference stopping rule on synthetic
configurations, float32, fit_pca init.  python tools/fit_benchmark.py [c2|c5|c3s ...]"""
import sys, os, time, faulthandler, signal
faulthandler.register(signal.SIGUSR1, all_threads=False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sqfa_amd

CFG = {"c1": (10, 784, 4, "sqfa"), "c2": (100, 784, 8, "smsqfa"), "c2s": (100, 784, 8, "sqfa"),
       "c5": (100, 3072, 16, "sqfa"), "c3": (1000, 784, 16, "smsqfa")}

def stats(C, D, device, seed=1234):
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = min(D, 128)
    cov = torch.empty(C, D, D, device=device)
    mu = torch.empty(C, D, device=device)
    for c0 in range(0, C, 50):
        n = min(50, C - c0)
        A = (torch.randn(n, D, R, generator=g) / R ** 0.5).to(device)
        cov[c0:c0 + n] = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, device=device)
        mu[c0:c0 + n] = 0.1 * torch.randn(n, D, generator=g).to(device)
    return {"means": mu, "covariances": cov}

def run(name, max_epochs=300):
    C, D, K, model_name = CFG[name]
    dev = torch.device("cuda:0")
    st = stats(C, D, dev)
    cls = sqfa_amd.model.SQFA if model_name == "sqfa" else sqfa_amd.model.SecondMomentsSQFA
    model = cls(n_dim=D, n_filters=K, feature_noise=0.01).to(dev)
    model.fit_pca(data_statistics=st)
    calls = [0]
    orig = model._fused_closure_loss
    def counted(p):
        calls[0] += 1
        return orig(p)
    model._fused_closure_loss = counted
    replay = torch.cuda.CUDAGraph.replay
    def counted_replay(self):  # closures served by the captured HIP graph
        calls[0] += 1
        return replay(self)
    torch.cuda.CUDAGraph.replay = counted_replay
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, t = model.fit(data_statistics=st, max_epochs=max_epochs, show_progress=False, return_loss=True)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    torch.cuda.CUDAGraph.replay = replay
    print(f"{name}: {model_name} C={C} D={D} K={K}: fit() {wall:.2f} s, {len(loss)} epochs, {calls[0]} closures "
          f"({wall/calls[0]*1e3:.2f} ms/closure), final loss {loss[-1].item():.6f}", flush=True)

def warm_up():
    """One throw-away fit so that library initialisation is not billed to the first configuration."""
    dev = torch.device("cuda:0")
    st = stats(10, 64, dev)
    model = sqfa_amd.model.SQFA(n_dim=64, n_filters=2, feature_noise=0.01).to(dev)
    model.fit_pca(data_statistics=st)
    model.fit(data_statistics=st, max_epochs=3, show_progress=False)


if __name__ == "__main__":
    warm_up()
    for n in (sys.argv[1:] or ["c1", "c2", "c2s", "c5"]):
        run(n, 300 if n != "c3" else 40)
