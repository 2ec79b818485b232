"""Developer aid: time the full closure (projection + pair kernel + backward to the raw filters) on the GPU (metric M2)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sqfa_amd

def make_stats(C, D, seed=1234, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = min(D, 128)
    cov = torch.empty(C, D, D, device=device)
    mu = torch.empty(C, D, device=device)
    for c0 in range(0, C, 50):
        n = min(50, C - c0)
        A = (torch.randn(n, D, R, generator=g) / R ** 0.5).to(device)
        cov[c0:c0 + n] = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, device=device)
        mu[c0:c0 + n] = 0.1 * torch.randn(n, D, generator=g).to(device)
    return mu, cov

def run(C, D, K, model_name, reps=10):
    mu, cov = make_stats(C, D)
    torch.manual_seed(7)
    if model_name == "sqfa":
        model = sqfa_amd.model.SQFA(n_dim=D, n_filters=K, feature_noise=0.01).cuda()
        stats = {"means": mu, "covariances": cov}
    else:
        model = sqfa_amd.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.01).cuda()
        stats = cov + mu[:, :, None] * mu[:, None, :]
        del cov
    prepared = model._prepare_statistics(stats)
    def closure():
        model.zero_grad()
        loss, flags = model._fused_closure_loss(prepared)
        loss.backward()
        return loss
    for _ in range(3): closure()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): l = closure()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    byts = 4 * C * D * D
    print(f"{model_name} C={C} D={D} K={K}: closure {t*1e3:.3f} ms ({1/t:.1f}/s); scatter stream {byts/1e9:.2f} GB -> {byts/t/1e9:.0f} GB/s if read once; loss {l.item():.5f}", flush=True)
    # projection only
    F = model.filters.detach().clone().requires_grad_(True)
    Sx = stats["covariances"] if isinstance(stats, dict) else stats
    def proj():
        S = sqfa_amd.linalg.conjugate_matrix(Sx, F)
        S.sum().backward()
    for _ in range(3): proj()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): proj()
    torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / reps
    print(f"    projection fwd+bwd alone {tp*1e3:.3f} ms -> {byts/tp/1e9:.0f} GB/s (one pass equivalent)", flush=True)

if __name__ == "__main__":
    run(100, 784, 8, "smsqfa")
    run(1000, 784, 16, "smsqfa")
    run(1000, 784, 16, "sqfa")
    run(100, 3072, 16, "sqfa")
