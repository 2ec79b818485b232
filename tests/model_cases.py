"""Shared checks of the model shell against the reference goldens (G3 closure, G4 fit, G5 quirks).
Used by the CPU tests (oracle pair backend injected) and by the GPU tests (HIP backend)."""
import contextlib

import numpy as np
import torch

from conftest import load_golden, rel_err

G3 = load_golden("g3_closure.npz")
G4 = load_golden("g4_fit.npz")
G5 = load_golden("g5_quirks.npz")


@contextlib.contextmanager
def default_dtype(dt):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def make_model(model_name, n_dim, K, noise, constraint, dtype, device):
    import sqfa_amd
    cls = sqfa_amd.model.SQFA if model_name == "sqfa" else sqfa_amd.model.SecondMomentsSQFA
    with default_dtype(dtype):
        model = cls(n_dim=n_dim, n_filters=K, feature_noise=noise, constraint=constraint)
    if dtype == torch.float64:
        model = model.double()
    return model.to(device)


# torch's `orthogonal` parametrization keeps a random `base` buffer created from the (unseeded)
# initial filters, so the captured orthogonal cases are not reproducible from the raw parameter
# alone (and in the reference most of them are degenerate: NaN gradients).  They stay in the
# golden file for the record; tests use the sphere / none cases and check orthogonal separately.
G3_KEYS = sorted({k[: -len("_loss_f64")] for k in G3 if k.endswith("_loss_f64") and "_orthogonal_" not in k})


def parse_g3_key(key):
    model_name, constraint, noise, K = key.split("_")
    return model_name, constraint, float(noise[1:]), int(K[1:])


def check_closure(key, dtype, device, tol_loss, tol_grad, tol_dist):
    model_name, constraint, noise, K = parse_g3_key(key)
    tag = "f64"
    model = make_model(model_name, 8, K, noise, constraint, dtype, device)
    with torch.no_grad():
        model.parametrizations.filters.original.copy_(torch.tensor(G3[f"raw_filters_K{K}"], dtype=dtype))
    cov = torch.tensor(G3["rotated_cov"], dtype=dtype, device=device)
    mu = torch.tensor(G3["rotated_mu"], dtype=dtype, device=device)
    inp = {"means": mu, "covariances": cov} if model_name == "sqfa" else cov
    assert rel_err(model.filters.detach().cpu(), G3[f"{key}_filters_{tag}"]) < (1e-12 if dtype == torch.float64 else 1e-6)
    D = model.get_class_distances(inp, regularized=True)
    assert np.abs(D.detach().cpu().numpy() - G3[f"{key}_D_{tag}"]).max() <= tol_dist * max(1.0, G3[f"{key}_D_{tag}"].max())
    # fused closure loss + backward to the raw parameter
    fused = model._fused_closure_loss(model._prepare_statistics(inp))
    assert fused is not None
    loss, flags = fused
    assert flags.tolist() == [0, 0]
    model.zero_grad()
    loss.backward()
    ref = float(G3[f"{key}_loss_{tag}"])
    assert abs(loss.item() - ref) <= tol_loss * abs(ref)
    g = model.parametrizations.filters.original.grad
    gref = G3[f"{key}_grad_{tag}"]
    assert np.linalg.norm(g.cpu().numpy() - gref) <= tol_grad * max(np.linalg.norm(gref), 1e-3)
    # generic (matrix) path gives the same gradient
    model.zero_grad()
    r, c = torch.tril_indices(5, 5, offset=-1)
    (-D[r.to(device), c.to(device)].mean()).backward()
    g2 = model.parametrizations.filters.original.grad
    assert np.linalg.norm(g2.cpu().numpy() - gref) <= tol_grad * max(np.linalg.norm(gref), 1e-3)


def fit_stats(dname, dtype, device):
    if dname == "rot":
        cov = torch.tensor(G4["rotated_cov"], dtype=dtype, device=device)
        return {"means": torch.zeros(5, 8, dtype=dtype, device=device), "covariances": cov}
    return {"means": torch.tensor(G4["syn_mu"], dtype=dtype, device=device),
            "covariances": torch.tensor(G4["syn_cov"], dtype=dtype, device=device)}


def check_fit(dname, model_name, K, noise, epochs, device, tol_loss, tol_filters, pairwise=False):
    dtype = torch.float64
    stats = fit_stats(dname, dtype, device)
    n_dim = stats["covariances"].shape[-1]
    model = make_model(model_name, n_dim, K, noise, "sphere", dtype, device)
    model.fit_pca(data_statistics=stats)
    if pairwise:
        key = f"{dname}_{model_name}_pairwise_K{K}"
    else:
        key = f"{dname}_{model_name}_K{K}_e{epochs}"
        assert rel_err(model.filters.detach().cpu(), G4[f"{key}_init"]) < 1e-10   # pca_from_scatter quirk
    loss, elapsed = model.fit(data_statistics=stats, max_epochs=epochs, show_progress=False, return_loss=True,
                              pairwise=pairwise)
    ref_loss = G4[f"{key}_loss"]
    assert loss.shape == ref_loss.shape, f"epoch counts differ: {loss.shape} vs {ref_loss.shape}"
    assert elapsed.shape == loss.shape
    assert np.abs(loss.numpy() - ref_loss).max() <= tol_loss
    assert rel_err(model.filters.detach().cpu(), G4[f"{key}_filters"]) <= tol_filters


def c2_statistics(C=100, D=784, seed=1234, dtype=torch.float64):
    """Same generator as tests/golden/make_golden.py:c2_statistics (BASELINE config c2 shape)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = min(D, 128)
    cov = torch.empty(C, D, D, dtype=dtype)
    mu = torch.empty(C, D, dtype=dtype)
    for c0 in range(0, C, 50):
        n = min(50, C - c0)
        A = (torch.randn(n, D, R, generator=g, dtype=torch.float32) / R ** 0.5).to(dtype)
        cov[c0:c0 + n] = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, dtype=dtype)
        mu[c0:c0 + n] = (0.1 * torch.randn(n, D, generator=g, dtype=torch.float32)).to(dtype)
    return {"means": mu, "covariances": cov}


def ragged_points(C=1000, d=5, seed=606):
    """Same generator as tests/golden/make_golden.py:ragged_points (golden G5c)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(2, 42, size=C)
    y = np.repeat(np.arange(C), sizes)
    X = (rng.standard_normal((len(y), d)) * rng.uniform(0.5, 2.0, size=(1, d)) + 0.3 * rng.standard_normal((C, d))[y])
    X = X.astype(np.float32).astype(np.float64)
    perm = rng.permutation(len(y))
    return X[perm], y[perm]


def check_class_statistics_vs_reference(device, dtype=torch.float64, tol=1e-11):
    """class_statistics / OAS on `device` against the reference's outputs: the small G5 case and the
    ragged 1000-class G5c case (means, covariances, second moments; float labels accepted)."""
    from sqfa_amd import statistics
    G5C = load_golden("g5c_class_statistics.npz")
    X = torch.tensor(G5["pts_X"], dtype=dtype, device=device)
    y = torch.tensor(G5["pts_y"], device=device)
    for est in ("empirical", "oas"):
        st = statistics.class_statistics(X, y, estimator=est)
        for k, v in st.items():
            assert v.device.type == torch.device(device).type
            assert rel_err(v.cpu(), G5[f"class_stats_{est}_{k}"]) < tol, (est, k)
    assert rel_err(statistics.oas_covariance(X).cpu(), G5["oas_cov"]) < tol
    assert rel_err(statistics.sample_covariance(X).cpu(), G5["sample_cov"]) < tol
    Xn, yn = ragged_points()
    assert np.array_equal(Xn[:3], G5C["check_X0"])
    Xr = torch.tensor(Xn, dtype=dtype, device=device)
    yr = torch.tensor(yn, device=device)
    for est in ("empirical", "oas"):
        st = statistics.class_statistics(Xr, yr, estimator=est)
        for k, v in st.items():
            assert rel_err(v.cpu(), G5C[f"{est}_{k}"]) < tol, (est, k)
    st = statistics.class_statistics(Xr, yr.to(dtype), estimator="empirical")
    assert rel_err(st["means"].cpu(), G5C["float_labels_means"]) < tol
