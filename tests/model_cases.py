"""Shared checks of the model shell against the reference goldens (G3 closure, G4 fit, G5 quirks).
Used by the CPU tests (oracle pair backend injected) and by the GPU tests (HIP backend)."""
import contextlib

import numpy as np
import torch

from conftest import load_golden, rel_err

G3 = load_golden("g3_closure.npz")
G4 = load_golden("g4_fit.npz")
G5 = load_golden("g5_quirks.npz")
G3O = load_golden("g3o_closure_orthogonal.npz")   # constraint="orthogonal" with the parametrization's `base` buffer stored


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
# initial filters, so G3's orthogonal cases are not reproducible from the raw parameter alone (and most
# of them are degenerate in the reference: a random raw diagonal truncates to 0 in the Householder map,
# NaN gradients).  Golden G3O (make_golden.py:g3o) repeats them with the buffer stored and raw parameters
# of the form the parametrization itself produces; G3's own orthogonal keys stay unused.
G3_KEYS = sorted({k[: -len("_loss_f64")] for k in G3 if k.endswith("_loss_f64") and "_orthogonal_" not in k})
G3O_KEYS = sorted({k[: -len("_loss_f64")] for k in G3O if k.endswith("_loss_f64")})
G3O_KEYS_F32 = [k for k in G3O_KEYS if f"{k}_grad_f32" in G3O and np.isfinite(G3O[f"{k}_grad_f32"]).all()]


def parse_g3_key(key):
    model_name, constraint, noise, K = key.split("_")
    return model_name, constraint, float(noise[1:]), int(K[1:])


def set_orthogonal_base(model, base):
    """Put a stored `base` into torch's orthogonal parametrization (reference: src/sqfa/model.py:416-431
    registers torch.nn.utils.parametrizations.orthogonal; its forward is  filters = (base @ Q(raw))^T)."""
    par = model.parametrizations.filters[0]
    par.base = torch.as_tensor(base, dtype=par.base.dtype, device=par.base.device)


def check_closure(key, dtype, device, tol_loss, tol_grad, tol_dist, G3=G3):
    model_name, constraint, noise, K = parse_g3_key(key)
    tag = "f64"
    model = make_model(model_name, 8, K, noise, constraint, dtype, device)
    if constraint == "orthogonal":
        set_orthogonal_base(model, G3[f"{key}_base"])
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


def check_orthogonal_fit(model_name, device, tol_loss=1e-6, tol_filters=1e-5):
    """Three epochs of a float64 fit under constraint="orthogonal" from the reference's exact starting point
    (its `base` buffer and raw parameter, golden G3O): per-epoch losses and learned filters."""
    key = f"{model_name}_orthogonal_fit_K3"
    dtype = torch.float64
    model = make_model(model_name, 8, 3, 1e-2, "orthogonal", dtype, device)
    set_orthogonal_base(model, G3O[f"{key}_base"])
    with torch.no_grad():
        model.parametrizations.filters.original.copy_(torch.tensor(G3O[f"{key}_raw"], dtype=dtype))
    assert rel_err(model.filters.detach().cpu(), G3O[f"{key}_init"]) < 1e-12
    stats = {"means": torch.tensor(G3O["rotated_mu"], dtype=dtype, device=device),
             "covariances": torch.tensor(G3O["rotated_cov"], dtype=dtype, device=device)}
    loss, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
    assert np.abs(loss.numpy() - G3O[f"{key}_loss"]).max() <= tol_loss
    F = model.filters.detach().cpu()
    assert rel_err(F, G3O[f"{key}_filters"]) <= tol_filters
    assert np.abs((F @ F.T).numpy() - np.eye(3)).max() < 1e-10


def check_early_epochs(golden, n_classes, n_dim, n_filters, epochs, device, tol_filters=1e-5, **fit_kwargs):
    """north_star "learned filters vs the reference to 1e-5", asserted where it is a property of the algorithm: the
    reference's float64 fit stopped after `epochs` epochs (src/sqfa/_optim.py:105-134, `max_epochs`), before its
    fixed-step trajectory amplifies rounding (goldens G6c: BASELINE config 1 shape, G7e: config 5 shape;
    make_golden.py:_early_fits).  Starts from the golden's exact initial filters; prints the reference's own
    distance from itself at that epoch (Cholesky-route distance_fun) next to the result."""
    G = load_golden(golden)
    stats = c2_statistics(C=n_classes, D=n_dim)
    assert np.allclose(stats["covariances"][0, :4, :4].numpy(), G["check_cov00"], rtol=1e-12)
    stats = {k: v.to(device) for k, v in stats.items()}
    model = make_model("sqfa", n_dim, n_filters, 0.01, "sphere", torch.float64, device)
    model.fit_pca(data_statistics=stats)
    init = torch.as_tensor(G["sqfa_init"], dtype=torch.float64, device=device)
    assert rel_err(model.filters.detach().cpu(), init.cpu()) < 1e-8      # fit_pca itself
    with torch.no_grad():
        model.parametrizations.filters.original.copy_(init)
    loss, _ = model.fit(data_statistics=stats, max_epochs=epochs, show_progress=False, return_loss=True, **fit_kwargs)
    ref_loss, ref_f = G[f"sqfa_e{epochs}_loss"], G[f"sqfa_e{epochs}_filters"]
    err = rel_err(model.filters.detach().cpu(), ref_f)
    own = rel_err(G[f"sqfa_e{epochs}_cholroute_filters"], ref_f)
    print(f"C={n_classes} D={n_dim} K={n_filters}, {epochs} epochs: filters vs reference {err:.2e} "
          f"(reference vs its own Cholesky-route run {own:.2e}); losses {np.abs(loss.numpy() - ref_loss).max():.2e}")
    assert len(loss) == len(ref_loss)
    assert np.abs(loss.numpy() - ref_loss).max() < 1e-6
    assert err <= tol_filters
    return err, own


def rot_k4_spread(model_name):
    """How far the REFERENCE's own three-epoch fit of the rot / K=4 case moves under rounding-level
    perturbations (golden G4D, make_golden.py:g4d: Cholesky-route distance_fun and eight 1e-14-perturbed
    starts): relative filter distance and final-loss distance from its unperturbed run.  smSQFA: 0.32 ... 0.48
    (a chaotic first step on this near-singular data set), SQFA: 5e-5 ... 5e-4."""
    G4D = load_golden("g4d_fit_rot_spread.npz")
    key = f"rot_{model_name}_K4_e3"
    ref_f, ref_l = G4[f"{key}_filters"], G4[f"{key}_loss"]
    fs = [G4D[f"{key}_cholroute_filters"]] + list(G4D[f"{key}_ensemble_filters"])
    ls = [G4D[f"{key}_cholroute_loss"]] + list(G4D[f"{key}_ensemble_loss"])
    return max(rel_err(f, ref_f) for f in fs), max(np.abs(l - ref_l).max() for l in ls)


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
