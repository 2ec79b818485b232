"""GPU parity of the remaining distance_fun operators against values AND
gradients computed by the reference (golden G5b, tests/golden/make_golden.py:g5b):
bhattacharyya / mahalanobis[_sq] / hellinger / fisher_rao_same_cov through the native Gaussian pair
kernel (sqfa_gauss_pair_terms), log_euclidean[_sq] through per-class logarithms + exact pairwise
distances.  Tolerances: float64 1e-9 (values) / 1e-8 (gradients); float32: max(1e-5, 5 x the
reference's own float32-vs-float64 deviation on that case, golden G5b's f32 keys) -- the same rule as the
affine-invariant family (tests/test_gpu_parity.py:_tols; 1e-5 is north_star's bound, 3e-5 until round 4)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G5B = load_golden("g5b_other_operators.npz")
CASES = [tuple(int(v) for v in c) for c in G5B["cases"]]
GAUSS_OPS = ("bhattacharyya", "mahalanobis_sq", "mahalanobis", "hellinger", "fisher_rao_same_cov")


def _tol(key, name, what, dtype, floor64, floor32=1e-5):
    if dtype == torch.float64:
        return floor64
    dev = rel_err(G5B[f"{key}_{name}{what}_f32"], G5B[f"{key}_{name}{what}_f64"])
    return max(floor32, 5 * dev)


def _inputs(key, nB, dtype):
    a = {"means": torch.tensor(G5B[f"{key}_muA"], dtype=dtype, device=DEV, requires_grad=True),
         "covariances": torch.tensor(G5B[f"{key}_covA"], dtype=dtype, device=DEV, requires_grad=True)}
    if not nB:
        return a, a
    b = {"means": torch.tensor(G5B[f"{key}_muB"], dtype=dtype, device=DEV, requires_grad=True),
         "covariances": torch.tensor(G5B[f"{key}_covB"], dtype=dtype, device=DEV, requires_grad=True)}
    return a, b


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", GAUSS_OPS)
@pytest.mark.parametrize("nA,nB,K", CASES)
def test_gaussian_pair_operators_vs_reference(nA, nB, K, name, dtype):
    from sqfa_amd import distances
    key = f"A{nA}_B{nB}_K{K}"
    a, b = _inputs(key, nB, dtype)
    D = getattr(distances, name)(a, b)
    ref = G5B[f"{key}_{name}_f64"]
    assert tuple(D.shape) == ref.shape                      # the reference's squeeze behaviour
    assert rel_err(D.detach().cpu(), ref) <= _tol(key, name, "", dtype, 1e-9)
    W = torch.tensor(G5B[f"{key}_W"], dtype=dtype, device=DEV)
    loss = (W.reshape(D.shape) * D).sum()
    wrt = [("gcovA", a["covariances"]), ("gmuA", a["means"])]
    if nB:
        wrt += [("gcovB", b["covariances"]), ("gmuB", b["means"])]
    grads = torch.autograd.grad(loss, [t for _, t in wrt])
    for (gname, _), g in zip(wrt, grads):
        refg = G5B[f"{key}_{name}_{gname}_f64"]
        if np.isnan(refg).any():   # the reference's own gradient is NaN (acosh'(1) = inf on the self-pair diagonal)
            assert torch.isnan(g).any()
            continue
        if np.abs(refg).max() == 0:                          # e.g. means gradient at K where delta = 0
            assert g.abs().max().item() <= 1e-6
            continue
        # hellinger saturates (exp(-bh) -> 0) for well separated classes: its gradient is then ~1e-7 of
        # the other operators' and made of cancelling terms, so its relative error floor is 1e-6
        floor64 = 1e-6 if name == "hellinger" else 1e-8
        assert rel_err(g.cpu(), refg) <= _tol(key, name, f"_{gname}", dtype, floor64), (name, gname)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", ["log_euclidean_sq", "log_euclidean"])
@pytest.mark.parametrize("nA,nB,K", CASES)
def test_log_euclidean_vs_reference(nA, nB, K, name, dtype, monkeypatch):
    """Values AND gradients of the reference (golden G5b) through the native per-class eigensolver (sqfa_spd_function /
    sqfa_spd_function_backward): torch.linalg.eigh (rocSOLVER) must not be reached on the GPU path."""
    from sqfa_amd import distances
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (_ for _ in ()).throw(AssertionError("library eigh on the GPU path")))
    key = f"A{nA}_B{nB}_K{K}"
    a, b = _inputs(key, nB, dtype)
    D = getattr(distances, name)(a["covariances"], b["covariances"])
    ref = G5B[f"{key}_{name}_f64"]
    assert tuple(D.shape) == ref.shape
    assert rel_err(D.detach().cpu(), ref) <= _tol(key, name, "", dtype, 1e-9)
    W = torch.tensor(G5B[f"{key}_W"], dtype=dtype, device=DEV)
    wrt = [("gcovA", a["covariances"])] + ([("gcovB", b["covariances"])] if nB else [])
    grads = torch.autograd.grad((W.reshape(D.shape) * D).sum(), [t for _, t in wrt])
    for (gname, _), g in zip(wrt, grads):
        assert rel_err(g.cpu(), G5B[f"{key}_{name}_{gname}_f64"]) <= _tol(key, name, f"_{gname}", dtype, 1e-8, 1e-5)


def test_gaussian_operators_large_batch_properties():
    """C=1000, K=16 (the size at which the reference's (C,C,K,K) tensor is 1 GB): symmetry, zero
    Mahalanobis diagonal, agreement of float32 with float64, and a model closure that uses
    bhattacharyya as distance_fun trains."""
    import sqfa_amd
    from sqfa_amd import distances
    g = torch.Generator().manual_seed(5)
    C, K = 1000, 16
    A = torch.randn(C, K, K + 4, generator=g, dtype=torch.float64)
    st64 = {"means": 0.5 * torch.randn(C, K, generator=g, dtype=torch.float64).to(DEV),
            "covariances": (A @ A.transpose(1, 2) / K + 0.1 * torch.eye(K, dtype=torch.float64)).to(DEV)}
    st32 = {k: v.float() for k, v in st64.items()}
    for name in GAUSS_OPS:
        D64 = getattr(distances, name)(st64, st64)
        D32 = getattr(distances, name)(st32, st32)
        assert D64.shape == (C, C) and torch.isfinite(D64).all()
        assert (D64 - D64.t()).abs().max().item() <= 1e-10 * D64.abs().max().item()
        assert rel_err(D32.cpu(), D64.cpu()) < 2e-5, name
    assert distances.mahalanobis_sq(st64, st64).diagonal().abs().max().item() < 1e-12
    stats = {"means": torch.randn(30, 40, generator=g).to(DEV), "covariances": None}
    B = torch.randn(30, 40, 60, generator=g)
    stats["covariances"] = (B @ B.transpose(1, 2) / 60).to(DEV)
    model = sqfa_amd.model.SQFA(n_dim=40, n_filters=3, feature_noise=1e-2, distance_fun=distances.bhattacharyya).to(DEV)
    loss, _ = model.fit(data_statistics=stats, max_epochs=4, show_progress=False, return_loss=True)
    assert torch.isfinite(loss).all() and loss[-1] < loss[0]
