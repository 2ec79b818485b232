"""The reference's own property tests for this path (its tests/test_distances.py:40-124 and
tests/test_linalg.py:182-245), re-stated against the HIP operators: shapes incl. the squeeze
rules, symmetry, zero diagonal, invariance under inversion, agreement with the log-Euclidean
distance at the identity, and generalized eigenvalues against scipy."""
import numpy as np
import pytest
import scipy.linalg
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sample_spd(n, m, seed):
    g = torch.Generator().manual_seed(seed)
    ev = 2 * torch.rand(n, m, generator=g, dtype=torch.float64) ** 2 + 0.01
    skew = torch.randn(n, m, m, generator=g, dtype=torch.float64).tril(-1)
    Q = torch.matrix_exp(skew - skew.transpose(1, 2))
    return (Q * ev[:, None, :]) @ Q.transpose(1, 2)


@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-9), (torch.float32, 1e-4)])
@pytest.mark.parametrize("n_classes", [1, 4, 8])
@pytest.mark.parametrize("n_dim", [2, 4, 6])
def test_distance_sq(n_classes, n_dim, dtype, atol):
    from sqfa_amd import distances
    spd = sample_spd(n_classes, n_dim, 100 * n_classes + n_dim).to(DEV, dtype)
    spd = spd.squeeze(0) if n_classes == 1 else spd
    d2 = distances.affine_invariant_sq(spd, spd)
    assert d2.shape == ((n_classes, n_classes) if n_classes != 1 else ())
    assert torch.allclose(d2, d2.T if d2.dim() else d2, atol=atol)
    diag = d2.diagonal() if d2.dim() else d2
    assert torch.allclose(diag, torch.zeros_like(diag), atol=atol)
    inv = torch.linalg.inv(spd.double()).to(dtype)
    assert torch.allclose(d2, distances.affine_invariant_sq(inv, inv), atol=50 * atol)
    eye = torch.eye(n_dim, dtype=dtype, device=DEV)
    ai = distances.affine_invariant_sq(spd, eye)
    le = distances.log_euclidean_sq(spd, eye)
    assert ai.shape == le.shape and torch.allclose(ai, le, atol=50 * atol)


@pytest.mark.parametrize("n_classes", [1, 4, 8])
@pytest.mark.parametrize("n_dim", [2, 4, 6])
def test_fisher_rao_sq(n_classes, n_dim):
    from sqfa_amd import distances
    spd = sample_spd(n_classes, n_dim, 7 * n_classes + n_dim).to(DEV)
    mu = torch.randn(n_classes, n_dim, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).to(DEV)
    if n_classes == 1:
        spd, mu = spd.squeeze(0), mu.squeeze(0)
    st = {"means": mu, "covariances": spd}
    fr = distances.fisher_rao_lower_bound_sq(st, st)
    assert fr.shape == ((n_classes, n_classes) if n_classes != 1 else ())
    assert torch.allclose(fr, fr.T if fr.dim() else fr, atol=1e-9)
    diag = fr.diagonal() if fr.dim() else fr
    assert torch.allclose(diag, torch.zeros_like(diag), atol=1e-9)


@pytest.mark.parametrize("nA", [1, 4, 8])
@pytest.mark.parametrize("nB", [1, 4, 8])
@pytest.mark.parametrize("n_dim", [2, 4, 6])
def test_generalized_eigenvalues_vs_scipy(nA, nB, n_dim):
    from sqfa_amd import linalg
    A = sample_spd(nA, n_dim, 11 * nA + n_dim)
    B = sample_spd(nB, n_dim, 13 * nB + n_dim + 1)
    lam = linalg.generalized_eigenvalues(A.to(DEV), B.to(DEV)).cpu()
    expect_dim = 3 - (nA == 1) - (nB == 1)
    assert lam.dim() == expect_dim and lam.shape[-1] == n_dim
    ref = np.empty((nA, nB, n_dim))
    for i in range(nA):
        for j in range(nB):
            ref[i, j] = np.sort(scipy.linalg.eigvals(A[i].numpy(), B[j].numpy()).real)[::-1]
    assert np.allclose(lam.numpy().reshape(nA, nB, n_dim), ref, atol=1e-9)


@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
@pytest.mark.parametrize("n_classes", [1, 4, 8])
@pytest.mark.parametrize("n_dim", [2, 4, 6])
def test_bhattacharyya(n_classes, n_dim, dtype, atol):
    """reference tests/test_distances.py:127-151 on the native Gaussian pair kernel: shape incl. the
    single-class squeeze, symmetry, zero diagonal."""
    from sqfa_amd import distances
    spd = sample_spd(n_classes, n_dim, 17 * n_classes + n_dim).to(DEV, dtype)
    mu = torch.randn(n_classes, n_dim, dtype=torch.float64, generator=torch.Generator().manual_seed(5)).to(DEV, dtype)
    if n_classes == 1:
        spd, mu = spd.squeeze(0), mu.squeeze(0)
    st = {"means": mu, "covariances": spd}
    bh = distances.bhattacharyya(st, st)
    assert bh.shape == ((n_classes, n_classes) if n_classes != 1 else ())
    assert torch.allclose(bh, bh.T if bh.dim() else bh, atol=atol)
    diag = bh.diagonal() if bh.dim() else bh
    assert torch.allclose(diag, torch.zeros_like(diag), atol=atol)
    he = distances.hellinger(st, st)
    assert he.shape == bh.shape and torch.isfinite(he).all()


@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
@pytest.mark.parametrize("n_classes", [1, 4, 8])
@pytest.mark.parametrize("n_dim", [2, 4, 6])
def test_mahalanobis(n_classes, n_dim, dtype, atol):
    """reference tests/test_distances.py:154-184: with covariances 2 I the Mahalanobis distance is the
    Euclidean distance / sqrt(2) (after subtracting the sqrt(eps) of the diagonal)."""
    from sqfa_amd import distances
    spd = (torch.eye(n_dim, dtype=dtype).repeat(n_classes, 1, 1) * 2.0).to(DEV)
    mu = torch.randn(n_classes, n_dim, dtype=torch.float64, generator=torch.Generator().manual_seed(9)).to(DEV, dtype)
    st = {"means": mu, "covariances": spd}
    ma = distances.mahalanobis(st, st)
    assert ma.shape == (n_classes, n_classes)          # mahalanobis_sq is not squeezed in the reference either
    ma = ma - torch.eye(n_classes, dtype=dtype, device=DEV) * 1e-3
    assert torch.allclose(ma, ma.T, atol=atol)
    assert torch.allclose(ma.diagonal(), torch.zeros(n_classes, dtype=dtype, device=DEV), atol=10 * atol)
    euclid = torch.cdist(mu, mu, p=2.0) / 2.0 ** 0.5
    # sqrt(d^2 + 1e-6) - d <= 1e-6 / (2 d): the reference's own tolerance for this comparison is 1e-5
    assert torch.allclose(ma, euclid, atol=1e-5)
    fr = distances.fisher_rao_same_cov(st, st)
    assert fr.shape == (n_classes, n_classes) and torch.isfinite(fr).all()
