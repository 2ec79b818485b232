"""GPU parity of the native per-class matrix functions (sqfa_spd_function / sqfa_spd_function_backward: Cholesky +
one-sided Jacobi to convergence, Daleckii-Krein backward) behind sqfa_amd.linalg.spd_log / spd_sqrt, against the
reference's expression (torch.linalg.eigh + einsum, /root/reference/src/sqfa/linalg.py:121-141, 165-183) evaluated in
float64 on the CPU, values and autograd gradients; the reference's own tests check the same functions against
scipy.linalg.logm / sqrtm up to m=17 (tests/test_linalg.py:334-389) -- here sqrt(M)^2 = M and exp(log M) = M.
Tolerances: float64 1e-12 (values) / 1e-10 (gradients); float32 1e-5 (north_star's bound; observed ~2e-7)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(M, fn):
    lam, V = torch.linalg.eigh(M)
    return torch.einsum("...ij,...j,...kj->...ik", V, fn(lam), V)


def _spd(n, m, seed, cond=1e3):
    g = torch.Generator().manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(n, m, m, generator=g, dtype=torch.float64))
    ev = torch.exp((torch.rand(n, m, generator=g, dtype=torch.float64) - 0.5) * np.log(cond))
    S = (Q * ev[:, None, :]) @ Q.transpose(1, 2)
    return 0.5 * (S + S.transpose(1, 2))


@pytest.fixture
def no_library_eigh(monkeypatch):
    """The GPU path must not reach torch.linalg.eigh (rocSOLVER): the native kernels are what these tests measure."""
    def refuse(*a, **k):
        raise AssertionError("torch.linalg.eigh called on the GPU path")
    real = torch.linalg.eigh
    monkeypatch.setattr(torch.linalg, "eigh", lambda M, *a, **k: refuse() if M.is_cuda else real(M, *a, **k))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", ["spd_log", "spd_sqrt"])
@pytest.mark.parametrize("n,m", [(1, 1), (3, 2), (7, 3), (5, 5), (33, 8), (12, 9), (65, 16), (20, 17), (9, 24), (6, 33), (5, 40), (3, 64)])
def test_values_and_gradients_vs_reference_expression(no_library_eigh, n, m, name, dtype):
    from sqfa_amd import linalg
    fn = torch.log if name == "spd_log" else torch.sqrt
    S64 = _spd(n, m, 100 * m + n)
    W = torch.randn(n, m, m, generator=torch.Generator().manual_seed(m), dtype=torch.float64)   # NOT symmetric: eigh's backward symmetrises
    Sr = S64.clone().requires_grad_()
    Fr = _reference(Sr, fn)
    (gr,) = torch.autograd.grad((W * Fr).sum(), Sr)
    S = S64.to(dtype).to(DEV).requires_grad_()
    F = getattr(linalg, name)(S)
    assert F.shape == S.shape and F.dtype == dtype
    (g,) = torch.autograd.grad((W.to(dtype).to(DEV) * F).sum(), S)
    vtol, gtol = (1e-12, 1e-10) if dtype == torch.float64 else (1e-5, 1e-5)
    if dtype == torch.float32:   # the float32 INPUT rounding moves the exact answer by cond * eps: compare at the rounded input
        Sr32 = S64.float().double().requires_grad_()
        Fr = _reference(Sr32, fn)
        (gr,) = torch.autograd.grad((W.float().double() * Fr).sum(), Sr32)
    ev, eg = rel_err(F.detach().cpu(), Fr.detach().numpy()), rel_err(g.cpu(), gr.numpy())
    assert ev <= vtol and eg <= gtol, (ev, eg)
    assert torch.equal(F, F.transpose(1, 2)) and torch.equal(g, g.transpose(1, 2))   # exactly symmetric outputs


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_inverse_properties_and_batch_dims(no_library_eigh, dtype):
    """sqrt(M) sqrt(M) = M and exp(log M) = M (what the reference's scipy comparisons pin); leading batch dims are kept."""
    from sqfa_amd import linalg
    M = _spd(24, 6, 3, cond=50).reshape(2, 3, 4, 6, 6).to(dtype).to(DEV)
    R = linalg.spd_sqrt(M)
    L = linalg.spd_log(M)
    assert R.shape == M.shape and L.shape == M.shape
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert rel_err((R @ R).cpu(), M.cpu().numpy()) <= tol
    assert rel_err(torch.matrix_exp(L).cpu(), M.cpu().numpy()) <= tol
    single = linalg.spd_log(M[0, 0, 0])                        # a bare (m, m) matrix
    assert single.shape == (6, 6) and torch.equal(single, L[0, 0, 0])


def test_repeated_eigenvalues_have_finite_gradients(no_library_eigh):
    """S = a I (and a block with a doubly repeated eigenvalue): eigh's autograd divides by lambda_k - lambda_l and returns
    NaN / inf; the divided differences here have the correct limit: d log(S) = sym(G) / a at S = a I."""
    from sqfa_amd import linalg
    m = 8
    S = (2.5 * torch.eye(m, dtype=torch.float64)).repeat(3, 1, 1)
    S[1] = torch.diag(torch.tensor([1.0, 1.0, 2.0, 2.0, 2.0, 3.0, 4.0, 4.0], dtype=torch.float64))
    S = S.to(DEV).requires_grad_()
    G = torch.randn(3, m, m, generator=torch.Generator().manual_seed(0), dtype=torch.float64).to(DEV)
    (g,) = torch.autograd.grad((G * linalg.spd_log(S)).sum(), S)
    assert torch.isfinite(g).all()
    sym = 0.5 * (G + G.transpose(1, 2))
    assert rel_err(g[0].cpu(), (sym[0] / 2.5).cpu().numpy()) <= 1e-13
    d = torch.diagonal(S[1]).detach()
    gam = torch.where(d[:, None] == d[None, :], 1.0 / d[:, None].expand(m, m),
                      (torch.log(d[:, None]) - torch.log(d[None, :])) / (d[:, None] - d[None, :] + (d[:, None] == d[None, :])))
    assert rel_err(g[1].cpu(), (sym[1] * gam).cpu().numpy()) <= 1e-13


def test_not_positive_definite_class_yields_nan_not_a_fault(no_library_eigh):
    from sqfa_amd import linalg
    S = _spd(5, 4, 9).to(DEV)
    S[2] = -S[2]
    L = linalg.spd_log(S)
    assert torch.isnan(L[2]).any()
    assert torch.isfinite(L[[0, 1, 3, 4]]).all()


def test_log_euclidean_runs_on_the_native_eigensolver(no_library_eigh):
    """log_euclidean[_sq] (reference: src/sqfa/distances.py:92-138) = native per-class logarithms + pairwise distances of the
    flattened logarithms; C=1000, m=16 against the float64 CPU expression on a sample, symmetric, zero diagonal."""
    from sqfa_amd import distances
    C, m = 1000, 16
    S64 = _spd(C, m, 77, cond=100)
    S = S64.float().to(DEV).requires_grad_()
    D = distances.log_euclidean_sq(S, S)
    assert D.shape == (C, C)
    assert (D - D.t()).abs().max().item() <= 1e-4 * D.abs().max().item() and D.diagonal().abs().max().item() <= 1e-3
    Lr = _reference(S64.float().double()[:40], torch.log)
    ref = ((Lr[:, None] - Lr[None]) ** 2).sum((-2, -1))
    assert rel_err(D[:40, :40].detach().cpu(), ref.numpy()) <= 1e-5
    distances.log_euclidean(S, S).sum().backward()
    assert torch.isfinite(S.grad).all()


_prng = np.random.default_rng(77)
RANDOM_SIZES = [(int(_prng.integers(1, 71)), int(_prng.integers(1, 65))) for _ in range(12)]


@pytest.mark.parametrize("n,m", RANDOM_SIZES)
def test_spd_log_random_sizes(no_library_eigh, n, m):
    """Seeded random (n, m <= 64): ragged class counts against every padded size / lane geometry of the per-class eigensolver."""
    test_values_and_gradients_vs_reference_expression(None, n, m, "spd_log", torch.float64)
    test_values_and_gradients_vs_reference_expression(None, n, m, "spd_sqrt", torch.float32)
