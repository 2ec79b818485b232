"""SPD matrix algebra with the reference's public names (src/sqfa/linalg.py).

``generalized_eigenvalues`` runs on the HIP pair kernel; ``spd_log`` / ``spd_sqrt`` of GPU tensors run on the native
per-class eigen-decomposition (sqfa_spd_function: Cholesky + one-sided Jacobi, closed-form backward) -- the building
block of ``log_euclidean``; CPU tensors, and the remaining helpers (not on the hot path, SURVEY.md
section 2), are thin torch expressions.
"""
import torch

from . import _native

__all__ = [
    "conjugate_matrix",
    "generalized_eigenvalues",
    "generalized_eigenvectors",
    "spd_sqrt",
    "spd_log",
    "spd_inv_sqrt",
]


def __dir__():
    return __all__


def _drop_unit_batch_dims(T, n_lead):
    """Remove the leading batch dims (among the first n_lead) that have size 1."""
    for d in range(n_lead - 1, -1, -1):
        if T.shape[d] == 1:
            T = T.squeeze(d)
    return T


def conjugate_matrix(A, B):
    """B A B^T for every combination of the batches: A (nA,d,d), B (nB,k,d) or (k,d) ->
    (nA,nB,k,k); batch dims of size 1 are squeezed (reference: src/sqfa/linalg.py:19-45)."""
    if A.dim() == 2:
        A = A[None]
    if B.dim() < 2:
        raise ValueError("B must have at least 2 dimensions.")
    if B.dim() == 2:
        out = B @ A @ B.transpose(-2, -1)            # (nA,k,k)
        return _drop_unit_batch_dims(out, 1)
    left = B[None] @ A[:, None]                       # (nA,nB,k,d)
    out = left @ B.transpose(-2, -1)[None]
    return _drop_unit_batch_dims(out, 2)


def generalized_eigenvalues(A, B):
    """Generalized eigenvalues of every pair (A_i, B_j), descending, shape (nA,nB,m) with
    unit batch dims squeezed (reference: src/sqfa/linalg.py:48-70).  Computed by the HIP
    pair kernel (Cholesky whitening + one-sided Jacobi); differentiable (closed-form backward
    through a second launch, _native.GeneralizedEigenvalues)."""
    A3 = A[None] if A.dim() == 2 else A
    B3 = B[None] if B.dim() == 2 else B
    lam = _native.generalized_eigenvalues_raw(A3, B3)
    return _drop_unit_batch_dims(lam, 2)


def _sym_eig_fn(M, fn):
    lam, V = torch.linalg.eigh(M)
    return (V * fn(lam).unsqueeze(-2)) @ V.transpose(-2, -1)


def spd_sqrt(M):
    """Symmetric square root (reference: src/sqfa/linalg.py:121-141).  GPU tensors: sqfa_spd_function (HIP)."""
    if _native.spd_function_supported(M):
        return _native.spd_function(M, _native.SPD_SQRT)
    return _sym_eig_fn(M, torch.sqrt)


def spd_log(M):
    """Matrix logarithm of SPD matrices (reference: src/sqfa/linalg.py:165-183).  GPU tensors: sqfa_spd_function (HIP),
    differentiable in closed form (finite also for repeated eigenvalues, where eigh's autograd is not)."""
    if _native.spd_function_supported(M):
        return _native.spd_function(M, _native.SPD_LOG)
    return _sym_eig_fn(M, torch.log)


def spd_inv_sqrt(M):
    """A whitening W with W M W^T = I.  Like the reference (src/sqfa/linalg.py:144-162) this
    is Lambda^{-1/2} V^T, not the symmetric inverse root."""
    lam, V = torch.linalg.eigh(M)
    return (V * torch.rsqrt(lam).unsqueeze(-2)).transpose(-2, -1)


def generalized_eigenvectors(A, B):
    """Unit-norm generalized eigenvectors (columns) and eigenvalues of (A_i, B_j), sorted by
    descending eigenvalue (reference: src/sqfa/linalg.py:73-118).  LDA helper, torch only."""
    A3 = A[None] if A.dim() == 2 else A
    B3 = B[None] if B.dim() == 2 else B
    W = spd_inv_sqrt(B3)                                            # (nB,m,m)
    M = W[None] @ A3[:, None] @ W.transpose(-2, -1)[None]           # (nA,nB,m,m)
    lam, V = torch.linalg.eigh(M)
    lam, V = lam.flip(-1), V.flip(-1)
    U = W.transpose(-2, -1)[None] @ V
    U = U / torch.linalg.norm(U, dim=-2, keepdim=True)
    return _drop_unit_batch_dims(U, 2), _drop_unit_batch_dims(lam, 2)
