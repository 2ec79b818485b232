"""Oracle B: float64 numpy closed form of the pairwise AIRM loss and its gradient.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Independent of torch autograd.  For a pair (A = S_i, B = S_j) with B = L L^T,
M = L^-1 A L^-T = V diag(lam) V^T and U = L^-T V (so U^T B U = I, U^T A U = diag(lam)):

    d^2 = scale * sum_k log(lam_k)^2            (src/sqfa/distances.py:66, :206)
    d   = sqrt(d^2 + 1e-6)                      (src/sqfa/distances.py:89, :237)
    dlam_k/dA = u_k u_k^T,  dlam_k/dB = -lam_k u_k u_k^T

so for a per-pair weight w (w = -1/P for the closure loss, src/sqfa/_optim.py:94)

    dL/dA += U diag(g) U^T,        g_k = w * dD/dd2 * scale * 2 log(lam_k) / lam_k
    dL/dB += -U diag(g * lam) U^T

This is the formula the HIP kernel implements; here it is
evaluated with LAPACK in float64 so the tests can check the formula itself
against the reference's autograd goldens, and the kernel against the formula.
"""
import numpy as np

EPS = 1e-6


def _pair_terms(A, L_inv):
    M = L_inv @ A @ L_inv.T
    M = 0.5 * (M + M.T)
    lam, V = np.linalg.eigh(M)
    U = L_inv.T @ V
    return lam, U


def pairwise(A, B=None, weights=None, scale=1.0, sqrt_mode=True):
    """All-pairs distances and weighted-sum gradient.

    A (nA,m,m); B (nB,m,m) or None for the symmetric self case (A is B), where
    only i>j pairs are evaluated and mirrored.  weights (nA,nB): L = sum w_ij D_ij
    (self case: only the strict lower triangle of `weights` is used).
    Returns D (nA,nB), gA, gB (gB is None in the self case; gA then holds the
    full gradient wrt the shared batch).
    """
    A = np.asarray(A, dtype=np.float64)
    self_case = B is None
    Bm = A if self_case else np.asarray(B, dtype=np.float64)
    nA, nB = A.shape[0], Bm.shape[0]
    Linv = np.linalg.inv(np.linalg.cholesky(Bm))
    D = np.zeros((nA, nB))
    gA = np.zeros_like(A)
    gB = np.zeros_like(Bm)
    for i in range(nA):
        for j in range(nB):
            if self_case and j >= i:
                continue
            lam, U = _pair_terms(A[i], Linv[j])
            loglam = np.log(lam)
            d2 = scale * np.sum(loglam ** 2)
            if sqrt_mode:
                d = np.sqrt(d2 + EPS)
                dd = 0.5 / d
            else:
                d = d2
                dd = 1.0
            D[i, j] = d
            if self_case:
                D[j, i] = d
            if weights is not None:
                g = weights[i, j] * dd * scale * 2.0 * loglam / lam
                gA[i] += (U * g) @ U.T
                contrib = -(U * (g * lam)) @ U.T
                if self_case:
                    gA[j] += contrib
                else:
                    gB[j] += contrib
    if self_case:
        np.fill_diagonal(D, np.sqrt(EPS) if sqrt_mode else 0.0)
        return D, gA, None
    return D, gA, gB


def closure_loss_and_grad(S, scale=1.0, sqrt_mode=True):
    """-mean_{i>j} D_ij and its gradient wrt S (C,m,m)."""
    S = np.asarray(S, dtype=np.float64)
    C = S.shape[0]
    P = C * (C - 1) // 2
    W = np.full((C, C), -1.0 / P)
    D, g, _ = pairwise(S, None, W, scale, sqrt_mode)
    iu = np.tril_indices(C, -1)
    return -D[iu].mean(), g, D


def generalized_eigenvalues(A, B):
    """(nA,nB,m) descending generalized eigenvalues via Cholesky whitening."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    Linv = np.linalg.inv(np.linalg.cholesky(B))
    out = np.empty((A.shape[0], B.shape[0], A.shape[-1]))
    for i in range(A.shape[0]):
        for j in range(B.shape[0]):
            lam, _ = _pair_terms(A[i], Linv[j])
            out[i, j] = lam[::-1]
    return out


def eigenvalue_weight_gradient(A, B, weights):
    """Gradient of sum_ijk weights[i,j,k] * lam_k(A_i, B_j) (lam descending, as returned by
    generalized_eigenvalues) wrt A and B: dlam/dA = u u^T, dlam/dB = -lam u u^T."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    Linv = np.linalg.inv(np.linalg.cholesky(B))
    gA, gB = np.zeros_like(A), np.zeros_like(B)
    for i in range(A.shape[0]):
        for j in range(B.shape[0]):
            lam, U = _pair_terms(A[i], Linv[j])     # ascending
            w = np.asarray(weights[i, j], dtype=np.float64)[::-1]
            gA[i] += (U * w) @ U.T
            gB[j] += -(U * (w * lam)) @ U.T
    return gA, gB


def embed_gaussian(means, covariances):
    means = np.asarray(means, dtype=np.float64)
    cov = np.asarray(covariances, dtype=np.float64)
    C, K = means.shape
    E = np.empty((C, K + 1, K + 1))
    E[:, :K, :K] = cov + means[:, :, None] * means[:, None, :]
    E[:, :K, K] = means
    E[:, K, :K] = means
    E[:, K, K] = 1.0
    return E


def embed_gaussian_backward(means, gE):
    """Pull a gradient wrt the embedding back to (means, covariances)."""
    K = means.shape[1]
    gcov = gE[:, :K, :K].copy()
    gmu = (gE[:, :K, :K] + np.swapaxes(gE[:, :K, :K], 1, 2)) @ means[:, :, None]
    gmu = gmu[:, :, 0] + gE[:, :K, K] + gE[:, K, :K]
    return gmu, gcov
