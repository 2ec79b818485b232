"""Oracle A: torch-CPU restatement of the reference's op sequence for the hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

This follows, step by step, what the reference executes for one closure
evaluation of the pairwise loss, so that it (i) reproduces the reference's
numbers including its float32 rounding behaviour and (ii) is a fair "port"
CPU baseline: the same LAPACK ``syevd`` calls on the same tensors, and torch
autograd for the backward.

    reference step                                             restated in
    ---------------------------------------------------------  ----------------------
    spd_inv_sqrt          src/sqfa/linalg.py:144-162           whitening_factor
    conjugate_matrix      src/sqfa/linalg.py:19-45             whiten_all_pairs
    generalized_eigenvalues  src/sqfa/linalg.py:48-70          generalized_eigenvalues
    affine_invariant_sq   src/sqfa/distances.py:46-67          affine_invariant_sq
    affine_invariant      src/sqfa/distances.py:70-89          affine_invariant
    _embed_gaussian       src/sqfa/distances.py:141-174        embed_gaussian
    fisher_rao_lower_bound[_sq]  src/sqfa/distances.py:177-237 fisher_rao_lower_bound[_sq]
    closure loss          src/sqfa/_optim.py:88-96             pairwise_loss
    check_distances_valid src/sqfa/_optim.py:16-30             distances_valid
"""
import torch

EPS = 1e-6  # src/sqfa/distances.py:29


def _as_batch(M):
    return M.unsqueeze(0) if M.dim() == 2 else M


def whitening_factor(B):
    """W with W B W^T = I, built exactly like the reference: eigh, scale columns by
    1/sqrt(eigenvalue), transpose (so W is NOT symmetric)."""
    evals, evecs = torch.linalg.eigh(B)
    scaled = evecs * torch.sqrt(1.0 / evals).unsqueeze(-2)
    return scaled.transpose(-2, -1)


def whiten_all_pairs(A, W):
    """(nA,m,m),(nB,m,m) -> (nA,nB,m,m): W_j A_i W_j^T for every ordered pair."""
    A = _as_batch(A)
    W = _as_batch(W)
    left = W.unsqueeze(0) @ A.unsqueeze(1)          # (nA,nB,m,m)
    return left @ W.transpose(-2, -1).unsqueeze(0)


def _squeeze_like_reference(T, nA_was_2d, B_was_2d):
    # conjugate_matrix squeezes dims (0,1) when size 1 (dim 0 only for a 2-D B)
    dims = (0,) if B_was_2d else (0, 1)
    for d in sorted(dims, reverse=True):
        if T.shape[d] == 1:
            T = T.squeeze(d)
    return T


def generalized_eigenvalues(A, B):
    """Spectrum of B^-1 A for all ordered pairs, descending; reference squeeze rules."""
    M = whiten_all_pairs(A, whitening_factor(_as_batch(B)))
    M = _squeeze_like_reference(M, A.dim() == 2, False)
    return torch.linalg.eigvalsh(M).flip(-1)


def affine_invariant_sq(A, B):
    lam = generalized_eigenvalues(A, B)
    return (torch.log(lam) ** 2).sum(-1)


def affine_invariant(A, B):
    return torch.sqrt(affine_invariant_sq(A, B) + EPS)


def embed_gaussian(means, covariances):
    """[[cov + mu mu^T, mu], [mu^T, 1]] -> (C, K+1, K+1)."""
    means = means.unsqueeze(0) if means.dim() == 1 else means
    covariances = _as_batch(covariances)
    C, K = means.shape
    top = covariances + means.unsqueeze(2) * means.unsqueeze(1)
    E = torch.cat(
        [torch.cat([top, means.unsqueeze(2)], dim=2),
         torch.cat([means, means.new_ones(C, 1)], dim=1).unsqueeze(1)],
        dim=1,
    )
    return E


def fisher_rao_lower_bound_sq(stats_A, stats_B):
    EA = embed_gaussian(stats_A["means"], stats_A["covariances"])
    EB = embed_gaussian(stats_B["means"], stats_B["covariances"])
    return affine_invariant_sq(EA, EB) / 2


def fisher_rao_lower_bound(stats_A, stats_B):
    return torch.sqrt(fisher_rao_lower_bound_sq(stats_A, stats_B) + EPS)


def pairwise_loss(D):
    """-mean over the strict lower triangle (src/sqfa/_optim.py:88,94)."""
    C = D.shape[0]
    r, c = torch.tril_indices(C, C, offset=-1)
    return -D[r, c].mean()


def distances_valid(D):
    """(has_nan, has_inf) over the whole matrix, like the reference's row-indexed guard."""
    return bool(torch.isnan(D).any()), bool(torch.isinf(D).any())


def pairwise_loss_and_grad(S, scale=1.0, sqrt_mode=True):
    """One M1 evaluation: S (C,m,m) -> loss, dloss/dS via autograd.
    scale=1 -> affine_invariant; scale=0.5 -> the Calvo-Oller form on embeddings."""
    S = S.detach().clone().requires_grad_(True)
    dsq = affine_invariant_sq(S, S) * scale
    D = torch.sqrt(dsq + EPS) if sqrt_mode else dsq
    loss = pairwise_loss(D)
    (g,) = torch.autograd.grad(loss, S)
    return loss.detach(), g, D.detach()
