"""Discriminability measures usable as a model's ``distance_fun`` (reference:
src/sqfa/distances.py; contract: docs/source/tutorials/distances.md:127-178, 460-491).

The affine-invariant family -- ``affine_invariant[_sq]`` on SPD batches and
``fisher_rao_lower_bound[_sq]`` on Gaussian statistics -- is the hot path and runs on the
hand-written HIP kernels (forward and backward).  The other operators are not defaults of
either model; for GPU tensors their pair-dependent terms run on the native Gaussian pair kernel
(``bhattacharyya`` / ``mahalanobis[_sq]`` / ``hellinger`` / ``fisher_rao_same_cov``: sqfa_gauss_pair_terms) or avoid
the reference's (nA,nB,m,m) tensor (``log_euclidean[_sq]``: per-class logarithms + exact pairwise distances); CPU
tensors keep the reference's torch expression.
"""
import torch

from . import _native
from .linalg import spd_log

__all__ = [
    "affine_invariant_sq",
    "affine_invariant",
    "log_euclidean_sq",
    "log_euclidean",
    "fisher_rao_lower_bound",
    "fisher_rao_lower_bound_sq",
    "bhattacharyya",
    "mahalanobis_sq",
    "mahalanobis",
    "hellinger",
    "fisher_rao_same_cov",
]


def __dir__():
    return __all__


EPSILON = _native.EPSILON


def _batch_of_matrices(M):
    return M[None] if M.dim() == 2 else M


def _batch_of_vectors(v):
    return v[None] if v.dim() == 1 else v


def _squeeze_pairs(D):
    """(nA,nB) -> drop the dims of size 1, as the reference's conjugate_matrix does."""
    if D.shape[1] == 1:
        D = D.squeeze(1)
    if D.shape[0] == 1:
        D = D.squeeze(0)
    return D


def _pair_matrix(A, B, scale, sqrt_mode):
    same = A is B
    A3 = _batch_of_matrices(A)
    if same and A3.shape[0] >= 2:
        D, _flag = _native.PairDistanceMatrix.apply(A3, None, scale, EPSILON, sqrt_mode)
    else:
        D, _flag = _native.PairDistanceMatrix.apply(A3, _batch_of_matrices(B), scale, EPSILON, sqrt_mode)
    return _squeeze_pairs(D)


# marks the callables the fused closure path may replace by a single loss+grad launch:
# name -> (input kind, scale, sqrt_mode)
_FUSED = {}


def _fusable(kind, scale, sqrt_mode):
    def deco(fn):
        _FUSED[fn] = (kind, scale, sqrt_mode)
        return fn
    return deco


def fused_spec(fn):
    """(kind, scale, sqrt_mode) if `fn` is one of the native affine-invariant operators."""
    return _FUSED.get(fn)


@_fusable("spd", 1.0, False)
def affine_invariant_sq(A, B):
    """Squared affine-invariant distance sum_k log^2 lambda_k(A_i, B_j): (nA,m,m),(nB,m,m)
    -> (nA,nB) (reference: src/sqfa/distances.py:46-67)."""
    return _pair_matrix(A, B, 1.0, False)


@_fusable("spd", 1.0, True)
def affine_invariant(A, B):
    """Affine-invariant distance sqrt(. + 1e-6) (reference: src/sqfa/distances.py:70-89)."""
    return _pair_matrix(A, B, 1.0, True)


def embed_gaussian(statistics):
    """Calvo-Oller embedding [[cov + mu mu^T, mu], [mu^T, 1]] of N(mu, cov) into SPD(K+1)
    (reference: src/sqfa/distances.py:141-174).  Differentiable torch glue."""
    mu = _batch_of_vectors(statistics["means"])
    cov = _batch_of_matrices(statistics["covariances"])
    C, K = mu.shape
    E = mu.new_empty(C, K + 1, K + 1)
    E[:, :K, :K] = cov + mu[:, :, None] * mu[:, None, :]
    E[:, :K, K] = mu
    E[:, K, :K] = mu
    E[:, K, K] = 1.0
    return E


_embed_gaussian = embed_gaussian  # reference's private name


def _fisher_rao(statistics_A, statistics_B, sqrt_mode):
    EA = embed_gaussian(statistics_A)
    EB = EA if statistics_A is statistics_B else embed_gaussian(statistics_B)
    return _pair_matrix(EA, EB, 0.5, sqrt_mode)


@_fusable("gaussian", 0.5, False)
def fisher_rao_lower_bound_sq(statistics_A, statistics_B):
    """Calvo & Oller lower bound of the squared Fisher-Rao distance between Gaussians:
    half the squared affine-invariant distance of the embeddings
    (reference: src/sqfa/distances.py:177-207)."""
    return _fisher_rao(statistics_A, statistics_B, False)


@_fusable("gaussian", 0.5, True)
def fisher_rao_lower_bound(statistics_A, statistics_B):
    """sqrt(fisher_rao_lower_bound_sq + 1e-6) (reference: src/sqfa/distances.py:210-237)."""
    return _fisher_rao(statistics_A, statistics_B, True)


# ------------------------------------------------------------------------------------------
# operators outside the hot path (GPU tensors: native Gaussian pair kernel / no pair tensor; CPU tensors: torch)


def log_euclidean_sq(A, B):
    """|| log A_i - log B_j ||_F^2 (reference: src/sqfa/distances.py:92-116)."""
    LA = spd_log(_batch_of_matrices(A))
    LB = spd_log(_batch_of_matrices(B))
    # || log A_i - log B_j ||_F^2 as exact pairwise distances between the flattened logarithms: the
    # reference's (nA,nB,m,m) difference tensor is never formed (C per-class logarithms + one (nA,nB) pass)
    d = torch.cdist(LA.flatten(1)[None], LB.flatten(1)[None], compute_mode="donot_use_mm_for_euclid_dist")[0]
    return torch.squeeze(d * d)


def log_euclidean(A, B):
    """reference: src/sqfa/distances.py:119-138"""
    return torch.sqrt(log_euclidean_sq(A, B) + EPSILON)


def _gaussian_inputs(statistics_A, statistics_B):
    muA = _batch_of_vectors(statistics_A["means"])
    covA = _batch_of_matrices(statistics_A["covariances"])
    muB = _batch_of_vectors(statistics_B["means"])
    covB = _batch_of_matrices(statistics_B["covariances"])
    return muA, covA, muB, covB


def _gauss_pair_terms(statistics_A, statistics_B):
    """(Q, LD, ldA, ldB): Q_ij = delta^T Sbar^-1 delta, LD_ij = logdet Sbar for the pair's mean
    covariance Sbar = (Sigma_i + Sigma_j)/2, and the per-class log-determinants.  GPU tensors go through the native pair kernel
    (one lane group per pair; nothing of size (nA,nB,K,K) is formed); CPU tensors keep the torch
    expression of the reference."""
    muA, covA, muB, covB = _gaussian_inputs(statistics_A, statistics_B)
    if covA.is_cuda and covA.shape[-1] <= _native.GAUSS_MAX_DIM and covA.dtype in (torch.float32, torch.float64):
        same = (statistics_A["means"] is statistics_B["means"]
                and statistics_A["covariances"] is statistics_B["covariances"])
        if same:
            Q, LD = _native.GaussPairTerms.apply(muA, covA, muA, covA, True)
            # Sbar_ii = Sigma_i exactly, so the diagonal IS the per-class log-determinant, from the same
            # arithmetic as the pair terms: D_ii and its (for hellinger 500x amplified) gradient then
            # cancel exactly, as they do in the reference (identical LU of mean_cov and cov on the diagonal)
            ld = torch.diagonal(LD)
            return Q, LD, ld, ld
        Q, LD = _native.GaussPairTerms.apply(muA, covA, muB.to(covA.dtype), covB.to(covA.dtype), False)
        return Q, LD, torch.logdet(covA), torch.logdet(covB)
    mid = 0.5 * (covA[:, None] + covB[None])
    delta = muA[:, None] - muB[None]
    sol = torch.linalg.solve(mid, delta.unsqueeze(-1)).squeeze(-1)
    return (delta * sol).sum(-1), torch.logdet(mid), torch.logdet(covA), torch.logdet(covB)


def bhattacharyya(statistics_A, statistics_B):
    """Bhattacharyya distance between Gaussians (reference: src/sqfa/distances.py:240-280):
    Q/8 + (logdet Sbar - (logdet Sigma_i + logdet Sigma_j)/2)/2."""
    Q, LD, ldA, ldB = _gauss_pair_terms(statistics_A, statistics_B)
    det_term = 0.5 * (LD - 0.5 * (ldA[:, None] + ldB[None]))
    return torch.squeeze(Q / 8 + det_term)


def mahalanobis_sq(statistics_A, statistics_B):
    """Squared Mahalanobis distance under the pair's mean covariance
    (reference: src/sqfa/distances.py:283-332; like the reference, not squeezed)."""
    return _gauss_pair_terms(statistics_A, statistics_B)[0]


def mahalanobis(statistics_A, statistics_B):
    """reference: src/sqfa/distances.py:335-361"""
    return torch.sqrt(mahalanobis_sq(statistics_A, statistics_B) + EPSILON)


def hellinger(statistics_A, statistics_B):
    """reference: src/sqfa/distances.py:364-393"""
    return torch.sqrt(1 - torch.exp(-bhattacharyya(statistics_A, statistics_B)) + EPSILON)


def fisher_rao_same_cov(statistics_A, statistics_B):
    """Exact Fisher-Rao distance for a shared covariance (the pair's mean covariance):
    sqrt(2) * acosh(1 + mahalanobis^2 / 4) (reference: src/sqfa/distances.py:396-432)."""
    d2 = mahalanobis_sq(statistics_A, statistics_B)
    # like the reference, sqrt(2) is a default-dtype tensor (float32-rounded under the float32 default)
    return torch.sqrt(torch.tensor(2.0)).to(d2.device) * torch.acosh(1 + d2 / 4)
