"""Class statistics and PCA initialisation (reference: src/sqfa/statistics.py).

Runs once per fit, before the hot path; stays in PyTorch-ROCm as BASELINE.json's north_star
prescribes, but allocates on the input's device (the reference allocates on the CPU only).
"""
import torch

__all__ = ["class_statistics", "oas_covariance", "sample_covariance", "pca", "pca_from_scatter"]


def __dir__():
    return __all__


def sample_covariance(points, assume_centered=False):
    """Sample covariance of (n_points, n_dim) data; 1/(n-1) after centring, 1/n when the data
    is declared centred (reference: statistics.py:97-124)."""
    n = points.shape[0]
    if assume_centered:
        return points.T @ points / n
    centred = points - points.mean(dim=0)
    return centred.T @ centred / (n - 1)


def oas_covariance(points, assume_centered=False):
    """Oracle Approximating Shrinkage estimate, Chen et al. 2010 (reference: statistics.py:57-94)."""
    n, d = points.shape
    S = sample_covariance(points, assume_centered=assume_centered)
    tr = torch.trace(S)
    tr2 = (S * S).sum()
    rho = ((1 - 2 / d) * tr2 + tr * tr) / ((n + 1 - 2 / d) * (tr2 - tr * tr / d))
    rho = torch.clamp(rho, max=1.0)
    target = torch.eye(d, dtype=S.dtype, device=S.device) * (tr / d)
    return (1 - rho) * S + rho * target


def class_statistics(points, labels, estimator="empirical"):
    """Per-class mean, covariance and second moment.  Labels are integers 0..C-1
    (reference: statistics.py:8-54; keys 'means', 'covariances', 'second_moments').

    The reference selects each class with a boolean mask inside a Python loop (one
    ``nonzero`` host sync per class).  Here the points are sorted by label once (stable, so
    every class keeps its original point order and the sums are the same), the class
    boundaries come back in ONE host transfer, and each class is a contiguous slice."""
    if estimator not in ("empirical", "oas"):
        raise ValueError("estimator must be 'empirical' or 'oas'")
    labels = labels.to(points.device)
    n_classes = int(labels.max()) + 1
    d = points.shape[-1]
    order = torch.sort(labels, stable=True).indices
    bounds = torch.cumsum(torch.bincount(labels, minlength=n_classes), 0).tolist()
    sorted_points = points[order]
    means = points.new_zeros(n_classes, d)
    covs = points.new_zeros(n_classes, d, d)
    start = 0
    for c in range(n_classes):
        pts = sorted_points[start:bounds[c]]
        start = bounds[c]
        means[c] = pts.mean(dim=0)
        covs[c] = sample_covariance(pts) if estimator == "empirical" else oas_covariance(pts)
    second = covs + means[:, :, None] * means[:, None, :]
    return {"means": means, "covariances": covs, "second_moments": second}


def pca(points, n_components=None):
    """Leading principal directions as rows, by descending variance (reference: statistics.py:127-160)."""
    n, d = points.shape
    if n_components is None:
        n_components = min(n, d)
    if n_components > d:
        raise ValueError("n_components must be less than or equal to n_dim.")
    cov = sample_covariance(points)
    # One-off (D,D) eigh, run through LAPACK on the host like the reference: eigenvectors are
    # only defined up to sign (and up to a rotation inside degenerate eigenspaces), and
    # rocSOLVER makes different choices, which would start every fit from a different point.
    _, vecs = torch.linalg.eigh(cov.cpu())
    return vecs[:, d - n_components:].flip(1).T.to(cov.device)


def pca_from_scatter(scatters, n_components=None):
    """PCA initialisation from class scatter matrices.  NOTE (reference quirk, SURVEY.md Q1,
    statistics.py:189-190, pinned by its tests/test_training.py:184-188): the mean scatter
    matrix is handed to ``pca`` *as if it were a data matrix*, so the result is the PCA of
    its rows, not its leading eigenvectors.  Reproduced on purpose: it is the starting point
    of every ``fit_pca(data_statistics=...)`` trajectory."""
    d = scatters.shape[-1]
    if n_components is None:
        n_components = d
    if n_components > d:
        raise ValueError("n_components must be less than or equal to n_dim.")
    return pca(scatters.mean(dim=0), n_components=n_components)
