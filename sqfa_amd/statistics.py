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


# padded class batches hold at most this many point-matrix elements (x2 for the centred copy)
_BATCH_ELEMENTS = 1 << 27


def class_statistics(points, labels, estimator="empirical"):
    """Per-class mean, covariance and second moment.  Labels are integers 0..C-1
    (reference: statistics.py:8-54; keys 'means', 'covariances', 'second_moments').

    The reference selects each class with a boolean mask inside a Python loop (one
    ``nonzero`` host sync and ~8 small launches per class: launch-bound for many classes).
    Here the points are sorted by label once (stable), the class sizes come back in ONE host
    transfer, and classes of similar size are processed together as zero-padded batches
    ``(G, n_max, D)``: masked centring, one batched GEMM ``Xc^T Xc`` per group, batched OAS
    shrinkage.  Same estimators (centre first, then 1/(n-1)), different summation order."""
    if estimator not in ("empirical", "oas"):
        raise ValueError("estimator must be 'empirical' or 'oas'")
    labels = labels.to(points.device).long()   # the reference accepts float labels too (it compares with ==)
    n_classes = int(labels.max()) + 1
    d = points.shape[-1]
    order = torch.sort(labels, stable=True).indices
    counts_dev = torch.bincount(labels, minlength=n_classes)
    counts = counts_dev.tolist()
    starts_dev = torch.cumsum(counts_dev, 0) - counts_dev
    sorted_points = points[order]
    means = points.new_zeros(n_classes, d)
    covs = points.new_zeros(n_classes, d, d)
    # groups of classes of similar size (little padding), bounded in memory
    by_size = sorted(range(n_classes), key=lambda c: counts[c])
    pos = 0
    while pos < n_classes:
        group = [by_size[pos]]
        pos += 1
        while pos < n_classes:
            n_max = max(counts[by_size[pos]], 1)
            if (len(group) + 1) * n_max * d > _BATCH_ELEMENTS or n_max > 2 * max(counts[group[0]], 16):
                break
            group.append(by_size[pos])
            pos += 1
        cls = torch.as_tensor(group, device=points.device)
        m, c = _batched_moments(sorted_points, starts_dev[cls], counts_dev[cls], max(counts[group[-1]], 1), estimator)
        means[cls] = m
        covs[cls] = c
    second = covs + means[:, :, None] * means[:, None, :]
    return {"means": means, "covariances": covs, "second_moments": second}


def _batched_moments(sorted_points, starts, counts, n_max, estimator):
    """Mean and covariance of G classes stored contiguously in `sorted_points` (class g occupies
    rows starts[g] .. starts[g]+counts[g]), as one zero-padded (G, n_max, D) batch."""
    d = sorted_points.shape[-1]
    dtype = sorted_points.dtype
    r = torch.arange(n_max, device=sorted_points.device)
    mask = r[None, :] < counts[:, None]                                   # (G, n_max)
    idx = starts[:, None] + torch.minimum(r[None, :], (counts[:, None] - 1).clamp(min=0))
    idx = idx.clamp(max=sorted_points.shape[0] - 1)
    keep = mask[:, :, None].to(dtype)
    batch = sorted_points[idx].mul_(keep)                                 # (G, n_max, D), zero padded
    n = counts.to(dtype)
    mean = batch.sum(dim=1) / n[:, None]
    centred = batch.sub_(mean[:, None, :]).mul_(keep)                     # in place: padding stays zero
    S = torch.bmm(centred.transpose(1, 2), centred) / (n - 1)[:, None, None]
    if estimator == "oas":
        # Chen et al. 2010, as oas_covariance below, per class
        tr = torch.diagonal(S, dim1=1, dim2=2).sum(dim=1)
        tr2 = (S * S).sum(dim=(1, 2))
        rho = ((1 - 2 / d) * tr2 + tr * tr) / ((n + 1 - 2 / d) * (tr2 - tr * tr / d))
        rho = torch.clamp(rho, max=1.0)
        eye = torch.eye(d, dtype=dtype, device=S.device)
        S = (1 - rho)[:, None, None] * S + (rho * tr / d)[:, None, None] * eye
    return mean, S


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
    # LAPACK with one thread per CPU this process may actually use: with every hardware thread
    # of a large host under a small cgroup quota the (3072, 3072) case takes 5 s instead of ~2.
    saved = torch.get_num_threads()
    torch.set_num_threads(min(saved, usable_cpus()))
    try:
        _, vecs = torch.linalg.eigh(cov.cpu())
    finally:
        torch.set_num_threads(saved)
    return vecs[:, d - n_components:].flip(1).T.to(cov.device)


def usable_cpus():
    """CPUs this process is entitled to: the cgroup v2 quota when there is one, else the affinity mask."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


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
