"""SQFA models with the reference's API (src/sqfa/model.py): ``SecondMomentsSQFA`` and
``SQFA`` with ``fit`` / ``fit_pca`` / ``transform`` / ``transform_scatters`` /
``get_class_distances``.  Orchestration only: the class statistics, the projection
F Sigma_c F^T, the parametrizations and LBFGS stay in PyTorch-ROCm; the pairwise distances
and their backward run on the HIP kernels (distances.py, _native.py).
"""
import torch
import torch.nn as nn
from torch.nn.utils.parametrizations import orthogonal
from torch.nn.utils.parametrize import register_parametrization, remove_parametrizations

from . import _native, distances
from ._optim import fitting_loop
from .constraints import FixedFilters, Identity, Sphere
from .linalg import conjugate_matrix
from .statistics import class_statistics, pca, pca_from_scatter

__all__ = ["SecondMomentsSQFA", "SQFA"]


def __dir__():
    return __all__


_DICT_KEYS = {"means", "covariances"}


def _check_statistics(data_statistics, needs_dict=False):
    """Accept a (C,D,D) tensor-like or a dict with 'means' and 'covariances'
    (reference: model.py:56-94; same exception types)."""
    if isinstance(data_statistics, dict):
        missing = _DICT_KEYS - set(data_statistics.keys())
        if missing:
            raise ValueError(
                f"`data_statistics` dictionary must contain the keys {_DICT_KEYS}. Missing keys: {missing}"
            )
        return
    if not hasattr(data_statistics, "shape"):
        raise TypeError(
            "`data_statistics` must be either a dict with 'means' and 'covariances' "
            "or a torch.Tensor of shape (n_classes, n_dim, n_dim)."
        )
    if needs_dict:
        raise TypeError(
            "`data_statistics` must be a dictionary with 'means' and 'covariances' when `needs_dict` is True."
        )


_scatter_cache = {}   # (id(cov), id(mu)) -> (weakref cov, weakref mu, versions, second moments); entries die with their inputs


def _stats_to_scatter(statistics):
    """Second-moment matrices from either input form (reference: model.py:24-53).  For a dict the (C,D,D) sum
    cov + mu mu^T is formed ONCE per pair of input tensors (and versions) and the same tensor object is handed out
    again: the reference recomputes it inside every call, and a fresh tensor per call would also make
    every get_class_distances / transform_scatters call re-check the symmetry of (C,D,D) values."""
    if not isinstance(statistics, dict):
        return statistics
    _check_statistics(statistics)
    mu, cov = statistics["means"], statistics["covariances"]
    if not (torch.is_tensor(mu) and torch.is_tensor(cov)) or mu.requires_grad or cov.requires_grad:
        return cov + mu[:, :, None] * mu[:, None, :]
    import weakref
    key = (id(cov), id(mu))
    hit = _scatter_cache.get(key)
    if hit is not None and hit[0]() is cov and hit[1]() is mu and hit[2] == (cov._version, mu._version):
        return hit[3]
    out = cov + mu[:, :, None] * mu[:, None, :]
    drop = lambda _r, k=key: _scatter_cache.pop(k, None)
    try:
        _scatter_cache[key] = (weakref.ref(cov, drop), weakref.ref(mu, drop), (cov._version, mu._version), out)
    except TypeError:
        pass
    while len(_scatter_cache) > 4:            # a handful of live statistics at most (each entry holds C*D*D values)
        _scatter_cache.pop(next(iter(_scatter_cache)))
    return out


class SecondMomentsSQFA(nn.Module):
    """smSQFA: filters maximising the mean pairwise affine-invariant distance between the
    class second-moment matrices of the filtered data (reference: model.py:97-444)."""

    def __init__(self, n_dim, feature_noise=0, n_filters=2, filters=None, distance_fun=None,
                 constraint="sphere"):
        super().__init__()
        if filters is None:
            filters = torch.randn(n_filters, n_dim)
        else:
            filters = torch.as_tensor(filters, dtype=torch.float32)
        if filters.shape[0] > filters.shape[1]:
            raise ValueError("Number of filters must be less than or equal to the data dimension.")
        self.filters = nn.Parameter(filters)
        # the reference sizes this from n_filters even when `filters` is given;
        # sizing it from the filters is the superset behaviour
        self.register_buffer("noise_mat", self._noise_matrix(feature_noise, filters.shape[0]))
        self.distance_fun = distances.affine_invariant if distance_fun is None else distance_fun
        self.constraint = constraint
        self._add_constraint(constraint)
        self.pair_shard = None   # optional sqfa_amd.parallel.PairShard for multi-GPU fits
        self.class_shard = None  # optional sqfa_amd.parallel.ClassShard: statistics hold local classes only

    @staticmethod
    def _noise_matrix(feature_noise, k):
        return torch.as_tensor(feature_noise, dtype=torch.float32) * torch.eye(k)

    # ------------------------------------------------------------------ transforms
    def transform_scatters(self, data_scatters):
        """(C,D,D) scatter matrices -> (C,K,K) feature scatters F S F^T (reference: model.py:172-188).
        float32 GPU inputs go through the single-pass streaming kernel (scatters are read from
        HBM once per closure, forward and backward together); other inputs use the torch
        expression of conjugate_matrix."""
        filters = self.filters
        native = _native.project_scatters(data_scatters, filters)
        if native is not None:
            return native.squeeze(0) if native.shape[0] == 1 else native
        return conjugate_matrix(data_scatters, filters)

    def transform(self, data_points):
        """(N,D) points -> (N,K) features (reference: model.py:222-237)."""
        return data_points @ self.filters.T

    # ------------------------------------------------------------------ distances
    def _feature_scatters(self, data_statistics, regularized):
        S = self.transform_scatters(_stats_to_scatter(data_statistics))
        if regularized:
            S = S + self.noise_mat[None]
        return S

    def get_class_distances(self, data_statistics, regularized=False):
        """(C,C) pairwise distances between the class feature scatters
        (reference: model.py:190-220)."""
        S = self._feature_scatters(data_statistics, regularized)
        return self.distance_fun(S, S)

    # hooks used by the fitting loop ------------------------------------------------------
    def _prepare_statistics(self, data_statistics):
        # a dict is reduced to second moments once instead of inside every closure; symmetric float32 GPU
        # statistics of many classes are also packed once into their lower block triangle: every closure then streams
        # ~52 % of the bytes (sqfa_project_scatters_packed; _native.prepare_packed decides and keeps the packed copy)
        scatters = _stats_to_scatter(data_statistics)
        _native.prepare_packed(scatters, self.filters.shape[0])
        return scatters

    def _fused_input(self, prepared):
        return self._feature_scatters(prepared, True)

    _fused_kind = "spd"

    def _has_fused_closure(self):
        """True when distance_fun is the native operator this model's fused closure evaluates."""
        spec = distances.fused_spec(self.distance_fun)
        return spec is not None and spec[0] == self._fused_kind

    # the whole closure as ONE autograd node (8 / 11 launches instead of ~40): _native.FusedClosure
    SINGLE_NODE_CLOSURE = True

    def _single_node_inputs(self, prepared, allow_class_shard=False):
        """(raw parameter, scatters, means, sphere?) when the closure can run as one node: a single
        Sphere or Identity parametrization on the filters, supported shapes; class-sharded statistics
        only for the fitting loop's staged (graph) closure, which places the collectives itself."""
        if not self.SINGLE_NODE_CLOSURE or (self.class_shard is not None and not allow_class_shard):
            return None
        plist = getattr(getattr(self, "parametrizations", None), "filters", None)
        if plist is None or len(plist) != 1 or type(plist[0]) not in (Sphere, Identity):
            return None
        raw = plist.original
        if isinstance(prepared, dict):
            scatters, means = prepared["covariances"], prepared["means"]
        else:
            scatters, means = prepared, None
        if not torch.is_tensor(scatters) or scatters.dim() != 3 or scatters.dtype != raw.dtype:
            return None
        if not _native.fused_closure_supported(raw, scatters, means):
            return None
        return raw, scatters, means, type(plist[0]) is Sphere

    def _fused_closure_loss(self, prepared):
        """(loss, flags) through one fused loss+gradient launch, or None when the model's
        distance_fun is not a native affine-invariant operator."""
        if not self._has_fused_closure():
            return None
        spec = distances.fused_spec(self.distance_fun)
        _, scale, sqrt_mode = spec
        single = self._single_node_inputs(prepared)
        if single is not None:
            raw, scatters, means, sphere = single
            C = scatters.shape[0]
            weight = -1.0 / (C * (C - 1) // 2)
            shard, reducer = (0, 1), None
            if self.pair_shard is not None:
                shard, reducer = self.pair_shard.shard, self.pair_shard.reduce
            noise = self._noise_scalar()
            if noise is None:
                return self._fused_closure_loss_chain(prepared, scale, sqrt_mode)
            return _native.FusedClosure.apply(raw, scatters, means, noise, scale, sqrt_mode, weight, shard, reducer, sphere)
        return self._fused_closure_loss_chain(prepared, scale, sqrt_mode)

    def _noise_scalar(self):
        """feature_noise as a host scalar when noise_mat is (still) noise * I, else None; read back once
        per buffer state, never inside a graph capture."""
        key = (self.noise_mat.data_ptr(), self.noise_mat._version, tuple(self.noise_mat.shape))
        cached = getattr(self, "_noise_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        nm = self.noise_mat.detach()
        value = float(nm[0, 0])
        if not bool(torch.equal(nm, value * torch.eye(nm.shape[0], dtype=nm.dtype, device=nm.device))):
            value = None
        self._noise_cache = (key, value)
        return value

    def _fused_closure_loss_chain(self, prepared, scale, sqrt_mode):
        """The same loss as a chain of autograd nodes (parametrization -> projection -> noise ->
        [embedding] -> PairwiseLoss): any parametrization, class-sharded statistics, odd filter counts."""
        S = self._fused_input(prepared)
        if self.class_shard is not None:
            S = self.class_shard.gather(S)
        C = S.shape[0]
        weight = -1.0 / (C * (C - 1) // 2)
        shard, reducer = (0, 1), None
        if self.pair_shard is not None:
            shard, reducer = self.pair_shard.shard, self.pair_shard.reduce
        return _native.PairwiseLoss.apply(S, scale, distances.EPSILON, sqrt_mode, weight, shard, reducer)

    def _sync_gradients(self):
        """Called by the fitting loop after backward: sums the filter gradient over class shards."""
        if self.class_shard is not None:
            self.class_shard.reduce_gradients(self.parameters())

    def _n_classes_total(self, prepared):
        if self.class_shard is not None:
            return self.class_shard.n_classes
        return prepared["means"].shape[0] if isinstance(prepared, dict) else prepared.shape[0]

    # ------------------------------------------------------------------ fitting
    def fit_pca(self, X=None, data_statistics=None):
        """Set the filters to PCA directions (of X, or via ``pca_from_scatter``)
        (reference: model.py:239-268)."""
        if X is None and data_statistics is None:
            raise ValueError("Either X or data_statistics must be provided.")
        k = self.filters.shape[0]
        if data_statistics is None:
            components = pca(X, k)
        elif self.class_shard is not None:
            # statistics hold the local classes only: average the scatters over all ranks first
            local_sum = _stats_to_scatter(data_statistics).sum(dim=0)
            torch.distributed.all_reduce(local_sum, group=self.class_shard.group)
            if k > local_sum.shape[-1]:
                raise ValueError("n_components must be less than or equal to n_dim.")
            components = pca(local_sum / self.class_shard.n_classes, n_components=k)
        else:
            components = pca_from_scatter(_stats_to_scatter(data_statistics), k)
        self._replace_filters(components)

    def _replace_filters(self, new_filters, n_row_fixed=0):
        remove_parametrizations(self, "filters")
        self.filters = nn.Parameter(new_filters)
        self._add_constraint(self.constraint)
        if n_row_fixed:
            register_parametrization(self, "filters", FixedFilters(n_row_fixed=n_row_fixed))

    def fit(self, X=None, y=None, data_statistics=None, max_epochs=300, lr=0.1, estimator="empirical",
            pairwise=False, show_progress=True, return_loss=False, atol=1e-6, **kwargs):
        """Fit the filters with LBFGS (reference: model.py:270-414).  Either ``X, y`` or
        ``data_statistics``; ``pairwise=True`` learns the filters two at a time, holding the
        earlier ones fixed.  Extra keyword arguments go to torch.optim.LBFGS."""
        if data_statistics is None:
            if X is None or y is None:
                raise ValueError("Either data_statistics or X and y must be provided.")
            data_statistics = class_statistics(X, y, estimator=estimator)
        _check_statistics(data_statistics)
        loop = dict(max_epochs=max_epochs, lr=lr, show_progress=show_progress, return_loss=True,
                    atol=atol, **kwargs)
        if not pairwise:
            loss, elapsed = fitting_loop(model=self, data_statistics=data_statistics, **loop)
        else:
            loss, elapsed = self._fit_pairwise(data_statistics, loop)
        return (loss, elapsed) if return_loss else None

    def _fit_pairwise(self, data_statistics, loop):
        k_total = self.filters.shape[0]
        if k_total % 2 != 0:
            raise ValueError("Number of filters must be even for pairwise training.")
        start_filters = self.filters.detach().clone()
        noise_level = self.noise_mat.detach().clone()[0, 0]
        loss = torch.tensor([])
        elapsed = torch.tensor([])
        for stage in range(k_total // 2):
            done = 2 * stage
            learned = self.filters.detach().clone()
            fresh = start_filters[done:done + 2]
            init = fresh.contiguous() if stage == 0 else torch.cat((learned, fresh))
            self._replace_filters(init, n_row_fixed=done)
            self.register_buffer(
                "noise_mat", noise_level * torch.eye(done + 2, dtype=noise_level.dtype, device=noise_level.device))
            stage_loss, stage_time = fitting_loop(model=self, data_statistics=data_statistics, **loop)
            # drop the FixedFilters layer again
            remove_parametrizations(self, "filters")
            self._add_constraint(self.constraint)
            if elapsed.numel() > 0:
                stage_time = stage_time + elapsed[-1]
            loss = torch.cat((loss, stage_loss))
            elapsed = torch.cat((elapsed, stage_time))
        return loss, elapsed

    def _add_constraint(self, constraint="none"):
        """'none' | 'sphere' | 'orthogonal' (reference: model.py:416-431)."""
        if constraint == "none":
            register_parametrization(self, "filters", Identity())
        elif constraint == "sphere":
            register_parametrization(self, "filters", Sphere())
        elif constraint == "orthogonal":
            orthogonal(self, "filters")

    def __dir__(self):
        return ["filters", "noise_mat", "distance_fun", "constraint", "transform_scatters",
                "get_class_distances", "transform", "fit_pca"]


class SQFA(SecondMomentsSQFA):
    """SQFA: uses class means and covariances; the default discriminability measure is the
    Calvo-Oller lower bound of the Fisher-Rao distance (reference: model.py:447-630)."""

    _fused_kind = "gaussian"

    def __init__(self, n_dim, feature_noise=0, n_filters=2, filters=None, distance_fun=None,
                 constraint="sphere"):
        super().__init__(
            n_dim=n_dim, feature_noise=feature_noise, n_filters=n_filters, filters=filters,
            distance_fun=distances.fisher_rao_lower_bound if distance_fun is None else distance_fun,
            constraint=constraint,
        )

    def _feature_statistics(self, data_statistics, regularized):
        cov = self.transform_scatters(data_statistics["covariances"])
        if regularized:
            cov = cov + self.noise_mat[None]
        return {"means": self.transform(data_statistics["means"]), "covariances": cov}

    def get_class_distances(self, data_statistics, regularized=False):
        """(C,C) pairwise distances between the class feature Gaussians
        (reference: model.py:508-546)."""
        if not isinstance(data_statistics, dict):
            raise TypeError("data_statistics must be a dictionary with 'means' and 'covariances' keys.")
        stats = self._feature_statistics(data_statistics, regularized)
        return self.distance_fun(stats, stats)

    def _prepare_statistics(self, data_statistics):
        if not isinstance(data_statistics, dict):
            raise TypeError("data_statistics must be a dictionary with 'means' and 'covariances' keys.")
        _native.prepare_packed(data_statistics["covariances"], self.filters.shape[0])   # see SecondMomentsSQFA._prepare_statistics
        return data_statistics

    def _fused_input(self, prepared):
        return distances.embed_gaussian(self._feature_statistics(prepared, True))

    def fit(self, X=None, y=None, data_statistics=None, max_epochs=300, lr=0.1, estimator="empirical",
            pairwise=False, show_progress=True, return_loss=False, atol=1e-6, **kwargs):
        """Fit with LBFGS (reference: model.py:548-630); ``data_statistics`` must be a dict."""
        if data_statistics is None:
            if X is None or y is None:
                raise ValueError("Either data_statistics or X and y must be provided.")
            data_statistics = class_statistics(X, y, estimator=estimator)
        else:
            _check_statistics(data_statistics, needs_dict=True)
        out = super().fit(data_statistics=data_statistics, max_epochs=max_epochs, lr=lr, estimator=estimator,
                          pairwise=pairwise, show_progress=show_progress, return_loss=True, atol=atol, **kwargs)
        return out if return_loss else None
