"""sqfa_amd -- Supervised Quadratic Feature Analysis with an MI355X-native training core.

Same sub-modules and public names as the reference package ``sqfa`` (model, distances,
linalg, statistics, constraints, _optim); the pairwise affine-invariant distance path is
implemented by hand-written HIP kernels behind the C ABI in include/sqfa_hip.h.
"""
from . import _optim, constraints, distances, linalg, model, parallel, statistics  # noqa: F401

__all__ = ["model", "distances", "linalg", "statistics", "constraints", "parallel"]
__version__ = "0.1.0"
