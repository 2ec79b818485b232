"""Filter parametrizations (reference: src/sqfa/constraints.py).  O(K*D) elementwise torch."""
import torch
import torch.nn as nn

__all__ = ["Sphere", "Identity", "FixedFilters"]


def __dir__():
    return __all__


class Sphere(nn.Module):
    """Each filter (row) is scaled to unit Euclidean norm (reference: constraints.py:17-54)."""

    def forward(self, X):
        return X / torch.linalg.vector_norm(X, dim=-1, keepdim=True)

    def right_inverse(self, S):
        return S


class Identity(nn.Module):
    """No constraint; exists so every model has a parametrization (reference: constraints.py:58-92)."""

    def forward(self, X):
        return X

    def right_inverse(self, S):
        return S


class FixedFilters(nn.Module):
    """The first ``n_row_fixed`` rows receive no gradient (reference: constraints.py:95-141)."""

    def __init__(self, n_row_fixed):
        super().__init__()
        self.n_row_fixed = n_row_fixed

    def forward(self, X):
        k = self.n_row_fixed
        return torch.cat([X[:k].detach(), X[k:]], dim=0)

    def right_inverse(self, X):
        return X
