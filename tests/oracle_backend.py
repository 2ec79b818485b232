"""Test infrastructure: a pair backend with the signature of sqfa_amd._native.hip_pair_backend
that evaluates the pairs with the float64 closed-form ORACLE on the CPU.  Substituted for the module attribute
sqfa_amd._native._pair_backend by CPU tests to exercise the host logic (model shell,
fitting loop, tile sharding + all-reduce) without a GPU.  Never used by the product."""
import ctypes

import numpy as np
import torch

from oracle import closed_form
from sqfa_amd import _lib


def tiling(nA, nB, m, dtype_code):
    lib = _lib.load()
    vals = [ctypes.c_int() for _ in range(5)]
    st = lib.sqfa_airm_tiling(nA, nB, m, dtype_code, *[ctypes.byref(v) for v in vals])
    assert st == 0, st
    return [v.value for v in vals]  # tile_i, tile_j, n_tiles_i, n_tiles_j, padded_m


def oracle_pair_backend(A, B, *, scale, eps, sqrt_mode, weights, uniform_weight, shard,
                        want_loss, want_grad, want_dist, want_eig):
    assert abs(eps - closed_form.EPS) < 1e-18
    dt, dev = A.dtype, A.device
    An = A.detach().cpu().double().numpy()
    self_mode = B is None
    Bn = An if self_mode else B.detach().cpu().double().numpy()
    nA, nB, m = An.shape[0], Bn.shape[0], An.shape[-1]
    code = _lib.SQFA_F32 if dt == torch.float32 else _lib.SQFA_F64
    ti, tj, _, _, _ = tiling(nA, 0 if self_mode else nB, m, code)
    # pair mask of this shard
    ii, jj = np.meshgrid(np.arange(nA), np.arange(nB), indexing="ij")
    mask = ((ii // ti + jj // tj) % shard[1]) == shard[0]
    if self_mode:
        mask &= ii > jj
    if weights is None:
        W = np.full((nA, nB), float(uniform_weight))
    else:
        W = weights.detach().cpu().double().numpy().copy()
        if self_mode:
            W = np.tril(W + W.T, -1)
    W = np.where(mask, W, 0.0)
    # like the kernel, a non-SPD class yields NaN distances for every pair that touches it
    badA = np.array([np.linalg.eigvalsh(0.5 * (a + a.T)).min() <= 0 for a in An])
    badB = badA if self_mode else np.array([np.linalg.eigvalsh(0.5 * (b + b.T)).min() <= 0 for b in Bn])
    if badA.any() or badB.any():
        An = An.copy()
        An[badA] = np.eye(m)
        if self_mode:
            Bn = An
        else:
            Bn = Bn.copy()
            Bn[badB] = np.eye(m)
    D, gA, gB = closed_form.pairwise(An, None if self_mode else Bn, W, scale, bool(sqrt_mode))
    D[badA, :] = np.nan
    D[:, badB] = np.nan
    out = {"loss": None, "gradA": None, "gradB": None, "dist": None, "eig": None}
    evaluated = mask
    if want_loss:
        out["loss"] = torch.tensor(float((W * D)[evaluated].sum()), dtype=dt, device=dev)
    if want_grad:
        out["gradA"] = torch.tensor(gA, dtype=dt, device=dev)
        out["gradB"] = None if self_mode else torch.tensor(gB, dtype=dt, device=dev)
    if want_dist:
        Dm = np.where(mask | (mask.T if self_mode else False), D, 0.0)
        if self_mode:
            np.fill_diagonal(Dm, np.sqrt(eps) if sqrt_mode else 0.0)
        out["dist"] = torch.tensor(Dm, dtype=dt, device=dev)
    if want_eig:
        out["eig"] = torch.tensor(closed_form.generalized_eigenvalues(An, Bn), dtype=dt, device=dev)
    bad = D[evaluated]
    out["nonfinite"] = torch.tensor([int(np.isnan(bad).sum()), int(np.isinf(bad).sum())], dtype=torch.int32, device=dev)
    return out


def oracle_eigenvalues_backward(A, B, eig_weights):
    """Stand-in for sqfa_amd._native.hip_eigenvalues_backward on the CPU (the oracle's eig output is
    already descending, so the kernel-order weights are in descending order too)."""
    gA, gB = closed_form.eigenvalue_weight_gradient(A.detach().cpu().double().numpy(), B.detach().cpu().double().numpy(),
                                                    eig_weights.detach().cpu().double().numpy())
    return torch.tensor(gA, dtype=A.dtype, device=A.device), torch.tensor(gB, dtype=B.dtype, device=B.device)
