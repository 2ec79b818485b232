"""Randomised shape sweep of the C-ABI entry point against the float64 closed-form oracle:
self and cross mode, ragged class counts around the tile edges of every geometry, uniform and
per-pair weights, sqrt and squared distances, float32 and float64, 1..3 shards."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import closed_form

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def spd(rng, n, m):
    X = rng.standard_normal((n, 2 * m + 3, m))
    return np.einsum("cnm,cnk->cmk", X, X) / (2 * m + 3) + 0.05 * np.eye(m)


CASES = []
_rng = np.random.default_rng(2025)
for m in (1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 17, 18, 20, 23, 24, 25, 31, 32, 33, 36, 40, 41, 48, 57, 64):
    for _ in range(int(os.environ.get("SQFA_FUZZ_ROUNDS", "2"))):  # soak runs: SQFA_FUZZ_ROUNDS=30
        self_mode = bool(_rng.integers(0, 2))
        nA = int(_rng.integers(2, 40 if m <= 17 else 14))
        nB = 0 if self_mode else int(_rng.integers(1, 30 if m <= 17 else 10))
        CASES.append((m, nA, nB, bool(_rng.integers(0, 2)), bool(_rng.integers(0, 2)), int(_rng.integers(1, 4)),
                      int(_rng.integers(0, 2))))


@pytest.fixture
def class_factors_always():
    """Force the class factor pass (K0b) on: by default it only runs for launches with >= 1e4-1e5 pairs, far more than
    these cases have (sqfa_airm_options::class_factor_policy, include/sqfa_hip.h: a per-call argument)."""
    from sqfa_amd import _native
    with _native.policies(class_factor=1):
        yield


@pytest.mark.parametrize("m,nA,nB,sqrt_mode,weighted,shards,f64", [c for c in CASES if c[0] >= 9])
def test_random_case_through_class_factor_pass(class_factors_always, m, nA, nB, sqrt_mode, weighted, shards, f64):
    """The same sweep with the factor pass forced on (sizes whose geometry has one: MR >= 12): ragged class counts, cross
    mode (the pass touches the A side only), identity-padded sizes, shards."""
    test_random_case(m, nA, nB, sqrt_mode, weighted, shards, f64)


@pytest.fixture
def mean_metric_factors():
    """The opt-in variant of the factor pass (sqfa_airm_options::mean_metric_policy = 1: columns orthogonalised in the metric
    of the mean class, class_factor_mean_kernel), forced on together with the pass itself."""
    from sqfa_amd import _native
    with _native.policies(class_factor=1, mean_metric=1):
        yield


@pytest.mark.parametrize("m,nA,nB,sqrt_mode,weighted,shards,f64", [c for c in CASES if 9 <= c[0] <= 17 or 25 <= c[0] <= 32])
def test_random_case_through_mean_metric_factor_pass(mean_metric_factors, m, nA, nB, sqrt_mode, weighted, shards, f64):
    """The same sweep through the mean-metric factor pass (sizes that have it): ragged counts, cross mode (the mean is that
    of the A side), identity-padded sizes, shards (every shard forms the same mean)."""
    test_random_case(m, nA, nB, sqrt_mode, weighted, shards, f64)


@pytest.fixture
def regular_rows_only():
    """Keep small launches on the regular lane geometries: by default these cases (a few hundred pairs) run on the
    small-launch rows of configs.hpp wherever one exists (sqfa_airm_options::geometry_policy, include/sqfa_hip.h)."""
    from sqfa_amd import _native
    with _native.policies(geometry=-1):
        yield


@pytest.mark.parametrize("m,nA,nB,sqrt_mode,weighted,shards,f64", [c for c in CASES if (c[0] <= 20 or 25 <= c[0] <= 32) and not c[6]])
def test_random_case_on_regular_rows(regular_rows_only, m, nA, nB, sqrt_mode, weighted, shards, f64):
    """The float32 sizes that have a small-launch row, forced onto their regular row (what C=1000 problems run on)."""
    test_random_case(m, nA, nB, sqrt_mode, weighted, shards, f64)


@pytest.mark.parametrize("m", [5, 8, 9, 12, 16, 17, 19, 20, 29, 32])
def test_small_launch_and_regular_rows_agree(m):
    """Both lane geometries of a padded size evaluate the same problem: results may differ by rounding only; the workspace
    size that holds for every policy covers both."""
    from sqfa_amd import _lib, _native
    lib = _lib.load()
    rng = np.random.default_rng(11 * m)
    C = 30
    A = torch.tensor(spd(rng, C, m), dtype=torch.float32, device=DEV)
    outs = {}
    for mode in (-1, 1):
        # the any-policy size covers the size of either row, whichever is asked for first
        assert lib.sqfa_airm_workspace_bytes(C, 0, m, 0) >= lib.sqfa_airm_workspace_bytes_sharded(C, 0, m, 0, 1, mode) > 0
        with _native.policies(geometry=mode):
            outs[mode] = _native.hip_pair_backend(A, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=1.0,
                                                  shard=(0, 1), want_loss=True, want_grad=True, want_dist=True, want_eig=True)
    reg, small = outs[-1], outs[1]
    assert small["nonfinite"].tolist() == [0, 0]
    assert abs(small["loss"].item() - reg["loss"].item()) <= 2e-6 * abs(reg["loss"].item())
    assert rel_err(small["gradA"].cpu(), reg["gradA"].cpu().numpy()) <= 2e-5
    assert rel_err(small["dist"].cpu(), reg["dist"].cpu().numpy()) <= 1e-5
    assert rel_err(small["eig"].sort(dim=-1).values.cpu(), reg["eig"].sort(dim=-1).values.cpu().numpy()) <= 1e-5
    assert not torch.equal(small["gradA"], reg["gradA"])  # a different geometry did run


@pytest.mark.parametrize("m,nA,nB,sqrt_mode,weighted,shards,f64", CASES)
def test_random_case(m, nA, nB, sqrt_mode, weighted, shards, f64):
    from sqfa_amd import _native
    rng = np.random.default_rng(1000 * m + 10 * nA + nB)
    dtype = torch.float64 if f64 else torch.float32
    A = spd(rng, nA, m)
    B = None if nB == 0 else spd(rng, nB, m)
    nBe = nA if B is None else nB
    W = rng.standard_normal((nA, nBe)) if weighted else None
    scale = 0.5 if m % 2 else 1.0
    Wref = W if W is not None else np.full((nA, nBe), 0.37)
    if B is None and W is not None:
        Wref = np.tril(W + W.T, -1)
    D_ref, gA_ref, gB_ref = closed_form.pairwise(A, B, Wref, scale, sqrt_mode)
    mask = np.tril(np.ones((nA, nBe), bool), -1) if B is None else np.ones((nA, nBe), bool)
    loss_ref = (Wref * D_ref)[mask].sum()
    At = torch.tensor(A, dtype=dtype, device=DEV)
    Bt = None if B is None else torch.tensor(B, dtype=dtype, device=DEV)
    Wt = None if W is None else torch.tensor(W, dtype=dtype, device=DEV)
    loss = 0.0
    gA = torch.zeros_like(At)
    gB = None if Bt is None else torch.zeros_like(Bt)
    D = torch.zeros(nA, nBe, dtype=dtype, device=DEV)
    for r in range(shards):
        out = _native.hip_pair_backend(At, Bt, scale=scale, eps=1e-6, sqrt_mode=sqrt_mode, weights=Wt,
                                       uniform_weight=0.37, shard=(r, shards), want_loss=True, want_grad=True,
                                       want_dist=True, want_eig=False)
        assert out["nonfinite"].tolist() == [0, 0]
        loss += out["loss"].item()
        gA += out["gradA"]
        if gB is not None:
            gB += out["gradB"]
        if shards == 1:
            D = out["dist"]
    ltol, gtol = (1e-10, 1e-8) if f64 else (2e-5, 2e-4)
    assert abs(loss - loss_ref) <= ltol * max(1.0, np.abs(Wref * D_ref)[mask].sum())
    assert rel_err(gA.cpu(), gA_ref) <= gtol
    if gB is not None:
        assert rel_err(gB.cpu(), gB_ref) <= gtol
    if shards == 1:
        assert np.abs(D.cpu().numpy() - D_ref).max() <= (1e-9 if f64 else 5e-5) * max(1.0, D_ref.max())


@pytest.mark.parametrize("K,embed", [(16, True), (16, False), (32, True), (8, True)])
def test_many_random_filter_draws_f32_vs_f64(class_factors_always, K, embed):
    """Rare-event guard: the float32 kernel against the float64 kernel on the SQFA / SecondMomentsSQFA feature matrices
    of many random filter draws (C=300 classes: ~45 000 pairs per draw).  A kernel variant tried in round 2 passed
    every other test and still returned ONE wrong pair in ~90 000 for about one draw in a hundred
    (tools/lodger_check.py); the loss of such a draw is off by 2e-4, sound draws agree to ~2e-7."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import model_cases as mc
    from sqfa_amd import _native
    C, D = 300, 64
    stats = {k: v.to("cuda:0") for k, v in mc.c2_statistics(C=C, D=D).items()}
    P = C * (C - 1) // 2
    m = K + 1 if embed else K
    worst = 0.0
    for seed in range(120 if K <= 16 else 25):
        g = torch.Generator(device="cpu").manual_seed(seed)
        F = torch.randn(K, D, generator=g, dtype=torch.float64).to("cuda:0")
        F = F / F.norm(dim=1, keepdim=True)
        S = torch.einsum("kd,cde,le->ckl", F, stats["covariances"].double(), F) + 0.01 * torch.eye(K, device="cuda:0", dtype=torch.float64)
        mu = stats["means"].double() @ F.T
        if embed:
            E = torch.zeros(C, m, m, dtype=torch.float64, device="cuda:0")
            E[:, :K, :K] = S + mu[:, :, None] * mu[:, None, :]
            E[:, :K, K] = mu
            E[:, K, :K] = mu
            E[:, K, K] = 1
        else:
            E = S + mu[:, :, None] * mu[:, None, :]
        l64, f64 = _native.PairwiseLoss.apply(E, 0.5, 1e-6, True, -1.0 / P, (0, 1), None)
        l32, f32 = _native.PairwiseLoss.apply(E.float(), 0.5, 1e-6, True, -1.0 / P, (0, 1), None)
        assert f64.tolist() == [0, 0] and f32.tolist() == [0, 0]
        worst = max(worst, abs(l32.item() - l64.item()) / abs(l64.item()))
    print(f"K={K} embed={embed}: worst relative loss difference float32 vs float64 over the draws {worst:.2e}")
    assert worst < 3e-6


@pytest.mark.parametrize("m", [12, 16, 17, 24, 32, 33])
def test_unrelated_ill_conditioned_classes_f32(class_factors_always, m):
    """The other end of the input range from the benchmark's similar classes: classes with nothing in common (random
    eigenvectors, eigenvalues log-uniform over a condition number of 1e4), where the pair sweeps need the most rounds.  The class
    factor pass (K0b) and the float32 stop threshold were tuned on benchmark-like data; this pins their accuracy here,
    float32 against the float64 closed form on the same (float32-rounded) matrices."""
    from sqfa_amd import _native
    rng = np.random.default_rng(m)
    C = 48
    Q = np.linalg.qr(rng.standard_normal((C, m, m)))[0]
    ev = np.exp((rng.random((C, m)) - 0.5) * np.log(1e4))
    A = np.einsum("cik,ck,cjk->cij", Q, ev, Q)
    A = (0.5 * (A + A.transpose(0, 2, 1))).astype(np.float32).astype(np.float64)
    W = np.full((C, C), 1.0 / (C * (C - 1) // 2))
    D_ref, g_ref, _ = closed_form.pairwise(A, None, W, 1.0, True)
    mask = np.tril(np.ones((C, C), bool), -1)
    out = _native.hip_pair_backend(torch.tensor(A, dtype=torch.float32, device=DEV), None, scale=1.0, eps=1e-6, sqrt_mode=True,
                                   weights=None, uniform_weight=W[0, 0], shard=(0, 1), want_loss=True, want_grad=True,
                                   want_dist=True, want_eig=False)
    assert out["nonfinite"].tolist() == [0, 0]
    loss_ref = (W * D_ref)[mask].sum()
    e_loss = abs(out["loss"].item() - loss_ref) / abs(loss_ref)
    e_dist = (np.abs(out["dist"].cpu().numpy() - D_ref)[mask] / D_ref[mask]).max()
    e_grad = rel_err(out["gradA"].cpu(), g_ref)
    print(f"m={m}: loss {e_loss:.2e}  distances (max) {e_dist:.2e}  gradient {e_grad:.2e}")
    # measured: loss <= 2.7e-7, distances <= 4.4e-7, gradient <= 2.2e-6 (without the factor pass: 3.5e-7, 4.7e-7, 2.4e-6)
    assert e_loss <= 1e-6 and e_dist <= 2e-6 and e_grad <= 1e-5


@pytest.mark.parametrize("m,f64", [(12, False), (16, False), (17, False), (20, False), (32, False), (33, False), (48, False), (64, False),
                                   (16, True), (17, True), (20, True), (64, True),
                                   (40, False), (24, True), (32, True), (33, True)])  # last row: the 2-D lane layouts (float64 m=48 has no pass)
def test_class_factor_pass_on_and_off_agree(m, f64):
    """K0b replaces the Cholesky factor of each A-side class by another factor of the same matrix: the result may only
    move by rounding.  Both policies against each other (and the control's return value)."""
    from sqfa_amd import _lib, _native
    lib = _lib.load()
    rng = np.random.default_rng(7 * m)
    C = 40 if m <= 33 else 12
    dtype = torch.float64 if f64 else torch.float32
    A = torch.tensor(spd(rng, C, m), dtype=dtype, device=DEV)
    outs = {}
    for mode in (-1, 1):
        with _native.policies(class_factor=mode):
            outs[mode] = _native.hip_pair_backend(A, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=1.0,
                                                  shard=(0, 1), want_loss=True, want_grad=True, want_dist=True, want_eig=False)
    off, on = outs[-1], outs[1]
    assert on["nonfinite"].tolist() == [0, 0]
    ltol, gtol = (1e-12, 1e-10) if f64 else (1e-6, 2e-5)
    assert abs(on["loss"].item() - off["loss"].item()) <= ltol * abs(off["loss"].item())
    assert rel_err(on["gradA"].cpu(), off["gradA"].cpu().numpy()) <= gtol
    assert not torch.equal(on["gradA"], off["gradA"])  # the pass did run (a different factor rounds differently)


@pytest.mark.parametrize("m,f64", [(12, False), (16, False), (17, False), (32, False), (16, True), (17, True), (29, True)])
def test_mean_metric_factor_pass_on_and_off_agree(m, f64):
    """The mean-metric pass hands K1 L_i V with another orthogonal V: still a factor of the same matrix, so the result
    may only move by rounding -- also for classes that share an ill-conditioned mean (where it changes the factors most)."""
    from sqfa_amd import _native
    g = torch.Generator().manual_seed(3 * m)
    C = 40
    dtype = torch.float64 if f64 else torch.float32
    q, _ = torch.linalg.qr(torch.randn(m, m, generator=g, dtype=torch.float64))
    root = (q * torch.logspace(0, 1.5, m, dtype=torch.float64)) @ q.T             # cond(Sbar) = 1e3
    E = torch.randn(C, m, m, generator=g, dtype=torch.float64) * 0.3 / m ** 0.5
    M = torch.eye(m, dtype=torch.float64) + 0.5 * (E + E.transpose(1, 2))
    A = (root @ (M @ M.transpose(1, 2)) @ root).to(dtype).to(DEV)
    outs = {}
    for mode in (0, 1):
        with _native.policies(class_factor=1, mean_metric=mode):
            outs[mode] = _native.hip_pair_backend(A, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=1.0,
                                                  shard=(0, 1), want_loss=True, want_grad=True, want_dist=True, want_eig=False)
    off, on = outs[0], outs[1]
    assert on["nonfinite"].tolist() == [0, 0]
    ltol, gtol = (1e-12, 1e-9) if f64 else (1e-6, 3e-5)                              # gradients of cond-1e3 classes: cond * eps
    assert abs(on["loss"].item() - off["loss"].item()) <= ltol * abs(off["loss"].item())
    assert rel_err(on["gradA"].cpu(), off["gradA"].cpu().numpy()) <= gtol
    assert not torch.equal(on["gradA"], off["gradA"])  # the other pass did run


@pytest.mark.parametrize("m", [16, 17, 33])
def test_non_spd_class_through_class_factor_pass(class_factors_always, m):
    """A class that is not positive definite (Cholesky breaks down part-way: some columns of its factor are NaN, some are not)
    must come out of the factor pass as NaN distances for exactly the pairs that touch it -- flagged, never a hang or a fault,
    and without contaminating the other classes."""
    from sqfa_amd import _native
    rng = np.random.default_rng(m)
    C = 20
    A = spd(rng, C, m)
    bad = 7
    w, Q = np.linalg.eigh(A[bad])
    w[m // 2] = -0.3                                    # indefinite: the pivot goes negative half-way through the factorisation
    A[bad] = (Q * w) @ Q.T
    out = _native.hip_pair_backend(torch.tensor(A, dtype=torch.float32, device=DEV), None, scale=1.0, eps=1e-6, sqrt_mode=True,
                                   weights=None, uniform_weight=1.0, shard=(0, 1), want_loss=True, want_grad=True,
                                   want_dist=True, want_eig=False)
    assert out["nonfinite"].tolist()[0] == C - 1
    D = out["dist"].cpu().numpy()
    touched = np.zeros((C, C), bool)
    touched[bad, :] = touched[:, bad] = True
    off = ~np.eye(C, dtype=bool)
    assert np.isnan(D[touched & off]).all() and np.isfinite(D[~touched & off]).all()
    good = A.copy()
    good[bad] = A[0]
    D_ref, _, _ = closed_form.pairwise(good, None, np.ones((C, C)), 1.0, True)
    keep = ~touched & off
    assert np.abs(D[keep] - D_ref[keep]).max() <= 5e-5 * D_ref[keep].max()
