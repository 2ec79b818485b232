"""Pin the CPU oracles against golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import closed_form, reference_path

G1 = load_golden("g1_airm_self.npz")
G1X = load_golden("g1x_airm_cross.npz")
G2 = load_golden("g2_fisher_rao.npz")

G1_CASES = [tuple(c) for c in G1["cases"]]
G1X_CASES = [tuple(c) for c in G1X["cases"]]
G2_CASES = [tuple(c) for c in G2["cases"]]


@pytest.mark.parametrize("C,m", G1_CASES)
def test_reference_path_self_f64(C, m):
    key = f"C{C}_m{m}"
    S = torch.tensor(G1[f"{key}_S"])
    loss, g, D = reference_path.pairwise_loss_and_grad(S)
    assert rel_err(D, G1[f"{key}_d_f64"]) < 1e-12
    assert abs(float(loss) - float(G1[f"{key}_loss_f64"])) < 1e-12 * max(1, abs(float(loss)))
    assert rel_err(g, G1[f"{key}_grad_f64"]) < 1e-9
    loss, g, D = reference_path.pairwise_loss_and_grad(S, sqrt_mode=False)
    assert rel_err(D, G1[f"{key}_dsq_f64"]) < 1e-12
    assert rel_err(g, G1[f"{key}_grad_sq_f64"]) < 1e-9


@pytest.mark.parametrize("C,m", G1_CASES)
def test_reference_path_self_f32_matches_reference_f32(C, m):
    # same op sequence in float32 -> same rounding as the reference's float32 run
    key = f"C{C}_m{m}"
    S = torch.tensor(G1[f"{key}_S"], dtype=torch.float32)
    loss, g, D = reference_path.pairwise_loss_and_grad(S)
    assert rel_err(D, G1[f"{key}_d_f32"]) < 5e-5
    assert rel_err(g, G1[f"{key}_grad_f32"]) < 5e-3  # f32 autograd through eigh is itself noisy


@pytest.mark.parametrize("C,m", G1_CASES)
def test_closed_form_self(C, m):
    key = f"C{C}_m{m}"
    S = G1[f"{key}_S"]
    loss, g, D = closed_form.closure_loss_and_grad(S)
    assert rel_err(D, G1[f"{key}_d_f64"]) < 1e-10
    assert abs(loss - float(G1[f"{key}_loss_f64"])) < 1e-10 * max(1, abs(loss))
    assert rel_err(g, G1[f"{key}_grad_f64"]) < 1e-8
    loss, g, D = closed_form.closure_loss_and_grad(S, sqrt_mode=False)
    assert rel_err(D, G1[f"{key}_dsq_f64"]) < 1e-10
    assert rel_err(g, G1[f"{key}_grad_sq_f64"]) < 1e-8


@pytest.mark.parametrize("nA,nB,m", G1X_CASES)
def test_cross_batches(nA, nB, m):
    key = f"A{nA}_B{nB}_m{m}"
    A, B, W = G1X[f"{key}_A"], G1X[f"{key}_B"], G1X[f"{key}_W"]
    # torch restatement, incl. the squeeze rules
    lam = reference_path.generalized_eigenvalues(torch.tensor(A), torch.tensor(B))
    assert tuple(lam.shape) == G1X[f"{key}_lam_f64"].shape
    assert rel_err(lam, G1X[f"{key}_lam_f64"]) < 1e-11
    d = reference_path.affine_invariant(torch.tensor(A), torch.tensor(B))
    assert tuple(d.shape) == G1X[f"{key}_d_f64"].shape
    assert rel_err(d, G1X[f"{key}_d_f64"]) < 1e-11
    # closed form incl. weighted gradients
    for sqrt_mode, name in ((True, "d"), (False, "dsq")):
        D, gA, gB = closed_form.pairwise(A, B, W, 1.0, sqrt_mode)
        assert rel_err(D.reshape(G1X[f"{key}_{name}_f64"].shape), G1X[f"{key}_{name}_f64"]) < 1e-10
        assert rel_err(gA, G1X[f"{key}_gA_{name}_f64"]) < 1e-8
        assert rel_err(gB, G1X[f"{key}_gB_{name}_f64"]) < 1e-8
    lam_cf = closed_form.generalized_eigenvalues(A, B)
    assert rel_err(lam_cf.reshape(G1X[f"{key}_lam_f64"].shape), G1X[f"{key}_lam_f64"]) < 1e-10
    # gradient of a weighted sum of the eigenvalues (the reference's op is autograd-transparent)
    gA, gB = closed_form.eigenvalue_weight_gradient(A, B, G1X[f"{key}_Wlam"].reshape(lam_cf.shape))
    assert rel_err(gA, G1X[f"{key}_gA_lam_f64"]) < 1e-8
    assert rel_err(gB, G1X[f"{key}_gB_lam_f64"]) < 1e-8


@pytest.mark.parametrize("C,K", G2_CASES)
def test_fisher_rao(C, K):
    key = f"C{C}_K{K}"
    mu, cov = G2[f"{key}_mu"], G2[f"{key}_cov"]
    st = {"means": torch.tensor(mu, requires_grad=True), "covariances": torch.tensor(cov, requires_grad=True)}
    emb = reference_path.embed_gaussian(st["means"], st["covariances"])
    assert rel_err(emb.detach(), G2[f"{key}_emb_f64"]) < 1e-14
    fr = reference_path.fisher_rao_lower_bound(st, st)
    assert rel_err(fr.detach(), G2[f"{key}_fr_f64"].reshape(fr.shape)) < 1e-11
    loss = reference_path.pairwise_loss(fr)
    gmu, gcov = torch.autograd.grad(loss, (st["means"], st["covariances"]))
    assert rel_err(gmu, G2[f"{key}_gmu_f64"]) < 1e-8
    assert rel_err(gcov, G2[f"{key}_gcov_f64"]) < 1e-8
    # closed form through the embedding
    E = closed_form.embed_gaussian(mu, cov)
    loss_cf, gE, D = closed_form.closure_loss_and_grad(E, scale=0.5)
    assert abs(loss_cf - float(G2[f"{key}_loss_f64"])) < 1e-10
    gmu_cf, gcov_cf = closed_form.embed_gaussian_backward(mu, gE)
    assert rel_err(gmu_cf, G2[f"{key}_gmu_f64"]) < 1e-8
    assert rel_err(gcov_cf, G2[f"{key}_gcov_f64"]) < 1e-8
    loss_cf, gE, D = closed_form.closure_loss_and_grad(E, scale=0.5, sqrt_mode=False)
    assert rel_err(D, G2[f"{key}_frsq_f64"].reshape(D.shape)) < 1e-10
    gmu_cf, gcov_cf = closed_form.embed_gaussian_backward(mu, gE)
    assert rel_err(gmu_cf, G2[f"{key}_gmu_sq_f64"]) < 1e-8
    assert rel_err(gcov_cf, G2[f"{key}_gcov_sq_f64"]) < 1e-8


def test_reference_invariants_hold_for_oracle():
    # properties the reference's own tests assert (tests/test_distances.py:40-97):
    # symmetry, zero diagonal, invariance under inversion
    S = G1["C8_m4_S"]
    D, _, _ = closed_form.pairwise(S, S, None, 1.0, False)
    assert np.allclose(D, D.T, atol=1e-10)
    assert np.allclose(np.diag(D), 0, atol=1e-10)
    Dinv, _, _ = closed_form.pairwise(np.linalg.inv(S), np.linalg.inv(S), None, 1.0, False)
    assert np.allclose(D, Dinv, atol=1e-9)
    Dt = reference_path.affine_invariant_sq(torch.tensor(S), torch.tensor(S)).numpy()
    assert np.allclose(D, Dt, atol=1e-10)
