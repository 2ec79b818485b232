"""GPU tests of the model shell on the HIP backend against the reference goldens:
closure values (G3), fit trajectories (G4), in float64 and float32."""
import numpy as np
import pytest
import torch

import model_cases as mc
from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("key", mc.G3_KEYS)
def test_closure_f64(key):
    mc.check_closure(key, torch.float64, DEV, tol_loss=1e-10, tol_grad=1e-7, tol_dist=1e-9)


@pytest.mark.parametrize("key", mc.G3_KEYS)
def test_closure_f32(key):
    """float32 closure against the float64 reference values.  north_star: 1e-5 relative on the loss;
    the gradient bound is tied to the reference's OWN float32-vs-float64 deviation on the same
    closure (golden G3 holds both): max(3e-5, 5 x that deviation), as in test_gpu_parity._tols."""
    ref_dev = rel_err(mc.G3[f"{key}_grad_f32"], mc.G3[f"{key}_grad_f64"])
    mc.check_closure(key, torch.float32, DEV, tol_loss=1e-5, tol_grad=max(3e-5, 5 * ref_dev), tol_dist=2e-5)


@pytest.mark.parametrize("key", mc.G3O_KEYS)
def test_orthogonal_closure_f64(key):
    """constraint="orthogonal" (reference src/sqfa/model.py:416-431 -> torch's orthogonal parametrization) against
    the reference's closure values with the parametrization's `base` buffer taken from the golden (G3O)."""
    mc.check_closure(key, torch.float64, DEV, tol_loss=1e-10, tol_grad=1e-7, tol_dist=1e-9, G3=mc.G3O)


@pytest.mark.parametrize("key", mc.G3O_KEYS_F32)
def test_orthogonal_closure_f32(key):
    ref_dev = rel_err(mc.G3O[f"{key}_grad_f32"], mc.G3O[f"{key}_grad_f64"])
    mc.check_closure(key, torch.float32, DEV, tol_loss=1e-5, tol_grad=max(3e-5, 5 * ref_dev), tol_dist=2e-5, G3=mc.G3O)


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_orthogonal_fit_f64(model_name):
    mc.check_orthogonal_fit(model_name, DEV)


@pytest.mark.parametrize("dname", ["rot", "syn"])
@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
@pytest.mark.parametrize("K,noise", [(2, 1e-3), (4, 1e-2)])
@pytest.mark.parametrize("epochs", [1, 3])
def test_short_fit_trajectories_f64(dname, model_name, K, noise, epochs):
    """rot / K=4 past the first epoch is chaotic in the reference itself: bounds = 2 x the reference's own spread
    under rounding-level perturbations at that point (golden G4D: filters 0.32-0.48 smSQFA, 5e-5-5e-4 SQFA)."""
    tol_loss, tol_filters = 1e-6, 1e-7
    if dname == "rot" and K == 4 and epochs > 1:
        spread_f, spread_l = mc.rot_k4_spread(model_name)
        tol_loss, tol_filters = max(1e-6, 2 * spread_l), max(1e-7, 2 * spread_f)
    mc.check_fit(dname, model_name, K, noise, epochs, DEV, tol_loss=tol_loss, tol_filters=tol_filters)


@pytest.mark.parametrize("dname,model_name,K,noise", [("rot", "smsqfa", 2, 1e-3), ("rot", "sqfa", 2, 1e-3),
                                                       ("syn", "smsqfa", 2, 1e-3), ("syn", "sqfa", 2, 1e-3)])
def test_full_fit_filters_match_reference_f64(dname, model_name, K, noise):
    """north_star: learned filters vs the reference to 1e-5 (float64 criterion):
    same epoch count, per-epoch losses and converged filters."""
    mc.check_fit(dname, model_name, K, noise, 300, DEV, tol_loss=1e-6, tol_filters=1e-5)


def _start_from(model, init):
    """Put the reference's exact initial filters into the raw parameter.  fit_pca here agrees with the
    reference's to ~2e-14 (host LAPACK, thread-count dependent), which is enough to move the end point of the
    ill-posed fits (tools/c1_probe.py): trajectory comparisons start from the golden's own initial filters."""
    init = torch.as_tensor(init, dtype=model.filters.dtype, device=model.filters.device)
    assert rel_err(model.filters.detach().cpu(), init.cpu()) < 1e-8      # fit_pca itself is checked here
    with torch.no_grad():
        model.parametrizations.filters.original.copy_(init)


def _reference_drift(golden_b, key_a, golden_a, key_b):
    """Relative distance between two fits of the REFERENCE itself that differ only by rounding:
    its eigh-whitened distance (golden_a) vs the mathematically identical Cholesky-route
    distance_fun plugged into the reference's own model (golden_b; make_golden.py:g4b/g7b)."""
    return rel_err(golden_b[key_b], golden_a[key_a])


def _ensemble_spread(model_name, opt, reference_filters):
    """Largest distance from `reference_filters` among the reference's own fits started from 8 copies
    of the same initialisation perturbed by 1e-14 relative noise (golden G4c, make_golden.py:g4c):
    3e-4 ... 3.4e-3 with the fixed step, up to 5e-2 with strong_wolfe.  The optimum of the K=4
    fits is nearly flat, so this -- not 1e-5 -- is what "the same filters" can mean here."""
    from conftest import load_golden
    G4C = load_golden("g4c_fit_ensemble.npz")
    return max(rel_err(f, reference_filters) for f in G4C[f"syn_{model_name}_K4_{opt}_filters"])


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_full_fit_flat_orbit_case_f64(model_name):
    """K=4 on the synthetic set converges along the nearly flat orbit F -> G F of the AIRM loss.
    How well-posed "filters to 1e-5" is here is MEASURED on the reference: golden G4b holds the
    reference's own fit with its distance_fun swapped for the same function evaluated through a
    Cholesky whitening (rounding-level difference per evaluation) -- the reference then drifts from
    itself by 6e-4 (smSQFA) / 7e-4 (SQFA) in the learned filters -- and golden G4c an ensemble of
    reference fits from 1e-14-perturbed initialisations (spread up to 1.9e-3 / 3.4e-3).  The GPU fit
    must stay within 2x the largest of those reference-vs-reference distances, match the converged
    loss to 1e-5 and the epoch count to +-6."""
    from conftest import load_golden
    G4B = load_golden("g4b_fit_wellposed.npz")
    stats = mc.fit_stats("syn", torch.float64, DEV)
    model = mc.make_model(model_name, 50, 4, 1e-2, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    _start_from(model, mc.G4[f"syn_{model_name}_K4_e300_init"])
    loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    key = f"syn_{model_name}_K4_e300"
    ref = mc.G4[f"{key}_loss"]
    drift = _reference_drift(G4B, f"{key}_filters", mc.G4, f"syn_{model_name}_K4_cholroute_filters")
    F = model.filters.detach().cpu()
    err = min(rel_err(F, mc.G4[f"{key}_filters"]), rel_err(F, G4B[f"syn_{model_name}_K4_cholroute_filters"]))
    spread = max(drift, _ensemble_spread(model_name, "fixed", mc.G4[f"{key}_filters"]))
    print(f"flat orbit {model_name}: GPU vs reference {err:.2e}; reference vs reference: Cholesky route {drift:.2e}, "
          f"perturbed-init ensemble max {spread:.2e}")
    assert abs(loss[-1].item() - ref[-1]) <= 1e-5 * abs(ref[-1])
    assert abs(len(loss) - len(ref)) <= 6
    assert err <= 2 * spread


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_pairwise_fit_f64(model_name):
    """Pairwise (two filters at a time) training is well-posed: the reference drifts from itself by
    7e-11 (smSQFA) / 1.7e-6 (SQFA) under the Cholesky-route swap (golden G4b), so here the
    north_star criterion applies as stated: learned filters to 1e-5."""
    from conftest import load_golden
    G4B = load_golden("g4b_fit_wellposed.npz")
    stats = mc.fit_stats("syn", torch.float64, DEV)
    model = mc.make_model(model_name, 50, 4, 1e-2, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True, pairwise=True)
    key = f"syn_{model_name}_pairwise_K4"
    ref = mc.G4[f"{key}_loss"]
    drift = _reference_drift(G4B, f"{key}_filters", mc.G4, f"{key}_cholroute_filters")
    err = rel_err(model.filters.detach().cpu(), mc.G4[f"{key}_filters"])
    print(f"pairwise {model_name}: GPU vs reference {err:.2e}; reference vs reference (Cholesky route) {drift:.2e}")
    assert t.shape == loss.shape and (t[1:] >= t[:-1]).all()
    assert len(loss) == len(ref)
    assert abs(loss[-1].item() - ref[-1]) <= 1e-6 * abs(ref[-1])
    assert err <= max(1e-5, 4 * drift)
    assert model.noise_mat.shape == (4, 4) and model.filters.shape == (4, 50)


@pytest.mark.parametrize("pairwise", [False, True])
@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_strong_wolfe_fit_f64(model_name, pairwise):
    """LBFGS keyword arguments are forwarded (reference src/sqfa/_optim.py:78-82):
    line_search_fn="strong_wolfe" against the reference's own strong-Wolfe fits (golden G4b).  The line
    search does not make the K=4 optimum well-posed either (the reference drifts from itself by
    8e-4 / 4.7e-3 under the Cholesky-route swap and by up to 5e-2 over the perturbed-init ensemble of
    golden G4c), so the bound is again 2x the reference's own largest drift; pairwise training is
    well-posed (1e-8 / 3e-4) and gets max(1e-5, 4x its drift)."""
    from conftest import load_golden
    G4B = load_golden("g4b_fit_wellposed.npz")
    tag = "pairwise_K4" if pairwise else "K4"
    stats = mc.fit_stats("syn", torch.float64, DEV)
    model = mc.make_model(model_name, 50, 4, 1e-2, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    _start_from(model, mc.G4[f"syn_{model_name}_K4_e300_init"])
    loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True,
                        pairwise=pairwise, line_search_fn="strong_wolfe")
    key = f"syn_{model_name}_{tag}_wolfe"
    ref = G4B[f"{key}_loss"]
    drift = rel_err(G4B[f"{key}_cholroute_filters"], G4B[f"{key}_filters"])
    F = model.filters.detach().cpu()
    err = min(rel_err(F, G4B[f"{key}_filters"]), rel_err(F, G4B[f"{key}_cholroute_filters"]))
    bound = max(1e-5, 4 * drift) if pairwise else 2 * max(drift, _ensemble_spread(model_name, "wolfe", G4B[f"{key}_filters"]))
    print(f"strong_wolfe {model_name} {tag}: GPU vs reference {err:.2e}; reference vs reference (Cholesky route) {drift:.2e}; "
          f"bound {bound:.2e}; epochs {len(loss)} vs {len(ref)}")
    assert abs(loss[-1].item() - ref[-1]) <= 1e-5 * abs(ref[-1])
    assert abs(len(loss) - len(ref)) <= 3
    assert err <= bound


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_fit_three_epochs_f32(model_name):
    """float32: trajectories separate quickly -- on this (chaotic, K=4) case the reference's OWN
    float32 run is 4.5e-3 away from its float64 run by the third epoch.  Criteria: the first
    epoch (no optimizer history yet) agrees with the reference's float32 run to 1e-5, and the
    three-epoch trajectory stays within three times the reference's own float32-vs-float64
    spread (or 2e-3 relative, whichever is larger) of the float64 trajectory."""
    import sqfa_amd
    key = f"syn_{model_name}_K4_e3_f32"
    stats = {k: v.float() for k, v in mc.fit_stats("syn", torch.float64, DEV).items()}
    cls = sqfa_amd.model.SQFA if model_name == "sqfa" else sqfa_amd.model.SecondMomentsSQFA
    model = cls(n_dim=50, n_filters=4, feature_noise=1e-2).to(DEV)
    model.fit_pca(data_statistics=stats)
    loss, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
    ref32, ref64 = mc.G4[f"{key}_loss"], mc.G4[f"syn_{model_name}_K4_e3_loss"]
    assert abs(loss[0].item() - ref32[0]) < 1e-5 * abs(ref32[0])
    spread = np.abs(ref32 - ref64).max()
    assert np.abs(loss.numpy() - ref64).max() < max(3 * spread, 2e-3 * np.abs(ref64).max())


def test_fit_from_points_on_gpu_and_error_paths():
    import sqfa_amd
    X = torch.tensor(mc.G5["pts_X"], dtype=torch.float32, device=DEV)
    y = torch.tensor(mc.G5["pts_y"], device=DEV)
    model = sqfa_amd.model.SQFA(n_dim=6, feature_noise=0.01, n_filters=2).to(DEV)
    model.fit_pca(X=X)
    loss, t = model.fit(X=X, y=y, max_epochs=5, show_progress=False, return_loss=True)
    assert torch.isfinite(loss).all() and loss[-1] <= loss[0]
    assert model.transform(X).shape == (120, 2)
    bad = sqfa_amd.model.SecondMomentsSQFA(n_dim=6, feature_noise=0.0, n_filters=2).to(DEV)
    S = torch.eye(6, device=DEV).repeat(4, 1, 1)
    S[1] = -S[1]
    with pytest.raises(ValueError, match="NaN"):
        bad.fit(data_statistics=S, max_epochs=2, show_progress=False)


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_c2_config_full_fit_matches_reference_f64(model_name):
    """BASELINE config c2 shape (C=100, n_dim=784, n_filters=8, feature_noise=0.01): a full
    float64 fit on the GPU against the reference's own float64 CPU fit (golden G6): same
    number of epochs, per-epoch losses, and learned filters to 1e-5 (north_star criterion)."""
    from conftest import load_golden
    G6 = load_golden("g6_fit_c2.npz")
    stats = mc.c2_statistics()
    assert np.allclose(stats["covariances"][0, :4, :4].numpy(), G6["check_cov00"], rtol=1e-12)
    stats = {k: v.to(DEV) for k, v in stats.items()}
    model = mc.make_model(model_name, 784, 8, 0.01, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    assert rel_err(model.filters.detach().cpu(), G6[f"{model_name}_init"]) < 1e-9
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    ref = G6[f"{model_name}_loss"]
    assert len(loss) == len(ref)
    assert np.abs(loss.numpy() - ref).max() < 1e-6
    err = rel_err(model.filters.detach().cpu(), G6[f"{model_name}_filters"])
    print(f"c2 {model_name}: {len(loss)} epochs, GPU fit {t[-1].item():.2f} s vs reference CPU "
          f"{float(G6[f'{model_name}_seconds']):.1f} s (8 vCPU build container), filters rel err {err:.2e}")
    assert err < 1e-5


@pytest.mark.parametrize("golden,C,D,K,epochs", [("g6c_fit_c1_early.npz", 10, 784, 4, 5), ("g6c_fit_c1_early.npz", 10, 784, 4, 15),
                                                  ("g7e_fit_c5_early.npz", 100, 3072, 16, 3), ("g7e_fit_c5_early.npz", 100, 3072, 16, 5)])
def test_early_epoch_filters_match_reference_f64(golden, C, D, K, epochs):
    """north_star "filters vs reference to 1e-5" on BASELINE configs 1 and 5 themselves, at points where it is a
    property of the algorithm: the reference's float64 fits stopped after a few epochs (goldens G6c / G7e)."""
    mc.check_early_epochs(golden, C, D, K, epochs, DEV, tol_filters=1e-5)


@pytest.mark.parametrize("epochs", [1, 2])
def test_c5_strong_wolfe_early_epochs_f64(epochs):
    """LBFGS keyword arguments are forwarded (reference src/sqfa/_optim.py:78-82): BASELINE config 5 with
    line_search_fn="strong_wolfe", the reference's learned filters after 1 and 2 epochs (golden G7f; with the line search
    the reference agrees with its own Cholesky-route run to 9e-14 there) to 1e-5.  (The FULL strong-Wolfe fit of this
    configuration is ~11 000 closures, ~8 CPU-hours per reference run: test_c5_config_strong_wolfe_fit_f64 stays skipped.)"""
    mc.check_early_epochs("g7f_fit_c5_wolfe_early.npz", 100, 3072, 16, epochs, DEV, tol_filters=1e-5,
                          line_search_fn="strong_wolfe")


def _c5_full_fit():
    """One float64 fit of the c5 shape from the golden's exact initial filters, shared by the two tests below."""
    from conftest import load_golden
    if "c5" not in _c5_full_fit.__dict__:
        G7 = load_golden("g7_fit_c5.npz")
        stats = mc.c2_statistics(C=100, D=3072)
        assert np.allclose(stats["covariances"][0, :4, :4].numpy(), G7["check_cov00"], rtol=1e-12)
        stats = {k: v.to(DEV) for k, v in stats.items()}
        model = mc.make_model("sqfa", 3072, 16, 0.01, "sphere", torch.float64, DEV)
        model.fit_pca(data_statistics=stats)
        _start_from(model, G7["sqfa_init"])
        loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
        _c5_full_fit.c5 = (loss.numpy(), t[-1].item(), model.filters.detach().cpu().numpy())
    return _c5_full_fit.c5


def test_c5_config_full_fit_prefix_matches_reference_f64():
    """BASELINE config c5 shape (CIFAR-100-shaped: C=100, n_dim=3072, n_filters=16, SQFA): the full float64 fit on the
    GPU against the reference's float64 CPU fit (golden G7).  What is well-posed on this configuration is asserted
    hard: the trajectory prefix (six epochs to 1e-6; per-epoch differences grow ~3x per epoch) -- and, in
    test_early_epoch_filters_match_reference_f64, the learned filters after 3 and 5 epochs to 1e-5.  Past that the
    reference's own fixed-step LBFGS run is chaotic (goldens G7b/G7c/G7d: 1e-14 ... 1e-8 perturbations of ITS initial
    filters move ITS end point by 2e-3 ... 1.2 in the filters, -1.8 ... -5.95 in the final loss): the end point is
    reported below, not asserted here."""
    from conftest import load_golden
    G7 = load_golden("g7_fit_c5.npz")
    loss, seconds, F = _c5_full_fit()
    ref = G7["sqfa_loss"]
    print(f"c5 sqfa: {len(loss)} epochs (reference {len(ref)}), GPU fit {seconds:.2f} s vs reference CPU "
          f"{float(G7['sqfa_seconds']):.1f} s; final loss {loss[-1]:.8f} vs {ref[-1]:.8f}")
    print("   per-epoch |loss - reference|:", np.abs(loss[:len(ref)] - ref[:len(loss)]).round(8))
    assert np.isfinite(loss).all()
    assert np.abs(loss[:6] - ref[:6]).max() < 1e-6
    # hard sanity bound on the END of the run: a fit that diverged after the sixth epoch must not pass on finite
    # losses alone.  The reference's own end points on record -- unperturbed (G7), its Cholesky route (G7b), two 1e-14
    # perturbed starts (G7c), 1e-12 / 1e-10 / 1e-8 perturbed starts (G7d) -- span final losses -5.954 ... -1.808 and
    # 12 ... 15 epochs; the GPU run must end inside that span (1e-3 margin, epochs +-2), at a loss below its start.
    G7B, G7C, G7D = (load_golden(n) for n in ("g7b_fit_c5_wellposed.npz", "g7c_fit_c5_ensemble.npz", "g7d_fit_c5_ensemble2.npz"))
    finals = np.concatenate([[ref[-1], G7B["sqfa_cholroute_loss"][-1]], G7C["sqfa_final_loss"], G7D["sqfa_final_loss"]])
    epochs = np.concatenate([[len(ref), len(G7B["sqfa_cholroute_loss"])], G7C["sqfa_epochs"], G7D["sqfa_epochs"]])
    assert finals.min() - 1e-3 <= loss[-1] <= finals.max() + 1e-3, (loss[-1], finals)
    assert epochs.min() - 2 <= len(loss) <= epochs.max() + 2, (len(loss), epochs)
    assert loss[-1] < loss[0]


@pytest.mark.xfail(strict=False, reason="ensemble membership of a chaotic end point: the reference itself leaves its own "
                                        "1e-14 ensemble under a 1e-12 perturbation (golden G7d); informational")
def test_c5_config_full_fit_end_point_in_reference_ensemble_f64(record_property):
    """Does the GPU fit END inside the reference's own 1e-14-perturbation ensemble (filters within 2x the
    reference-vs-reference drift of 2.2e-3, same 13 epochs)?  Not a property of the algorithm (see above) --
    recorded as a property of the run and allowed to fail."""
    from conftest import load_golden
    G7, G7B, G7C = load_golden("g7_fit_c5.npz"), load_golden("g7b_fit_c5_wellposed.npz"), load_golden("g7c_fit_c5_ensemble.npz")
    loss, _, F = _c5_full_fit()
    ref, ref_b = G7["sqfa_loss"], G7B["sqfa_cholroute_loss"]
    drift = max([rel_err(G7B["sqfa_cholroute_filters"], G7["sqfa_filters"])]
                + [rel_err(f, G7["sqfa_filters"]) for f in G7C["sqfa_filters"]])       # 2.2e-3, 2.1e-3, 1.7e-4
    drift_loss = max([abs(ref_b[-1] - ref[-1])] + [abs(v - ref[-1]) for v in G7C["sqfa_final_loss"]])
    err = min(rel_err(F, G7["sqfa_filters"]), rel_err(F, G7B["sqfa_cholroute_filters"]))
    record_property("c5_end_point_filters_vs_reference", float(err))
    record_property("c5_reference_vs_reference_drift", float(drift))
    record_property("c5_epochs", int(len(loss)))
    print(f"c5 end point: filters vs reference {err:.2e} (reference vs reference {drift:.2e}), epochs {len(loss)} vs {len(ref)}")
    assert len(loss) == len(ref)
    assert err <= 2 * drift
    assert min(abs(loss[-1] - ref[-1]), abs(loss[-1] - ref_b[-1])) <= 2 * drift_loss


def test_c5_config_strong_wolfe_fit_f64():
    """The same configuration with line_search_fn="strong_wolfe" (forwarded to LBFGS,
    src/sqfa/_optim.py:78-82) against the reference's strong-Wolfe fits (golden G7b, both distance
    routes): bound = max(1e-5, 2x the reference-vs-reference drift under that optimizer)."""
    from conftest import load_golden
    G7B = load_golden("g7b_fit_c5_wellposed.npz")
    if "sqfa_wolfe_cholroute_filters" not in G7B:
        # the strong-Wolfe c5 fit is ~11 000 closures (203 epochs; 35 s on the GPU in float64): at ~2.5 s per
        # closure for the reference on the 8-vCPU build container that is ~8 CPU-hours per reference run,
        # so these two goldens were not generated (the K=4 strong-Wolfe goldens of G4b/G4c were)
        pytest.skip("strong-Wolfe c5 goldens not generated (8 CPU-hours per reference run)")
    stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=100, D=3072).items()}
    model = mc.make_model("sqfa", 3072, 16, 0.01, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True,
                        line_search_fn="strong_wolfe")
    ref = G7B["sqfa_wolfe_loss"]
    drift = rel_err(G7B["sqfa_wolfe_cholroute_filters"], G7B["sqfa_wolfe_filters"])
    F = model.filters.detach().cpu()
    err = min(rel_err(F, G7B["sqfa_wolfe_filters"]), rel_err(F, G7B["sqfa_wolfe_cholroute_filters"]))
    print(f"c5 strong_wolfe: {len(loss)} epochs (reference {len(ref)}), {t[-1].item():.2f} s vs "
          f"{float(G7B['sqfa_wolfe_seconds']):.0f} s; filters {err:.2e}; reference vs reference {drift:.2e}; "
          f"final loss {loss[-1].item():.8f} vs {ref[-1]:.8f}")
    assert abs(len(loss) - len(ref)) <= 2
    assert np.abs(loss.numpy()[:3] - ref[:3]).max() < 1e-6
    assert err <= max(1e-5, 2 * drift)


@pytest.mark.parametrize("host_lbfgs", [True, False])
@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_graph_captured_closure_matches_eager(model_name, host_lbfgs, monkeypatch):
    """The HIP-graph closure replays exactly the eager arithmetic: same
    per-epoch losses and filters, with the optimizer state on the host and on the device."""
    import sqfa_amd._optim as opt
    stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=24, D=96).items()}
    monkeypatch.setattr(opt, "HOST_SIDE_LBFGS", host_lbfgs)
    monkeypatch.setattr(opt, "HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT", 8192)  # host_lbfgs=True: the host-side state (the default only with a line search)
    replays = [0]
    original_replay = torch.cuda.CUDAGraph.replay

    def counting_replay(self):
        replays[0] += 1
        return original_replay(self)

    monkeypatch.setattr(torch.cuda.CUDAGraph, "replay", counting_replay)
    runs = {}
    for use_graph in (False, True):
        monkeypatch.setattr(opt, "GRAPH_CLOSURE", use_graph)
        model = mc.make_model(model_name, 96, 4, 0.01, "sphere", torch.float64, DEV)
        model.fit_pca(data_statistics=stats)
        loss, _ = model.fit(data_statistics=stats, max_epochs=6, show_progress=False, return_loss=True)
        runs[use_graph] = (loss.numpy(), model.filters.detach().cpu().numpy())
        if not use_graph:
            assert replays[0] == 0
    assert replays[0] > 20, "the graph path was not taken"
    assert np.abs(runs[True][0] - runs[False][0]).max() < 1e-12
    assert rel_err(runs[True][1], runs[False][1]) < 1e-10


def test_graph_captured_closure_reports_nonfinite(monkeypatch):
    """Validity flags are read after every replay: statistics that turn indefinite AFTER the
    capture still raise the reference's ValueError."""
    import sqfa_amd
    import sqfa_amd._optim as opt
    monkeypatch.setattr(opt, "GRAPH_WARMUP_CLOSURES", 1)
    S = mc.c2_statistics(C=6, D=12)["covariances"].to(torch.float32).to(DEV)
    model = sqfa_amd.model.SecondMomentsSQFA(n_dim=12, feature_noise=0.0, n_filters=2).to(DEV)
    calls = [0]
    original = model._fused_closure_loss

    def poisoned(prepared):
        calls[0] += 1
        return original(prepared)

    model._fused_closure_loss = poisoned
    original_replay = torch.cuda.CUDAGraph.replay

    def replay_on_bad_data(self):
        S[1].neg_()  # the captured graph reads the statistics in place
        return original_replay(self)

    monkeypatch.setattr(torch.cuda.CUDAGraph, "replay", replay_on_bad_data)
    with pytest.raises(ValueError, match="NaN"):
        model.fit(data_statistics=S, max_epochs=3, show_progress=False)
    assert calls[0] >= 2  # warm-up closure + the capture


def test_device_side_compact_lbfgs_matches_torch_lbfgs(monkeypatch):
    """Parameters too large for the host-side optimizer state (> 8192 elements): CompactLBFGS with
    its fused read-backs on the device against torch.optim.LBFGS itself, same closures."""
    import sqfa_amd._optim as opt
    stats = {k: v.to(DEV) for k, v in mc.c2_statistics(C=12, D=2304).items()}
    runs = {}
    for compact in (False, True):
        monkeypatch.setattr(opt, "COMPACT_LBFGS", compact)
        model = mc.make_model("smsqfa", 2304, 4, 0.01, "sphere", torch.float64, DEV)   # 9216 parameters
        model.fit_pca(data_statistics=stats)
        loss, _ = model.fit(data_statistics=stats, max_epochs=4, show_progress=False, return_loss=True)
        runs[compact] = (loss.numpy(), model.filters.detach().cpu().numpy())
    assert np.abs(runs[True][0] - runs[False][0]).max() < 1e-9
    assert rel_err(runs[True][1], runs[False][1]) < 1e-7


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_class_statistics_on_gpu_vs_reference(dtype, tol):
    """SURVEY.md 8f rank 2 on the device: class_statistics / OAS (batched, device-resident) against
    the reference's outputs -- the G5 case and a ragged 1000-class case (golden G5c)."""
    mc.check_class_statistics_vs_reference(DEV, dtype, tol)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("constraint", ["sphere", "none"])
@pytest.mark.parametrize("model_name,C,D,K", [("smsqfa", 24, 96, 4), ("sqfa", 24, 96, 4), ("sqfa", 300, 64, 16),
                                              ("smsqfa", 7, 132, 32), ("sqfa", 5, 40, 8), ("sqfa", 12, 64, 9),
                                              ("smsqfa", 300, 96, 2), ("sqfa", 40, 48, 1), ("smsqfa", 9, 72, 17)])
def test_single_node_closure_matches_autograd_chain(model_name, C, D, K, constraint, dtype):
    """_native.FusedClosure (sphere -> projection -> noise/embedding -> pair loss as ONE autograd node,
    8 / 11 launches) against the chain of autograd nodes it replaces: same loss, flags and gradient
    with respect to the raw filter parameter."""
    import sqfa_amd
    stats = {k: v.to(dtype).to(DEV) for k, v in mc.c2_statistics(C=C, D=D).items()}
    inp = stats if model_name == "sqfa" else stats["covariances"] + stats["means"][:, :, None] * stats["means"][:, None, :]
    model = mc.make_model(model_name, D, K, 0.01, constraint, dtype, DEV)
    prepared = model._prepare_statistics(inp)
    results = []
    for single in (True, False):
        model.SINGLE_NODE_CLOSURE = single
        assert (model._single_node_inputs(prepared) is not None) == single
        model.zero_grad()
        loss, flags = model._fused_closure_loss(prepared)
        (3.0 * loss).backward()                      # a non-unit incoming gradient
        results.append((loss.item(), flags.tolist(), model.parametrizations.filters.original.grad.clone()))
    (l1, f1, g1), (l0, f0, g0) = results
    assert f1 == f0 == [0, 0]
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert abs(l1 - l0) <= tol * abs(l0)
    assert rel_err(g1.cpu(), g0.cpu()) <= (1e-10 if dtype == torch.float64 else 2e-4)


def test_c1_config_full_fit_matches_reference_f64():
    """BASELINE config 1 shape (the README example: 10 classes, n_dim=784, n_filters=4, feature_noise=0.01, SQFA)
    on the synthetic generator: full float64 fit on the GPU against the reference's CPU fit (golden G6b),
    which also holds the reference's own sensitivity on this K=4 optimum (Cholesky-route distance_fun and
    four 1e-14-perturbed initialisations: 1e-3 ... 3.3e-3).  The trajectory is compared from the reference's
    EXACT initial filters (this package's fit_pca agrees with them to 2e-14 -- host LAPACK with a different
    thread count -- and that alone moves the end point of this fit by 3.5e-2, at the chaotic episode around
    epoch 31 where the reference's own runs separate too: the sensitivity is heavy-tailed, see
    tools/c1_probe.py).  Bound: max(1e-5, 2x the reference's largest distance from itself)."""
    from conftest import load_golden
    G = load_golden("g6b_fit_c1.npz")
    stats = mc.c2_statistics(C=10, D=784)
    assert np.allclose(stats["covariances"][0, :4, :4].numpy(), G["check_cov00"], rtol=1e-12)
    stats = {k: v.to(DEV) for k, v in stats.items()}
    model = mc.make_model("sqfa", 784, 4, 0.01, "sphere", torch.float64, DEV)
    model.fit_pca(data_statistics=stats)
    _start_from(model, G["sqfa_init"])
    loss, t = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    ref = G["sqfa_loss"]
    assert len(loss) == len(ref)
    spread = max([rel_err(G["sqfa_cholroute_filters"], G["sqfa_filters"])]
                 + [rel_err(f, G["sqfa_filters"]) for f in G["sqfa_ensemble_filters"]])
    F = model.filters.detach().cpu()
    err = min([rel_err(F, G["sqfa_filters"]), rel_err(F, G["sqfa_cholroute_filters"])])
    print(f"c1 sqfa: {len(loss)} epochs (reference {len(ref)}), GPU fit {t[-1].item():.2f} s vs reference CPU "
          f"{float(G['sqfa_seconds']):.1f} s; filters {err:.2e}; reference vs reference up to {spread:.2e}; "
          f"final loss {loss[-1].item():.8f} vs {ref[-1]:.8f}")
    assert np.abs(loss.numpy()[:3] - ref[:3]).max() < 1e-6
    assert abs(loss[-1].item() - ref[-1]) <= max(1e-6, 2 * abs(G["sqfa_cholroute_loss"][-1] - ref[-1])) + 1e-5 * abs(ref[-1])
    assert err <= max(1e-5, 2 * spread)
