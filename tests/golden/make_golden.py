#!/usr/bin/env python3
"""Regenerate the golden vectors in this directory from the *reference* package.

Run only in the build container, where the reference source tree is mounted
read-only at /root/reference (it never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every array written here is DATA: inputs (generated below with this repo's own
generators, or produced by the reference's own test fixture
``rotated_classes_dataset``) and the outputs the reference computes for them.
Both float64 ("_f64") and float32 ("_f32") reference outputs are stored, the
float32 ones so that tests can use the reference's own f32-vs-f64 deviation as
the yardstick.

Groups:
  G1  per-evaluation affine-invariant distances, loss, dloss/dS      (a1-a6, a9, a11)
  G1x cross batches A != B with a weighted-sum loss                    (a1-a6)
  G2  Calvo-Oller / Fisher-Rao lower bound from (means, covariances)   (a7, a8)
  G3  closure: get_class_distances + loss + grad wrt the raw parameter (a12)
  G4  fit trajectories (loss per epoch, final filters)
  G5  quirks: pca_from_scatter, class_statistics, squeeze shapes, eigenvalue order
"""
import os
import sys

import numpy as np

REF_SRC = "/root/reference/src"
REF_TESTS = "/root/reference/tests"
sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)
sys.path.insert(0, REF_TESTS)

import torch  # noqa: E402

import sqfa  # noqa: E402  (the reference)
from make_examples import rotated_classes_dataset  # noqa: E402  (reference test fixture -> data)

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


# ---------------------------------------------------------------- input generators (ours)
def random_spd(rng, n, m, lo=0.02, hi=3.0):
    """n SPD matrices m x m with log-uniform spectrum in [lo, hi] and Haar eigenvectors."""
    out = np.empty((n, m, m))
    for k in range(n):
        q, r = np.linalg.qr(rng.standard_normal((m, m)))
        q = q * np.sign(np.diag(r))
        ev = np.exp(rng.uniform(np.log(lo), np.log(hi), m))
        out[k] = (q * ev) @ q.T
        out[k] = 0.5 * (out[k] + out[k].T)
    return out


def T(x, dtype):
    return torch.tensor(np.asarray(x), dtype=dtype)


def tril_loss(D):
    C = D.shape[0]
    idx = torch.tril_indices(C, C, offset=-1)
    return -torch.mean(D[idx[0], idx[1]])


# ---------------------------------------------------------------- G1
def g1():
    rng = np.random.default_rng(20251003)
    out = {}
    cases = [(2, 2), (5, 2), (5, 3), (8, 4), (10, 5), (16, 8), (12, 9), (10, 16), (9, 17),
             (6, 32), (5, 33), (3, 1), (7, 6), (20, 12), (4, 24), (3, 48), (3, 64), (37, 16)]
    out["cases"] = np.array(cases)
    for (C, m) in cases:
        S = random_spd(rng, C, m)
        key = f"C{C}_m{m}"
        out[f"{key}_S"] = S
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            St = T(S, dt).requires_grad_(True)
            dsq = sqfa.distances.affine_invariant_sq(St, St)
            d = sqfa.distances.affine_invariant(St, St)
            loss = tril_loss(d)
            (g,) = torch.autograd.grad(loss, St, retain_graph=True)
            loss_sq = tril_loss(dsq)
            (gsq,) = torch.autograd.grad(loss_sq, St)
            out[f"{key}_dsq_{tag}"] = dsq.detach().numpy()
            out[f"{key}_d_{tag}"] = d.detach().numpy()
            out[f"{key}_loss_{tag}"] = loss.detach().numpy()
            out[f"{key}_grad_{tag}"] = g.numpy()
            out[f"{key}_loss_sq_{tag}"] = loss_sq.detach().numpy()
            out[f"{key}_grad_sq_{tag}"] = gsq.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_airm_self.npz"), **out)


def g1x():
    rng = np.random.default_rng(77)
    out = {}
    cases = [(1, 1, 3), (1, 4, 4), (4, 1, 2), (4, 8, 6), (8, 4, 4), (5, 7, 16), (19, 3, 9), (6, 6, 17)]
    out["cases"] = np.array(cases)
    for (nA, nB, m) in cases:
        A = random_spd(rng, nA, m)
        B = random_spd(rng, nB, m)
        W = rng.standard_normal((nA, nB))
        key = f"A{nA}_B{nB}_m{m}"
        out[f"{key}_A"], out[f"{key}_B"], out[f"{key}_W"] = A, B, W
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            At = T(A, dt).requires_grad_(True)
            Bt = T(B, dt).requires_grad_(True)
            Wt = T(W, dt)
            lam = sqfa.linalg.generalized_eigenvalues(At, Bt)
            dsq = sqfa.distances.affine_invariant_sq(At, Bt)
            d = sqfa.distances.affine_invariant(At, Bt)
            out[f"{key}_lam_{tag}"] = lam.detach().numpy()
            out[f"{key}_dsq_{tag}"] = dsq.detach().numpy()
            out[f"{key}_d_{tag}"] = d.detach().numpy()
            for name, mat in (("d", d), ("dsq", dsq)):
                loss = torch.sum(Wt.reshape(mat.shape) * mat)
                gA, gB = torch.autograd.grad(loss, (At, Bt), retain_graph=True)
                out[f"{key}_gA_{name}_{tag}"] = gA.numpy()
                out[f"{key}_gB_{name}_{tag}"] = gB.numpy()
            # gradient of a weighted sum of the eigenvalues themselves (generalized_eigenvalues is
            # autograd-transparent in the reference); weights from a generator of their own so that
            # the arrays above stay bit-identical to the first generation
            Wl = np.random.default_rng(7700 + 100 * nA + 10 * nB + m).standard_normal(tuple(lam.shape))
            out[f"{key}_Wlam"] = Wl
            gA, gB = torch.autograd.grad(torch.sum(T(Wl, dt) * lam), (At, Bt), retain_graph=True)
            out[f"{key}_gA_lam_{tag}"] = gA.numpy()
            out[f"{key}_gB_lam_{tag}"] = gB.numpy()
    np.savez_compressed(os.path.join(HERE, "g1x_airm_cross.npz"), **out)


# ---------------------------------------------------------------- G2
def g2():
    rng = np.random.default_rng(4242)
    out = {}
    cases = [(2, 1), (5, 2), (8, 4), (16, 8), (9, 16), (5, 32), (11, 7)]
    out["cases"] = np.array(cases)
    for (C, K) in cases:
        cov = random_spd(rng, C, K)
        mu = 0.7 * rng.standard_normal((C, K))
        key = f"C{C}_K{K}"
        out[f"{key}_cov"], out[f"{key}_mu"] = cov, mu
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            covt = T(cov, dt).requires_grad_(True)
            mut = T(mu, dt).requires_grad_(True)
            st = {"means": mut, "covariances": covt}
            emb = sqfa.distances._embed_gaussian(st)
            frsq = sqfa.distances.fisher_rao_lower_bound_sq(st, st)
            fr = sqfa.distances.fisher_rao_lower_bound(st, st)
            loss = tril_loss(fr)
            gmu, gcov = torch.autograd.grad(loss, (mut, covt), retain_graph=True)
            loss_sq = tril_loss(frsq)
            gmu_sq, gcov_sq = torch.autograd.grad(loss_sq, (mut, covt))
            out[f"{key}_emb_{tag}"] = emb.detach().numpy()
            out[f"{key}_frsq_{tag}"] = frsq.detach().numpy()
            out[f"{key}_fr_{tag}"] = fr.detach().numpy()
            out[f"{key}_loss_{tag}"] = loss.detach().numpy()
            out[f"{key}_gmu_{tag}"] = gmu.numpy()
            out[f"{key}_gcov_{tag}"] = gcov.numpy()
            out[f"{key}_loss_sq_{tag}"] = loss_sq.detach().numpy()
            out[f"{key}_gmu_sq_{tag}"] = gmu_sq.numpy()
            out[f"{key}_gcov_sq_{tag}"] = gcov_sq.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_fisher_rao.npz"), **out)


# ---------------------------------------------------------------- G3
def synthetic_stats(rng, C, D, rank):
    cov = np.empty((C, D, D))
    for c in range(C):
        A = rng.standard_normal((D, rank)) / np.sqrt(rank)
        cov[c] = A @ A.T + 0.05 * np.eye(D)
    mu = 0.3 * rng.standard_normal((C, D))
    return mu, cov


def g3():
    rng = np.random.default_rng(303)
    out = {}
    rot = rotated_classes_dataset().double().numpy()
    out["rotated_cov"] = rot
    mu_rot = 0.2 * rng.standard_normal((rot.shape[0], rot.shape[1]))
    out["rotated_mu"] = mu_rot
    raw = {K: rng.standard_normal((K, 8)) for K in (1, 2, 3, 4)}
    for K, v in raw.items():
        out[f"raw_filters_K{K}"] = v
    for model_name in ("smsqfa", "sqfa"):
        for constraint in ("sphere", "none", "orthogonal"):
            for noise in (0.0, 1e-3, 1e-2):
                for K in (1, 2, 3, 4):
                    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                        torch.set_default_dtype(dt)
                        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
                        model = cls(n_dim=8, n_filters=K, feature_noise=noise, constraint=constraint)
                        if dt == torch.float64:
                            model = model.double()
                        # assign the raw parameter directly (bypassing right_inverse) so that the
                        # builder can do the same
                        with torch.no_grad():
                            prm = model.parametrizations.filters.original
                            prm.copy_(T(raw[K], dt))
                        stats = {"means": T(mu_rot, dt), "covariances": T(rot, dt)}
                        inp = stats if model_name == "sqfa" else T(rot, dt)
                        key = f"{model_name}_{constraint}_n{noise:g}_K{K}"
                        try:
                            D_ = model.get_class_distances(inp, regularized=True)
                        except Exception as err:  # the reference itself fails on this combo
                            print("  skipped", key, tag, type(err).__name__)
                            continue
                        loss = tril_loss(D_)
                        model.zero_grad()
                        loss.backward()
                        out[f"{key}_filters_{tag}"] = model.filters.detach().numpy()
                        out[f"{key}_D_{tag}"] = D_.detach().numpy()
                        out[f"{key}_loss_{tag}"] = loss.detach().numpy()
                        out[f"{key}_grad_{tag}"] = prm.grad.detach().numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g3_closure.npz"), **out)


# ---------------------------------------------------------------- G4
def g4():
    out = {}
    rot = rotated_classes_dataset().double()
    rng = np.random.default_rng(404)
    mu20, cov20 = synthetic_stats(rng, 20, 50, 10)
    out["syn_mu"], out["syn_cov"] = mu20, cov20
    out["rotated_cov"] = rot.numpy()
    torch.set_default_dtype(torch.float64)
    datasets = {
        "rot": {"means": torch.zeros(5, 8), "covariances": rot},
        "syn": {"means": T(mu20, torch.float64), "covariances": T(cov20, torch.float64)},
    }
    for dname, stats in datasets.items():
        n_dim = stats["covariances"].shape[-1]
        for model_name in ("smsqfa", "sqfa"):
            for (K, noise) in ((2, 1e-3), (4, 1e-2)):
                for epochs in (1, 3, 300):
                    cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
                    model = cls(n_dim=n_dim, n_filters=K, feature_noise=noise).double()
                    model.fit_pca(data_statistics=stats)
                    init = model.filters.detach().clone().numpy()
                    loss, _t = model.fit(data_statistics=stats, max_epochs=epochs,
                                         show_progress=False, return_loss=True)
                    key = f"{dname}_{model_name}_K{K}_e{epochs}"
                    out[f"{key}_init"] = init
                    out[f"{key}_loss"] = loss.numpy()
                    out[f"{key}_filters"] = model.filters.detach().numpy()
        # pairwise run
        for model_name in ("smsqfa", "sqfa"):
            cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
            model = cls(n_dim=n_dim, n_filters=4, feature_noise=1e-2).double()
            model.fit_pca(data_statistics=stats)
            loss, _t = model.fit(data_statistics=stats, max_epochs=300, pairwise=True,
                                 show_progress=False, return_loss=True)
            key = f"{dname}_{model_name}_pairwise_K4"
            out[f"{key}_loss"] = loss.numpy()
            out[f"{key}_filters"] = model.filters.detach().numpy()
    # float32 short trajectories (first 3 epochs) for the f32 criterion of SURVEY 8c
    torch.set_default_dtype(torch.float32)
    for dname, stats in datasets.items():
        st32 = {k: v.float() for k, v in stats.items()}
        n_dim = st32["covariances"].shape[-1]
        for model_name in ("smsqfa", "sqfa"):
            cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
            model = cls(n_dim=n_dim, n_filters=4, feature_noise=1e-2)
            model.fit_pca(data_statistics=st32)
            key = f"{dname}_{model_name}_K4_e3_f32"
            try:
                loss, _t = model.fit(data_statistics=st32, max_epochs=3, show_progress=False, return_loss=True)
            except Exception as err:  # the reference's own float32 run diverges on this dataset
                print("  skipped", key, type(err).__name__)
                continue
            out[f"{key}_loss"] = loss.numpy()
            out[f"{key}_filters"] = model.filters.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "g4_fit.npz"), **out)


# ---------------------------------------------------------------- G5
def g5():
    out = {}
    torch.set_default_dtype(torch.float64)
    rot = rotated_classes_dataset().double()
    out["rotated_cov"] = rot.numpy()
    for K in (1, 2, 4, 8):
        out[f"pca_from_scatter_K{K}"] = sqfa.statistics.pca_from_scatter(rot, K).numpy()
    rng = np.random.default_rng(55)
    X = rng.standard_normal((120, 6)) @ rng.standard_normal((6, 6))
    y = np.repeat(np.arange(4), 30)
    perm = rng.permutation(120)
    X, y = X[perm], y[perm]
    out["pts_X"], out["pts_y"] = X, y
    for est in ("empirical", "oas"):
        st = sqfa.statistics.class_statistics(T(X, torch.float64), torch.tensor(y), estimator=est)
        for k, v in st.items():
            out[f"class_stats_{est}_{k}"] = v.numpy()
    out["pca_X_K3"] = sqfa.statistics.pca(T(X, torch.float64), 3).numpy()
    out["oas_cov"] = sqfa.statistics.oas_covariance(T(X, torch.float64)).numpy()
    out["sample_cov"] = sqfa.statistics.sample_covariance(T(X, torch.float64)).numpy()
    # squeeze shapes
    spd = random_spd(rng, 4, 3)
    shapes = []
    for nA in (1, 4):
        for nB in (1, 4):
            A = T(spd[:nA], torch.float64)
            B = T(spd[:nB], torch.float64)
            shapes.append([nA, nB,
                           len(sqfa.distances.affine_invariant_sq(A, B).shape),
                           len(sqfa.linalg.generalized_eigenvalues(A, B).shape)])
    out["squeeze_shapes"] = np.array(shapes)
    out["squeeze_2d_A"] = np.array(sqfa.distances.affine_invariant_sq(T(spd[0], torch.float64),
                                                                     T(spd[:4], torch.float64)).shape)
    # other linalg helpers
    out["spd"] = spd
    out["spd_sqrt"] = sqfa.linalg.spd_sqrt(T(spd, torch.float64)).numpy()
    out["spd_log"] = sqfa.linalg.spd_log(T(spd, torch.float64)).numpy()
    W = sqfa.linalg.spd_inv_sqrt(T(spd, torch.float64))
    out["spd_inv_sqrt_whitened"] = torch.einsum("nij,njk,nlk->nil", W, T(spd, torch.float64), W).numpy()
    gv, ge = sqfa.linalg.generalized_eigenvectors(T(spd[:3], torch.float64), T(spd[1:4], torch.float64))
    out["gen_eigvec_abs"] = gv.abs().numpy()
    out["gen_eigval"] = ge.numpy()
    # remaining distance_fun operators: values only
    cov = random_spd(rng, 5, 4)
    mu = rng.standard_normal((5, 4))
    out["dist_cov"], out["dist_mu"] = cov, mu
    st = {"means": T(mu, torch.float64), "covariances": T(cov, torch.float64)}
    out["log_euclidean_sq"] = sqfa.distances.log_euclidean_sq(st["covariances"], st["covariances"]).numpy()
    out["log_euclidean"] = sqfa.distances.log_euclidean(st["covariances"], st["covariances"]).numpy()
    for name in ("bhattacharyya", "mahalanobis_sq", "mahalanobis", "hellinger", "fisher_rao_same_cov"):
        out[name] = getattr(sqfa.distances, name)(st, st).numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g5_quirks.npz"), **out)


# ---------------------------------------------------------------- G6
def c2_statistics(C=100, D=784, seed=1234, dtype=torch.float64):
    """BASELINE config c2-shaped synthetic Gaussians (SURVEY 8d generator), regenerated from the
    seed by the tests (the (C,D,D) tensor is too large to store): torch CPU generator only."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = min(D, 128)
    cov = torch.empty(C, D, D, dtype=dtype)
    mu = torch.empty(C, D, dtype=dtype)
    for c0 in range(0, C, 50):
        n = min(50, C - c0)
        A = (torch.randn(n, D, R, generator=g, dtype=torch.float32) / R ** 0.5).to(dtype)
        cov[c0:c0 + n] = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, dtype=dtype)
        mu[c0:c0 + n] = (0.1 * torch.randn(n, D, generator=g, dtype=torch.float32)).to(dtype)
    return {"means": mu, "covariances": cov}


def g6():
    """Full float64 fits of the reference on the c2-shaped configuration (C=100, n_dim=784,
    n_filters=8, feature_noise=0.01, fit_pca init): per-epoch losses and learned filters."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics()
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()
    for model_name in ("smsqfa", "sqfa"):
        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
        model = cls(n_dim=784, n_filters=8, feature_noise=0.01).double()
        model.fit_pca(data_statistics=stats)
        out[f"{model_name}_init"] = model.filters.detach().numpy().copy()
        import time as _t
        t0 = _t.time()
        loss, _t_ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
        out[f"{model_name}_seconds"] = np.array(_t.time() - t0)
        out[f"{model_name}_loss"] = loss.numpy()
        out[f"{model_name}_filters"] = model.filters.detach().numpy()
        print(model_name, "epochs", len(loss), "seconds", float(out[f"{model_name}_seconds"]), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g6_fit_c2.npz"), **out)


def g7():
    """Full float64 fit of the reference on the c5-shaped configuration (CIFAR-100-shaped:
    C=100, n_dim=3072, n_filters=16, feature_noise=0.01, SQFA, fit_pca init)."""
    import time as _t
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=100, D=3072)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()
    model = sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01).double()
    model.fit_pca(data_statistics=stats)
    out["sqfa_init"] = model.filters.detach().numpy().copy()
    t0 = _t.time()
    loss, _t_ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
    out["sqfa_seconds"] = np.array(_t.time() - t0)
    out["sqfa_loss"] = loss.numpy()
    out["sqfa_filters"] = model.filters.detach().numpy()
    print("c5 sqfa epochs", len(loss), "seconds", float(out["sqfa_seconds"]), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g7_fit_c5.npz"), **out)


# ---------------------------------------------------------------- G7b / G4b: how well-posed is "filters to 1e-5"?
def cholesky_fisher_rao(statsA, statsB):
    """The SAME function as the reference's fisher_rao_lower_bound (src/sqfa/distances.py:210-237),
    evaluated through a different but mathematically equivalent route (Cholesky whitening
    instead of the eigh whitening of src/sqfa/linalg.py:159-162), in plain torch with autograd.
    Passed to the REFERENCE's model as distance_fun: the two reference fits then differ only by
    rounding (1e-15 relative per evaluation), which measures how far the reference's own
    trajectory drifts under rounding-level perturbations."""
    EA = sqfa.distances._embed_gaussian(statsA)
    EB = sqfa.distances._embed_gaussian(statsB)
    L = torch.linalg.cholesky(EB)                                   # (nB,m,m)
    Y = torch.linalg.solve_triangular(L[None], EA[:, None], upper=False)       # L^-1 A
    M = torch.linalg.solve_triangular(L[None], Y.transpose(-1, -2), upper=False)  # L^-1 A L^-T
    M = 0.5 * (M + M.transpose(-1, -2))
    lam = torch.linalg.eigvalsh(M)
    return torch.sqrt(0.5 * torch.sum(torch.log(lam) ** 2, dim=-1) + 1e-6)


def cholesky_affine_invariant(A, B):
    """Same idea for SecondMomentsSQFA: affine_invariant (src/sqfa/distances.py:70-89) via Cholesky."""
    L = torch.linalg.cholesky(B)
    Y = torch.linalg.solve_triangular(L[None], A[:, None], upper=False)
    M = torch.linalg.solve_triangular(L[None], Y.transpose(-1, -2), upper=False)
    M = 0.5 * (M + M.transpose(-1, -2))
    lam = torch.linalg.eigvalsh(M)
    return torch.sqrt(torch.sum(torch.log(lam) ** 2, dim=-1) + 1e-6)


def _fit_record(out, key, model, stats, **fit_kwargs):
    import time as _t
    t0 = _t.time()
    loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True, **fit_kwargs)
    out[f"{key}_seconds"] = np.array(_t.time() - t0)
    out[f"{key}_loss"] = loss.numpy()
    out[f"{key}_filters"] = model.filters.detach().numpy()
    print(key, "epochs", len(loss), "seconds", float(out[f"{key}_seconds"]), "final", float(loss[-1]), flush=True)


def g4b():
    """syn (C=20, D=50) K=4 fits of the reference, float64: (i) with the Cholesky-route distance
    (reference-vs-reference drift of the fixed-step LBFGS trajectory), (ii) with
    line_search_fn="strong_wolfe" (kwargs are forwarded to LBFGS, src/sqfa/_optim.py:78-82),
    where the trajectory is well-posed, also for pairwise training."""
    out = {}
    torch.set_default_dtype(torch.float64)
    g4_data = np.load(os.path.join(HERE, "g4_fit.npz"))
    stats = {"means": T(g4_data["syn_mu"], torch.float64), "covariances": T(g4_data["syn_cov"], torch.float64)}
    for model_name in ("smsqfa", "sqfa"):
        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
        alt = cholesky_fisher_rao if model_name == "sqfa" else cholesky_affine_invariant
        for pairwise in (False, True):
            tag = "pairwise_K4" if pairwise else "K4"
            model = cls(n_dim=50, n_filters=4, feature_noise=1e-2, distance_fun=alt).double()
            model.fit_pca(data_statistics=stats)
            _fit_record(out, f"syn_{model_name}_{tag}_cholroute", model, stats, pairwise=pairwise)
            model = cls(n_dim=50, n_filters=4, feature_noise=1e-2).double()
            model.fit_pca(data_statistics=stats)
            _fit_record(out, f"syn_{model_name}_{tag}_wolfe", model, stats, pairwise=pairwise,
                        line_search_fn="strong_wolfe")
            model = cls(n_dim=50, n_filters=4, feature_noise=1e-2, distance_fun=alt).double()
            model.fit_pca(data_statistics=stats)
            _fit_record(out, f"syn_{model_name}_{tag}_wolfe_cholroute", model, stats, pairwise=pairwise,
                        line_search_fn="strong_wolfe")
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g4b_fit_wellposed.npz"), **out)


def g7b():
    """c5-shaped configuration (C=100, n_dim=3072, n_filters=16, SQFA, float64, fit_pca init):
    (i) the reference's fit with the Cholesky-route distance_fun (drift of the reference against
    itself under rounding-level differences; compare with G7), (ii) the reference's fit with
    line_search_fn="strong_wolfe", (iii) the same with the Cholesky-route distance.  (ii) and (iii) are
    ~11 000 closures each (~8 CPU-hours on the 8-vCPU build container): only (i) is in the committed file."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=100, D=3072)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()
    path = os.path.join(HERE, "g7b_fit_c5_wellposed.npz")
    for key, kwargs, fun in (("sqfa_cholroute", {}, cholesky_fisher_rao),
                             ("sqfa_wolfe", {"line_search_fn": "strong_wolfe"}, None),
                             ("sqfa_wolfe_cholroute", {"line_search_fn": "strong_wolfe"}, cholesky_fisher_rao)):
        extra = {"distance_fun": fun} if fun is not None else {}
        model = sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01, **extra).double()
        model.fit_pca(data_statistics=stats)
        out[f"{key}_init"] = model.filters.detach().numpy().copy()
        _fit_record(out, key, model, stats, **kwargs)
        np.savez_compressed(path, **out)   # keep partial results
    torch.set_default_dtype(torch.float32)


# ---------------------------------------------------------------- G5b: the other distance_fun operators, with gradients
GAUSS_OPS = ("bhattacharyya", "mahalanobis_sq", "mahalanobis", "hellinger", "fisher_rao_same_cov")


def g5b():
    """Values and gradients of the reference's remaining distance operators
    (src/sqfa/distances.py:92-138, 240-432) for self (A is B) and cross batches, float64 and float32:
    loss = sum(W * D) with a random non-symmetric W; gradients wrt means and covariances."""
    rng = np.random.default_rng(5151)
    out = {}
    cases = [(5, 0, 4), (4, 7, 3), (12, 0, 16), (100, 0, 8), (6, 0, 33), (9, 0, 1), (3, 5, 17), (40, 0, 2)]
    out["cases"] = np.array(cases)
    for (nA, nB, K) in cases:
        key = f"A{nA}_B{nB}_K{K}"
        covA, muA = random_spd(rng, nA, K, lo=0.05), 0.8 * rng.standard_normal((nA, K))
        out[f"{key}_covA"], out[f"{key}_muA"] = covA, muA
        if nB:
            covB, muB = random_spd(rng, nB, K, lo=0.05), 0.8 * rng.standard_normal((nB, K))
            out[f"{key}_covB"], out[f"{key}_muB"] = covB, muB
        W = rng.standard_normal((nA, nB or nA))
        out[f"{key}_W"] = W
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            def leaves():
                a = {"means": T(muA, dt).requires_grad_(True), "covariances": T(covA, dt).requires_grad_(True)}
                b = a if not nB else {"means": T(muB, dt).requires_grad_(True), "covariances": T(covB, dt).requires_grad_(True)}
                return a, b
            for name in GAUSS_OPS + ("log_euclidean_sq", "log_euclidean"):
                a, b = leaves()
                fn = getattr(sqfa.distances, name)
                D = fn(a["covariances"], b["covariances"]) if name.startswith("log_") else fn(a, b)
                out[f"{key}_{name}_{tag}"] = D.detach().numpy()
                loss = torch.sum(T(W, dt).reshape(D.shape) * D)
                wrt = [a["covariances"]] + ([b["covariances"]] if nB else [])
                if not name.startswith("log_"):
                    wrt += [a["means"]] + ([b["means"]] if nB else [])
                grads = torch.autograd.grad(loss, wrt)
                names = ["gcovA"] + (["gcovB"] if nB else [])
                if not name.startswith("log_"):
                    names += ["gmuA"] + (["gmuB"] if nB else [])
                for gname, g in zip(names, grads):
                    out[f"{key}_{name}_{gname}_{tag}"] = g.numpy()
    np.savez_compressed(os.path.join(HERE, "g5b_other_operators.npz"), **out)


# ---------------------------------------------------------------- G5c: class statistics at scale
def ragged_points(C=1000, d=5, seed=606):
    """Ragged class sizes (2..41 points), float32-representable coordinates, shuffled; regenerated
    from the seed by the tests (numpy Generator: identical on every platform)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(2, 42, size=C)
    y = np.repeat(np.arange(C), sizes)
    X = (rng.standard_normal((len(y), d)) * rng.uniform(0.5, 2.0, size=(1, d)) + 0.3 * rng.standard_normal((C, d))[y])
    X = X.astype(np.float32).astype(np.float64)
    perm = rng.permutation(len(y))
    return X[perm], y[perm]


def g5c():
    """class_statistics of the reference (src/sqfa/statistics.py:8-124) on a ragged 1000-class set,
    both estimators, float64; plus a float-label call (the reference accepts float labels)."""
    out = {}
    torch.set_default_dtype(torch.float64)
    X, y = ragged_points()
    out["check_X0"] = X[:3].copy()
    for est in ("empirical", "oas"):
        st = sqfa.statistics.class_statistics(T(X, torch.float64), torch.tensor(y), estimator=est)
        for k, v in st.items():
            out[f"{est}_{k}"] = v.numpy()
    st = sqfa.statistics.class_statistics(T(X, torch.float64), torch.tensor(y, dtype=torch.float64), estimator="empirical")
    out["float_labels_means"] = st["means"].numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g5c_class_statistics.npz"), **out)


def g4c():
    """Ensemble estimate of the reference's own sensitivity on the syn K=4 fits: the reference fitted
    from 8 copies of the fit_pca initialisation perturbed by 1e-14 relative noise (raw parameter),
    with its default distance functions, fixed-step and strong-Wolfe LBFGS.  The spread of the
    learned filters over the ensemble is the yardstick for "filters match" on these fits."""
    out = {}
    torch.set_default_dtype(torch.float64)
    g4_data = np.load(os.path.join(HERE, "g4_fit.npz"))
    stats = {"means": T(g4_data["syn_mu"], torch.float64), "covariances": T(g4_data["syn_cov"], torch.float64)}
    rng = np.random.default_rng(4343)
    for model_name in ("smsqfa", "sqfa"):
        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
        for opt, kwargs in (("fixed", {}), ("wolfe", {"line_search_fn": "strong_wolfe"})):
            filters, finals, epochs = [], [], []
            for sample in range(8):
                model = cls(n_dim=50, n_filters=4, feature_noise=1e-2).double()
                model.fit_pca(data_statistics=stats)
                with torch.no_grad():
                    prm = model.parametrizations.filters.original
                    prm.mul_(1.0 + 1e-14 * T(rng.standard_normal(tuple(prm.shape)), torch.float64))
                loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True, **kwargs)
                filters.append(model.filters.detach().numpy())
                finals.append(float(loss[-1]))
                epochs.append(len(loss))
            out[f"syn_{model_name}_K4_{opt}_filters"] = np.stack(filters)
            out[f"syn_{model_name}_K4_{opt}_final_loss"] = np.array(finals)
            out[f"syn_{model_name}_K4_{opt}_epochs"] = np.array(epochs)
            print(model_name, opt, epochs, finals, flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g4c_fit_ensemble.npz"), **out)


def g7c():
    """Two more samples of the reference's sensitivity on the c5 configuration: fixed-step fits from
    the fit_pca initialisation perturbed by 1e-14 relative noise (default distance function)."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=100, D=3072)
    rng = np.random.default_rng(7373)
    path = os.path.join(HERE, "g7c_fit_c5_ensemble.npz")
    filters, losses = [], []
    for sample in range(2):
        model = sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01).double()
        model.fit_pca(data_statistics=stats)
        with torch.no_grad():
            prm = model.parametrizations.filters.original
            prm.mul_(1.0 + 1e-14 * T(rng.standard_normal(tuple(prm.shape)), torch.float64))
        loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
        filters.append(model.filters.detach().numpy())
        losses.append(loss.numpy())
        print("c5 perturbed sample", sample, "epochs", len(loss), "final", float(loss[-1]), flush=True)
        out["sqfa_filters"] = np.stack(filters)
        out["sqfa_final_loss"] = np.array([l[-1] for l in losses])
        out["sqfa_epochs"] = np.array([len(l) for l in losses])
        np.savez_compressed(path, **out)
    torch.set_default_dtype(torch.float32)


def g7d():
    """Three more samples of the same sensitivity experiment with larger perturbations of the initial filters
    (1e-12, 1e-10, 1e-8 relative), with the per-epoch losses: does the reference itself ever end the c5 fit at a
    different stall point?  (Round 2: re-associating two float64 sums of the GPU closure did.)"""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=100, D=3072)
    rng = np.random.default_rng(9191)
    path = os.path.join(HERE, "g7d_fit_c5_ensemble2.npz")
    filters, losses, scales = [], [], []
    for sample, scale in enumerate((1e-12, 1e-10, 1e-8)):
        model = sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01).double()
        model.fit_pca(data_statistics=stats)
        with torch.no_grad():
            prm = model.parametrizations.filters.original
            prm.mul_(1.0 + scale * T(rng.standard_normal(tuple(prm.shape)), torch.float64))
        loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
        filters.append(model.filters.detach().numpy())
        losses.append(loss.numpy())
        scales.append(scale)
        print("c5 perturbed sample", sample, "scale", scale, "epochs", len(loss), "final", float(loss[-1]), flush=True)
        out["sqfa_filters"] = np.stack(filters)
        out["sqfa_final_loss"] = np.array([l[-1] for l in losses])
        out["sqfa_epochs"] = np.array([len(l) for l in losses])
        out["sqfa_scales"] = np.array(scales)
        width = max(len(l) for l in losses)
        out["sqfa_loss"] = np.stack([np.pad(l, (0, width - len(l)), constant_values=np.nan) for l in losses])
        np.savez_compressed(path, **out)
    torch.set_default_dtype(torch.float32)


# ---------------------------------------------------------------- G3b: transform_scatters at BASELINE shapes
def g3b():
    """The reference's transform_scatters / transform (src/sqfa/model.py:172-237, conjugate_matrix
    src/sqfa/linalg.py:19-45) on c3- and c4-shaped inputs regenerated from seeds by the tests
    (c2_statistics(C, D)): values and the gradient of a weighted sum wrt the raw filter parameter."""
    out = {}
    for (C, D, K) in ((3, 784, 16), (2, 2048, 32), (4, 132, 8)):
        key = f"C{C}_D{D}_K{K}"
        stats = c2_statistics(C=C, D=D)
        rng = np.random.default_rng(1000 + D + K)
        raw = rng.standard_normal((K, D))
        W = rng.standard_normal((C, K, K))
        out[f"{key}_raw"], out[f"{key}_W"] = raw, W
        out[f"{key}_check"] = stats["covariances"][0, :3, :3].numpy()
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            torch.set_default_dtype(dt)
            model = sqfa.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.0)
            if dt == torch.float64:
                model = model.double()
            with torch.no_grad():
                model.parametrizations.filters.original.copy_(T(raw, dt))
            S = model.transform_scatters(stats["covariances"].to(dt))
            Z = model.transform(stats["means"].to(dt))
            loss = torch.sum(T(W, dt) * S)
            loss.backward()
            out[f"{key}_S_{tag}"] = S.detach().numpy()
            out[f"{key}_Z_{tag}"] = Z.detach().numpy()
            out[f"{key}_grad_{tag}"] = model.parametrizations.filters.original.grad.numpy()
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g3b_transform_scatters.npz"), **out)


def g6b():
    """BASELINE config 1 shape (the README example's: 10 classes, n_dim=784, n_filters=4, feature_noise=0.01,
    SQFA) on the synthetic generator, float64, fit_pca init: the reference's fit, the same with the
    Cholesky-route distance_fun, and 4 fits from initialisations perturbed by 1e-14 (its own sensitivity)."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=10, D=784)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()
    rng = np.random.default_rng(6161)
    model = sqfa.model.SQFA(n_dim=784, n_filters=4, feature_noise=0.01).double()
    model.fit_pca(data_statistics=stats)
    out["sqfa_init"] = model.filters.detach().numpy().copy()
    _fit_record(out, "sqfa", model, stats)
    model = sqfa.model.SQFA(n_dim=784, n_filters=4, feature_noise=0.01, distance_fun=cholesky_fisher_rao).double()
    model.fit_pca(data_statistics=stats)
    _fit_record(out, "sqfa_cholroute", model, stats)
    ens = []
    for sample in range(4):
        model = sqfa.model.SQFA(n_dim=784, n_filters=4, feature_noise=0.01).double()
        model.fit_pca(data_statistics=stats)
        with torch.no_grad():
            prm = model.parametrizations.filters.original
            prm.mul_(1.0 + 1e-14 * T(rng.standard_normal(tuple(prm.shape)), torch.float64))
        loss, _ = model.fit(data_statistics=stats, max_epochs=300, show_progress=False, return_loss=True)
        ens.append(model.filters.detach().numpy())
        print("c1 perturbed", sample, len(loss), float(loss[-1]), flush=True)
    out["sqfa_ensemble_filters"] = np.stack(ens)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g6b_fit_c1.npz"), **out)


# ---------------------------------------------------------------- round 3: well-posed points of the ill-posed fits
def _early_fits(out, tag, make_model, stats, epochs_list, alt, **fit_kwargs):
    """The reference's filters after `max_epochs` = E epochs (src/sqfa/_optim.py:105-134), E small enough
    that the fixed-step trajectory has not yet amplified rounding: the point at which "filters to 1e-5"
    IS a property of the algorithm.  Also the same fit with the Cholesky-route distance_fun (the
    reference's own rounding-level yardstick at that epoch)."""
    for E in epochs_list:
        for route, dist in (("", None), ("_cholroute", alt)):
            model = make_model(dist)
            model.fit_pca(data_statistics=stats)
            if f"{tag}_init" not in out:
                out[f"{tag}_init"] = model.filters.detach().numpy().copy()
            loss, _ = model.fit(data_statistics=stats, max_epochs=E, show_progress=False, return_loss=True, **fit_kwargs)
            out[f"{tag}_e{E}{route}_loss"] = loss.numpy()
            out[f"{tag}_e{E}{route}_filters"] = model.filters.detach().numpy()
            print(tag, "epochs", E, route or "reference", "losses", loss.numpy(), flush=True)


def g7e():
    """BASELINE config 5 shape (C=100, n_dim=3072, n_filters=16, SQFA, float64): the reference's learned
    filters after 3 and 5 epochs."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=100, D=3072)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()
    path = os.path.join(HERE, "g7e_fit_c5_early.npz")

    def make(dist):
        return sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01, distance_fun=dist).double()

    _early_fits(out, "sqfa", make, stats, (3, 5), cholesky_fisher_rao)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(path, **out)


def g7f():
    """BASELINE config 5 shape with line_search_fn="strong_wolfe" (LBFGS kwargs are forwarded, src/sqfa/_optim.py:78-82):
    the reference's filters after 1 and 2 epochs (a full strong-Wolfe fit of this configuration is ~11 000 closures,
    ~8 CPU-hours per run: golden G7b never got it; two epochs are ~500 closures)."""
    out = {}
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(3)
    stats = c2_statistics(C=100, D=3072)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()

    def make(dist):
        return sqfa.model.SQFA(n_dim=3072, n_filters=16, feature_noise=0.01, distance_fun=dist).double()

    path = os.path.join(HERE, "g7f_fit_c5_wolfe_early.npz")
    for E in (1, 2):
        _early_fits(out, "sqfa", make, stats, (E,), cholesky_fisher_rao, line_search_fn="strong_wolfe")
        np.savez_compressed(path, **out)
    torch.set_default_dtype(torch.float32)


def g6c():
    """BASELINE config 1 shape (C=10, n_dim=784, n_filters=4, SQFA, float64): the reference's learned filters
    after 5 and 15 epochs (its chaotic episode starts around epoch 21, golden G6b)."""
    out = {}
    torch.set_default_dtype(torch.float64)
    stats = c2_statistics(C=10, D=784)
    out["check_cov00"] = stats["covariances"][0, :4, :4].numpy()

    def make(dist):
        return sqfa.model.SQFA(n_dim=784, n_filters=4, feature_noise=0.01, distance_fun=dist).double()

    _early_fits(out, "sqfa", make, stats, (5, 15), cholesky_fisher_rao)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g6c_fit_c1_early.npz"), **out)


def g4d():
    """rotated_classes_dataset, K=4, three epochs (the 'flat' case of G4): the reference's own spread at that
    point -- Cholesky-route distance_fun and 8 fits from 1e-14-perturbed initial filters."""
    out = {}
    torch.set_default_dtype(torch.float64)
    rot = rotated_classes_dataset().double()
    stats = {"means": torch.zeros(5, 8), "covariances": rot}
    rng = np.random.default_rng(4444)
    for model_name in ("smsqfa", "sqfa"):
        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
        alt = cholesky_fisher_rao if model_name == "sqfa" else cholesky_affine_invariant
        model = cls(n_dim=8, n_filters=4, feature_noise=1e-2, distance_fun=alt).double()
        model.fit_pca(data_statistics=stats)
        loss, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
        out[f"rot_{model_name}_K4_e3_cholroute_loss"] = loss.numpy()
        out[f"rot_{model_name}_K4_e3_cholroute_filters"] = model.filters.detach().numpy()
        filters, losses = [], []
        for sample in range(8):
            model = cls(n_dim=8, n_filters=4, feature_noise=1e-2).double()
            model.fit_pca(data_statistics=stats)
            with torch.no_grad():
                prm = model.parametrizations.filters.original
                prm.mul_(1.0 + 1e-14 * T(rng.standard_normal(tuple(prm.shape)), torch.float64))
            loss, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
            filters.append(model.filters.detach().numpy())
            losses.append(loss.numpy())
        out[f"rot_{model_name}_K4_e3_ensemble_filters"] = np.stack(filters)
        out[f"rot_{model_name}_K4_e3_ensemble_loss"] = np.stack(losses)
        print(model_name, "ensemble losses", np.stack(losses)[:, -1], flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g4d_fit_rot_spread.npz"), **out)


def g3o():
    """G3 for constraint="orthogonal" (src/sqfa/model.py:416-431 -> torch.nn.utils.parametrizations.orthogonal),
    reproducible this time: the parametrization's random `base` buffer is stored, and the raw parameter is of
    the form the parametrization itself produces and trains (diagonal -1 as returned by its right_inverse, the
    reflector entries random) -- G3's plain-random raw parameters had a diagonal that truncates to 0 in the
    Householder map, which made the reference return NaN gradients."""
    rng = np.random.default_rng(3030)
    out = {}
    rot = rotated_classes_dataset().double().numpy()
    out["rotated_cov"] = rot
    mu_rot = 0.2 * rng.standard_normal((rot.shape[0], rot.shape[1]))
    out["rotated_mu"] = mu_rot
    for K in (1, 2, 3, 4, 8):
        raw = 0.4 * rng.standard_normal((K, 8))
        raw[np.arange(K), np.arange(K)] = -1.0
        out[f"raw_filters_K{K}"] = raw
    for model_name in ("smsqfa", "sqfa"):
        for noise in (1e-3, 1e-2):
            for K in (1, 2, 3, 4, 8):
                for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
                    torch.set_default_dtype(dt)
                    torch.manual_seed(1000 * K + int(noise * 1e4))
                    cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
                    model = cls(n_dim=8, n_filters=K, feature_noise=noise, constraint="orthogonal")
                    if dt == torch.float64:
                        model = model.double()
                    key = f"{model_name}_orthogonal_n{noise:g}_K{K}"
                    par = model.parametrizations.filters[0]
                    base_key = f"{key}_base"
                    if base_key in out:       # float32 run: the float64 run's base, rounded
                        par.base = T(out[base_key], dt)
                    else:
                        out[base_key] = par.base.detach().double().numpy().copy()
                        out[f"{key}_map"] = np.array(par.orthogonal_map.value if hasattr(par.orthogonal_map, "value") else -1)
                    with torch.no_grad():
                        prm = model.parametrizations.filters.original
                        prm.copy_(T(out[f"raw_filters_K{K}"], dt))
                    stats = {"means": T(mu_rot, dt), "covariances": T(rot, dt)}
                    inp = stats if model_name == "sqfa" else T(rot, dt)
                    try:
                        D_ = model.get_class_distances(inp, regularized=True)
                    except Exception as err:
                        print("  skipped", key, tag, type(err).__name__)
                        continue
                    loss = tril_loss(D_)
                    model.zero_grad()
                    loss.backward()
                    out[f"{key}_filters_{tag}"] = model.filters.detach().numpy()
                    out[f"{key}_D_{tag}"] = D_.detach().numpy()
                    out[f"{key}_loss_{tag}"] = loss.detach().numpy()
                    out[f"{key}_grad_{tag}"] = prm.grad.detach().numpy()
                    print(key, tag, "loss", float(loss), "finite grad", bool(torch.isfinite(prm.grad).all()), flush=True)
    # a short orthogonal fit (three epochs, float64): trajectory and learned filters
    torch.set_default_dtype(torch.float64)
    for model_name in ("smsqfa", "sqfa"):
        cls = sqfa.model.SQFA if model_name == "sqfa" else sqfa.model.SecondMomentsSQFA
        torch.manual_seed(77)
        model = cls(n_dim=8, n_filters=3, feature_noise=1e-2, constraint="orthogonal").double()
        key = f"{model_name}_orthogonal_fit_K3"
        out[f"{key}_base"] = model.parametrizations.filters[0].base.detach().numpy().copy()
        out[f"{key}_raw"] = model.parametrizations.filters.original.detach().numpy().copy()
        out[f"{key}_init"] = model.filters.detach().numpy().copy()
        stats = {"means": T(mu_rot, torch.float64), "covariances": T(rot, torch.float64)}
        loss, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
        out[f"{key}_loss"] = loss.numpy()
        out[f"{key}_filters"] = model.filters.detach().numpy()
        print(key, loss.numpy(), flush=True)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "g3o_closure_orthogonal.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g1x", "g2", "g3", "g4", "g5"]
    for name in which:
        print("generating", name, flush=True)
        globals()[name]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
