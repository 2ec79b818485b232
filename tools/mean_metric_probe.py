"""Developer aid (numpy emulation, CPU): would class factors taken in the metric of the MEAN class save pair sweeps?
  K1's sweep count depends on how far F_i^T Sigma_j^-1 F_i starts from diagonal, F_i any factor of
Sigma_i.  K0b (shipped) makes F_i^T F_i diagonal (F_i = Q_i Lambda_i^1/2) -- the best class-level choice when the classes
scatter around a multiple of I.  For classes that share a dominant covariance, Sigma_c = Sbar^1/2 (I + E_c) Sbar^1/2, the
candidate is F_i with F_i^T Sbar^-1 F_i diagonal: F_i = Lbar U_i Sig_i from the SVD Lbar^-1 L_i = U_i Sig_i V_i^T.

Emulates the kernel's lane-group tournament (tools/jacobi_emulation.py) in float32 with the shipped stop rule (a sweep
whose cos^2 all stay below 3e-6 is the last); reports sweeps per pair and per wave of 16 (m <= 17) / 8 pairs.
    python tools/mean_metric_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jacobi_emulation as je

EARLY_COS2 = 3e-6


def sweeps_for(F, Linv, pairs, G, CPL, per_wave):
    ii, jj = pairs
    m = F.shape[-1]
    n = G * CPL
    X = np.zeros((len(ii), m, n))
    X[:, :, :m] = Linv[jj] @ F[ii]
    # the emulation stops on max |cos| < early_thr: patch the threshold to the kernel's cos^2 rule
    lam, sw, _ = jacobi_pairs_thr(X, G, CPL)
    return sw.mean(), sw.reshape(-1, per_wave).max(axis=1).mean()


def jacobi_pairs_thr(X, G, CPL):
    src = je.jacobi_pairs
    # re-run the emulation with its early threshold replaced (sqrt of the kernel's cos^2 bound)
    import types
    code = src.__code__
    consts = tuple(c for c in code.co_consts)
    fn = types.FunctionType(code, dict(src.__globals__, np=_NpProxy(np)), src.__name__, src.__defaults__, src.__closure__)
    return fn(X, G, CPL, np.float32, 14, True, False)


class _NpProxy:
    """numpy with finfo(float32).eps adjusted so that early_thr = sqrt(eps) * 0.5 equals sqrt(EARLY_COS2), tol unchanged
    in effect (tol only gates rotations far below the early threshold)."""

    def __init__(self, np_):
        self._np = np_

    def __getattr__(self, k):
        return getattr(self._np, k)

    def finfo(self, dt):
        real = self._np.finfo(dt)

        class F:
            eps = (2 * np.sqrt(EARLY_COS2)) ** 2
        return F if dt == np.float32 else real


def factors(S):
    L = np.linalg.cholesky(S)
    lam, Q = np.linalg.eigh(S)
    eigf = Q * np.sqrt(lam)[:, None, :]
    Sbar = S.mean(axis=0)
    Lbar = np.linalg.cholesky(Sbar)
    W = np.linalg.solve(Lbar, L)                      # Lbar^-1 L_i
    U, sig, Vt = np.linalg.svd(W)
    meanf = L @ Vt.transpose(0, 2, 1)                 # L_i V_i = Lbar U_i Sig_i
    return L, eigf, meanf


def shared_structure(rng, C, m, cond, spread):
    q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ev = np.exp(np.linspace(0.0, np.log(cond), m))
    root = (q * np.sqrt(ev)) @ q.T
    out = np.empty((C, m, m))
    for c in range(C):
        E = rng.standard_normal((m, m)) * spread / np.sqrt(m)
        E = 0.5 * (E + E.T)
        M = np.eye(m) + E
        M = M @ M.T                                   # I + 2E + E^2: SPD whatever the draw
        out[c] = root @ M @ root
    return out


def report(name, S, G, CPL, per_wave, npairs=640):
    C = S.shape[0]
    ii, jj = np.tril_indices(C, -1)
    sel = np.random.default_rng(1).permutation(len(ii))[:npairs]
    pairs = (ii[sel], jj[sel])
    L, eigf, meanf = factors(S)
    Linv = np.linalg.inv(L)
    res = [sweeps_for(F, Linv, pairs, G, CPL, per_wave) for F in (L, eigf, meanf)]
    print(f"{name:46s} Cholesky {res[0][0]:.2f} / {res[0][1]:.2f}   eigen-factor (K0b) {res[1][0]:.2f} / {res[1][1]:.2f}   "
          f"mean-metric {res[2][0]:.2f} / {res[2][1]:.2f}   gain over K0b {res[1][0] - res[2][0]:+.2f} / {res[1][1] - res[2][1]:+.2f}", flush=True)


if __name__ == "__main__":
    print("sweeps per pair / per wave; float32 emulation, stop rule cos^2 < 3e-6")
    rng = np.random.default_rng(0)
    report("BASELINE generator c3 (m=16)", je.baseline_like(60, 784, 16), 4, 4, 16)
    report("BASELINE generator c3-SQFA (m=17)", je.baseline_like(60, 784, 16, sqfa=True), 4, 5, 16)
    report("BASELINE generator c4 (m=32)", je.baseline_like(48, 2048, 32), 8, 4, 8)
    for cond in (1e2, 1e3, 1e4):
        for spread in (0.2, 0.6):
            report(f"shared Sbar cond {cond:g}, spread {spread} (m=16)", shared_structure(rng, 60, 16, cond, spread), 4, 4, 16)
    for cond in (1e2, 1e4):
        report(f"shared Sbar cond {cond:g}, spread 0.4 (m=17)", shared_structure(rng, 60, 17, cond, 0.4), 4, 5, 16)
        report(f"shared Sbar cond {cond:g}, spread 0.4 (m=32)", shared_structure(rng, 48, 32, cond, 0.4), 8, 4, 8)
