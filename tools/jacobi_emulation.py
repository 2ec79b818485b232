"""Developer aid (NOT on the product path, NOT an oracle): numpy emulation of the
lane-group one-sided Jacobi schedule used by the HIP pair kernel, to study
sweep counts and float32 accuracy on the CPU before touching the GPU.

Layout emulated: G lanes per pair, CPL column slots per lane, columns of
X = L_j^{-1} L_i are rotated (Hestenes) with unnormalised "fast" rotations;
per-sweep renormalisation; XOR tournament across lanes (lane^s, slot^t).

This is the ordering of rounds 1-3.  Round 4's kernels visit the same column pairs once per sweep in a different order (slot
exchanges + local rotations, pair_kernel.hpp: exchange_slots); sweep counts on the GPU moved by -0.05 ... +0.08 per wave round.
"""
import numpy as np


def make_spd(rng, n, m, cond=1e3):
    out = np.empty((n, m, m))
    for k in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((m, m)))
        ev = np.exp(rng.uniform(np.log(1.0 / cond), 0.0, m)) * rng.uniform(0.5, 4.0)
        out[k] = (q * ev) @ q.T
    return out


def jacobi_pairs(X, G, CPL, dtype=np.float32, max_sweeps=12, early=True, verbose=False):
    """X: (B, R, n) with n == G*CPL columns (zero columns allowed). Returns lam (B,n), sweeps (B,)"""
    X = X.astype(dtype).copy()
    B, R, n = X.shape
    assert n == G * CPL
    eps = np.finfo(dtype).eps
    tol = dtype(eps * np.sqrt(R))
    early_thr = dtype(np.sqrt(eps) * 0.5)
    col = lambda g, c: g * CPL + c
    done = np.zeros(B, bool)
    sweeps = np.zeros(B, int)
    f = np.ones((B, n), dtype)

    def rotate(p, q, nrm, maxrel):
        # p, q: arrays of column indices (disjoint pairs), processed simultaneously
        xp = X[:, :, p]
        xq = X[:, :, q]
        gam = np.einsum("brk,brk->bk", xp, xq).astype(dtype)
        a = nrm[:, p]
        b = nrm[:, q]
        ab = a * b
        rel2 = np.where(ab > 0, gam * gam / np.where(ab > 0, ab, 1), 0).astype(dtype)
        act = (rel2 > tol * tol) & (~done[:, None])
        maxrel[:] = np.maximum(maxrel, np.sqrt(np.where(done[:, None], 0, rel2)).max(axis=1))
        g2 = np.where(act, gam, 1).astype(dtype)
        zeta = ((b - a) / (dtype(2) * g2)).astype(dtype)
        t = (np.sign(zeta) + (zeta == 0)) / (np.abs(zeta) + np.sqrt(dtype(1) + zeta * zeta))
        t = np.where(act, t, 0).astype(dtype)
        c = (dtype(1) / np.sqrt(dtype(1) + t * t)).astype(dtype)
        sn = (t * c).astype(dtype)
        X[:, :, p] = c[:, None, :] * xp - sn[:, None, :] * xq
        X[:, :, q] = sn[:, None, :] * xp + c[:, None, :] * xq
        nrm[:, p] = a - t * gam
        nrm[:, q] = b + t * gam
        return act.any(axis=1)

    for sweep in range(max_sweeps):
        # renormalise + recompute norms
        X *= (dtype(1) / np.sqrt(f))[:, None, :]
        f[:] = 1
        nrm = np.einsum("brk,brk->bk", X, X).astype(dtype)
        maxrel = np.zeros(B, dtype)
        rotated = np.zeros(B, bool)
        # intra-lane pairs
        for c1 in range(CPL):
            for c2 in range(c1 + 1, CPL):
                p = np.array([col(g, c1) for g in range(G)])
                q = np.array([col(g, c2) for g in range(G)])
                rotated |= rotate(p, q, nrm, maxrel)
        # cross-lane
        for s in range(1, G):
            tmax = 1
            while tmax < CPL:
                tmax *= 2
            for t in range(tmax):
                p, q = [], []
                for g in range(G):
                    if g < (g ^ s):
                        for c in range(CPL):
                            if (c ^ t) < CPL:
                                p.append(col(g, c))
                                q.append(col(g ^ s, c ^ t))
                if p:
                    rotated |= rotate(np.array(p), np.array(q), nrm, maxrel)
        sweeps[~done] += 1
        if early:
            done |= maxrel < early_thr
        else:
            done |= ~rotated
        if verbose:
            print("sweep", sweep, "active", (~done).sum(), "maxrel", maxrel.max())
        if done.all():
            break
    X *= (dtype(1) / np.sqrt(f))[:, None, :]
    lam = np.einsum("brk,brk->bk", X, X).astype(dtype)
    return lam, sweeps, X


if __name__ == "__main__":
    import sys
    rng = np.random.default_rng(0)
    for (m, G, CPL) in [(16, 4, 4), (17, 4, 5), (8, 2, 4), (32, 8, 4), (33, 8, 5), (4, 1, 4)]:
        n = G * CPL
        S = make_spd(rng, 40, m, cond=float(sys.argv[1]) if len(sys.argv) > 1 else 1e3)
        L = np.linalg.cholesky(S)
        Linv = np.linalg.inv(L)
        ii, jj = np.tril_indices(40, -1)
        ii, jj = ii[:400], jj[:400]
        Xd = Linv[jj] @ L[ii]
        lam_ref = np.linalg.eigvalsh(Xd @ Xd.transpose(0, 2, 1))
        d2_ref = (np.log(lam_ref) ** 2).sum(-1)
        Xp = np.zeros((len(ii), m, n))
        Xp[:, :, :m] = Xd
        for early in (False, True):
            lam, sweeps, _ = jacobi_pairs(Xp, G, CPL, np.float32, early=early)
            lam = np.sort(lam, axis=1)[:, n - m:]
            d2 = (np.log(lam.astype(np.float64)) ** 2).sum(-1)
            rel = np.abs(np.sqrt(d2) - np.sqrt(d2_ref)) / np.sqrt(d2_ref)
            print(f"m={m} G={G} CPL={CPL} early={early}: sweeps mean {sweeps.mean():.2f} max {sweeps.max()}  "
                  f"rel d err max {rel.max():.2e} mean {rel.mean():.2e}; "
                  f"lam rel err max {np.abs(lam/lam_ref-1).max():.2e}")


def baseline_like(C, D, K, seed=1234, noise=0.01, sqfa=False):
    """SURVEY 8(d) synthetic generator, numpy restatement (float64)."""
    rng = np.random.default_rng(seed)
    R = min(D, 128)
    F = np.random.default_rng(7).standard_normal((K, D))
    F /= np.linalg.norm(F, axis=1, keepdims=True)
    S = np.empty((C, K + int(sqfa), K + int(sqfa)))
    for c in range(C):
        A = rng.standard_normal((D, R)) / np.sqrt(R)
        mu = 0.1 * rng.standard_normal(D)
        FA = F @ A
        cov = FA @ FA.T + 0.05 * (F @ F.T) + noise * np.eye(K)
        fm = F @ mu
        if sqfa:
            S[c, :K, :K] = cov + np.outer(fm, fm)
            S[c, :K, K] = fm
            S[c, K, :K] = fm
            S[c, K, K] = 1
        else:
            S[c] = cov + np.outer(fm, fm)
    return S
