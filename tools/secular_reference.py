"""Developer aid (numpy, CPU): the complete arithmetic of the planned "bordered" pair kernel for sizes m = K + 1, for ANY
SPD input (not only the Calvo-Oller embedding), checked against float64 LAPACK on the full pencil -- values AND gradients.

  B = [[A, b], [b^T, beta]]  ->  Schur form:  B = T diag(Sigma, beta) T^T,  Sigma = A - b b^T / beta,  T = [[I, mu], [0, 1]],  mu = b / beta
  pencil (B_i, B_j) after the congruence  P_j = diag(L_j^-1, beta_j^-1/2) T_j^-1  (L_j L_j^T = Sigma_j):
      M = P_j B_i P_j^T = diag(N, 0) + beta_i v v^T,   N = L_j^-1 Sigma_i L_j^-T,   v = (L_j^-1 (mu_i - mu_j); beta_j^-1/2)
  N = V diag(nu) V^T (the K x K Jacobi),  c = V^T v[:K]:   M ~ diag(nu, 0) + rho (c; v_b)(c; v_b)^T,  rho = beta_i
  eigenvalues = roots of  1 + rho sum_k z_k^2 / (p_k - lam) = 0  over the poles p = (nu, 0), z = (c, v_b); the root owned by
  pole p_k lies in (p_k, next larger pole); eigenvector ~ z_k / (p_k - lam); components with a negligible weight are
  deflated (root = pole, eigenvector = unit vector).
  generalized eigenvectors:  u = P_j^T diag(V, 1) w;   d^2 = scale * sum log^2 lam;
  dL/dB_i = sum_r g_r u_r u_r^T,   dL/dB_j = - sum_r g_r lam_r u_r u_r^T,   g_r = w * dD/dd2 * scale * 2 log(lam_r) / lam_r

The secular stage runs in float32 (numpy float32 arithmetic, every pole difference formed from the root's offset to its
origin pole, a safeguarded fixed-weight rational iteration: evaluation counts are printed); everything is compared
with float64 eigh of L_Bj^-1 B_i L_Bj^-T and the closed-form gradient.      python tools/secular_reference.py"""
import numpy as np

rng = np.random.default_rng(1)
F = np.float32


def spd(m, cond=1e3):
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ev = np.exp(rng.uniform(np.log(1.0 / np.sqrt(cond)), np.log(np.sqrt(cond)), m))
    S = (Q * ev) @ Q.T
    return 0.5 * (S + S.T)


def reference(Bi, Bj, scale=0.5, eps=1e-6, w=1.0):
    L = np.linalg.cholesky(Bj)
    Li = np.linalg.inv(L)
    lam, Vw = np.linalg.eigh(Li @ Bi @ Li.T)
    U = Li.T @ Vw                                  # B_j-normalised generalized eigenvectors
    d2 = scale * np.sum(np.log(lam) ** 2)
    D = np.sqrt(d2 + eps)
    g = w * 0.5 / D * scale * 2 * np.log(lam) / lam
    return D, (U * g) @ U.T, -(U * (g * lam)) @ U.T


def _minus_branch(c, a, b):
    """the root (a - sqrt(a^2 - 4 b c)) / (2 c) of  c T^2 - a T + b = 0, evaluated without cancellation: for a model
    c + a_o / (-T) + s / (G - T) with positive weights it is the one between the two model poles"""
    disc = np.sqrt(max(a * a - F(4) * b * c, F(0)))
    if c == 0:
        return b / a
    return (a - disc) / (F(2) * c) if a <= 0 else F(2) * b / (a + disc)


def secular_roots_f32(p, z2, rho):
    """roots of 1 + rho sum z2_k / (p_k - lam), float32; returns (origin index, offset) per owning pole, iterations.

    "Fixed weight" rational iteration (Li 1994; the family LAPACK's dlaed4 draws from), started at the middle of the root's
    interval: the pole nearer to the root (the ORIGIN; the offset t from it is the unknown) keeps its exact weight, the
    rest of the function -- summed WITHOUT the origin's term, so that a root hugging its pole does not cancel -- is
    modelled as c + s / (p_q - lam) through its value and slope, q = the pole at the other end of the interval.  The
    quadratic is solved for the NEW OFFSET itself, not for an increment (t + eta loses the distance to the pole when the
    root hugs it).  The largest root (no pole above) models the rest as a straight line.  Safeguards: Newton, then
    bisection, whenever a step leaves the bracket.  Stop when |f| is within its own rounding noise.
    Measured (this file's cases and tools/secular_probe.py's): 3-4 evaluations per root on average, <= 10 at worst; the
    plain safeguarded Newton this replaces needed 10-30 when the weights are small."""
    n = len(p)
    p, a = p.astype(F), (z2 * rho).astype(F)
    total = a.sum(dtype=F)
    origin, tau, iters = np.zeros(n, int), np.zeros(n, F), []
    tiny = F(1e-12) * (np.abs(p).max() + total)
    live = a > tiny
    eps = F(6e-8)
    for k in range(n):
        if not live[k]:                                # deflated: the root sits on the pole
            origin[k], tau[k] = k, F(0)
            iters.append(0)
            continue
        above = p[(p > p[k]) & live]
        ip = int(np.where(p == above.min())[0][0]) if above.size else None
        gap = F(p[ip] - p[k]) if ip is not None else total
        o, t, lo, hi = k, F(0.5) * gap, F(0), gap
        first = True
        it = 0
        for it in range(1, 40):
            delta = (p - p[o]).astype(F) - t           # p_j - lam with lam = p_o + t
            r = np.where(live, a / delta, F(0))
            w = F(1) + r.sum(dtype=F)
            if first and ip is not None and w < 0:     # root in the upper half: measure it from the upper pole
                o, t, lo, hi = ip, t - gap, -gap, F(0)
                delta = (p - p[o]).astype(F) - t
            first = False
            if abs(w) <= F(4) * eps * (F(1) + np.abs(r).sum(dtype=F)):
                break
            if w > 0:
                hi = min(hi, t)
            else:
                lo = max(lo, t)
            rest = r.copy()
            rest[o] = 0
            R0, Rp = F(1) + rest.sum(dtype=F), (rest / delta).sum(dtype=F)
            ao, Do = a[o], delta[o]
            if ip is not None:
                q = ip if o == k else k
                Dq, Gq = delta[q], F(p[q] - p[o])
                s, c = Dq * Dq * Rp, R0 - Dq * Rp
                tn = _minus_branch(c, c * Gq + ao + s, ao * Gq)
            else:                                       # A + Rp T - ao / T = 0 with the line's intercept A at T = 0
                A = R0 - Rp * t
                disc = np.sqrt(A * A + F(4) * Rp * ao)
                tn = F(2) * ao / (A + disc) if A > 0 else (disc - A) / (F(2) * Rp)
            if not np.isfinite(tn) or not (lo < tn < hi):
                tn = t - w / (Rp + ao / (Do * Do))
                if not (lo < tn < hi):
                    tn = F(0.5) * (lo + hi)
            if tn == t:
                break
            t = tn
        origin[k], tau[k] = o, t
        iters.append(it)
    return origin, tau, iters, live


def bordered(Bi, Bj, scale=0.5, eps=1e-6, w=1.0):
    K = Bi.shape[0] - 1
    def schur(B):
        A, b, beta = B[:K, :K], B[:K, K], B[K, K]
        return A - np.outer(b, b) / beta, b / beta, beta
    Si, mui, bi = schur(Bi)
    Sj, muj, bj = schur(Bj)
    L = np.linalg.cholesky(Sj)
    Li = np.linalg.inv(L)
    N = Li @ Si @ Li.T
    nu, V = np.linalg.eigh(N)                       # stands for the K-column one-sided Jacobi
    zK = V.T @ (Li @ (mui - muj))
    p = np.concatenate([nu, [0.0]])
    z = np.concatenate([zK, [1.0 / np.sqrt(bj)]])
    origin, tau, iters, live = secular_roots_f32(p, z * z, bi)
    p32, z32 = p.astype(F), z.astype(F)
    lam = (p32[origin] + tau).astype(np.float64)
    W = np.zeros((K + 1, K + 1))
    for k in range(K + 1):
        if not live[k]:
            W[k, k] = 1.0
            continue
        delta = (p32 - p32[origin[k]]).astype(F) - tau[k]
        wv = np.where(live, z32 / delta, F(0)).astype(F)
        W[:, k] = (wv / np.sqrt((wv * wv).sum(dtype=F))).astype(np.float64)
    # generalized eigenvectors u = P_j^T diag(V, 1) w,  P_j = diag(L^-1, bj^-1/2) T_j^-1
    q = Li.T @ (V @ W[:K])                           # (K, K+1)
    last = W[K] / np.sqrt(bj) - muj @ q
    U = np.vstack([q, last])
    d2 = scale * np.sum(np.log(lam) ** 2)
    D = np.sqrt(d2 + eps)
    g = w * 0.5 / D * scale * 2 * np.log(lam) / lam
    return D, (U * g) @ U.T, -(U * (g * lam)) @ U.T, iters


def embed(K, mean_scale):
    A = rng.standard_normal((K, 3 * K)); S = A @ A.T / (3 * K) + 0.05 * np.eye(K)
    mu = mean_scale * rng.standard_normal(K)
    return np.block([[S + np.outer(mu, mu), mu[:, None]], [mu[None, :], np.ones((1, 1))]])


for K in (4, 16, 32):
    for name, make in (("general SPD, cond 1e3", lambda: spd(K + 1)), ("embedding, means 0.3", lambda: embed(K, 0.3)),
                       ("embedding, means 1e-4", lambda: embed(K, 1e-4)), ("embedding, equal means (deflated)", None)):
        worst, its = np.zeros(3), []
        for _ in range(40 if K == 32 else 120):
            if make is None:
                Bi, Bj = embed(K, 0.3), None
                mu = Bi[:K, K].copy()
                A = rng.standard_normal((K, 3 * K)); S = A @ A.T / (3 * K) + 0.05 * np.eye(K)
                Bj = np.block([[S + np.outer(mu, mu), mu[:, None]], [mu[None, :], np.ones((1, 1))]])
            else:
                Bi, Bj = make(), make()
            D0, GA0, GB0 = reference(Bi, Bj)
            D1, GA1, GB1, it = bordered(Bi, Bj)
            its += it
            worst = np.maximum(worst, [abs(D1 - D0) / D0, np.linalg.norm(GA1 - GA0) / np.linalg.norm(GA0),
                                       np.linalg.norm(GB1 - GB0) / np.linalg.norm(GB0)])
        print(f"K={K:2d} {name:36s}: distance rel err {worst[0]:.1e}, dL/dB_i rel err {worst[1]:.1e}, dL/dB_j rel err {worst[2]:.1e}; "
              f"secular iterations mean {np.mean(its):.1f} max {max(its)}")
