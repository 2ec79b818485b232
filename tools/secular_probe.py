"""Developer aid (numpy, CPU): can SQFA's bordered (K+1) x (K+1) pencil be solved as the K x K problem plus a secular
equation in FLOAT32 with the relative accuracy log(lambda) needs?

For the Calvo-Oller embedding E = [[S + mu mu^T, mu], [mu^T, 1]] the whitened matrix of a pair is
    M = [[N + d d^T, d], [d^T, 1]],   N = L_j^-1 S_i L_j^-T,  d = L_j^-1 (mu_i - mu_j).
With N = V diag(nu) V^T and c = V^T d:  M ~ [[diag(nu) + c c^T, c], [c^T, 1]] = diag(nu, 0) + v v^T with v = (c; 1), whose
eigenvalues are the roots of  f(lam) = 1 + sum_k c_k^2 / (nu_k - lam) + 1 / (0 - lam)   (rank-one update of diag(nu, 0)),
interlacing the poles {0, nu_1 .. nu_K}; eigenvectors w ~ (D - lam)^-1 v.
This script solves the secular equation in float32 (bisection + Newton on the interval between two poles, in the
shifted variable measured from the nearer pole) and compares  d^2 = 1/2 sum log^2 lam  and the spectral function
G = sum_k g(lam_k) w_k w_k^T  (the gradient's shape) with float64 LAPACK on M.   python tools/secular_probe.py"""
import numpy as np

rng = np.random.default_rng(0)


def pencil(K, mean_scale):
    A = rng.standard_normal((K, 3 * K)); Si = A @ A.T / (3 * K) + 0.05 * np.eye(K)
    B = rng.standard_normal((K, 3 * K)); Sj = B @ B.T / (3 * K) + 0.05 * np.eye(K)
    mui, muj = mean_scale * rng.standard_normal(K), mean_scale * rng.standard_normal(K)
    L = np.linalg.cholesky(Sj)
    Li = np.linalg.inv(L)
    N = Li @ Si @ Li.T
    d = Li @ (mui - muj)
    return N, d


def secular_f32(nu, c):
    """roots of 1 + sum c_k^2/(p_k - lam) for poles p = (0, nu_1..nu_K) sorted ascending, weights (1, c^2): float32."""
    f = np.float32
    poles = np.concatenate([[0.0], nu]).astype(f)
    w2 = np.concatenate([[1.0], c * c]).astype(f)
    order = np.argsort(poles)
    poles, w2 = poles[order], w2[order]
    n = len(poles)
    roots_shift = np.zeros(n, dtype=f)   # root i = poles[base_i] + shift_i
    base = np.zeros(n, dtype=int)
    for i in range(n):
        lo_p = poles[i]
        hi_p = poles[i + 1] if i + 1 < n else f(poles[i] + w2.sum())   # last root < p_n + |v|^2
        # choose the nearer pole as origin by the sign of f at the midpoint
        mid = f(0.5) * (hi_p - lo_p)

        def feval(origin, tau):
            # f(origin_pole + tau) with differences formed relative to the origin pole: (p_k - p_o) - tau
            delta = (poles - poles[origin]).astype(f) - tau
            return f(1.0) + np.sum(w2 / delta, dtype=f)

        fm = feval(i, mid)
        if i + 1 < n and fm < 0:       # root in the upper half: measure from the upper pole
            origin, a, b = i + 1, -(hi_p - lo_p) + mid, f(0.0)
        else:
            origin, a, b = i, f(0.0), (mid if i + 1 < n else hi_p - lo_p)
        # bisection in tau (f is increasing between two poles); 60 halvings are plenty for float32
        for _ in range(60):
            t = f(0.5) * (a + b)
            if t == a or t == b:
                break
            if feval(origin, t) > 0:
                b = t
            else:
                a = t
        roots_shift[i], base[i] = f(0.5) * (a + b), origin
    return poles, w2, order, base, roots_shift


def run(K, mean_scale, trials=200):
    worst = np.zeros(3)
    for _ in range(trials):
        N, d = pencil(K, mean_scale)
        M = np.block([[N + np.outer(d, d), d[:, None]], [d[None, :], np.ones((1, 1))]])
        lam64, U64 = np.linalg.eigh(M)
        d2_ref = 0.5 * np.sum(np.log(lam64) ** 2)
        g = lambda x: 2 * np.log(x) / x
        G_ref = (U64 * g(lam64)) @ U64.T
        nu, V = np.linalg.eigh(N)                      # stands for the K x K Jacobi (exact here)
        c = V.T @ d
        poles, w2, order, base, shift = secular_f32(nu, c)
        f = np.float32
        lam32 = (poles[base] + shift).astype(f)
        # eigenvectors in the (V,1) basis: w ~ v_k / (p_k - lam), differences from the shifted form
        v = np.concatenate([[1.0], c])[order].astype(f)
        W = np.zeros((len(poles), len(poles)), dtype=f)
        for i in range(len(poles)):
            delta = (poles - poles[base[i]]).astype(f) - shift[i]
            w = v / delta
            W[:, i] = w / np.sqrt(np.sum(w * w, dtype=f))
        d2 = 0.5 * np.sum(np.log(lam32.astype(np.float64)) ** 2)
        # back to the original coordinates: rows ordered (0-pole = last coordinate, nu-poles = V basis)
        T = np.zeros((K + 1, K + 1))
        T[:K, :K] = V; T[K, K] = 1.0
        P = np.zeros((K + 1, K + 1))
        for new, old in enumerate(order):
            P[(K if old == 0 else old - 1), new] = 1.0   # pole `old` (0 = the border) sits at sorted position `new`
        U32 = T @ P @ W.astype(np.float64)
        G = (U32 * g(lam32.astype(np.float64))) @ U32.T
        worst = np.maximum(worst, [abs(d2 - d2_ref) / d2_ref, np.linalg.norm(G - G_ref) / np.linalg.norm(G_ref),
                                   np.abs(U32.T @ U32 - np.eye(K + 1)).max()])
    print(f"K={K} mean scale {mean_scale}: worst over {trials} pairs: d^2 rel err {worst[0]:.1e}, spectral function rel err "
          f"{worst[1]:.1e}, eigenvector orthogonality defect {worst[2]:.1e}")


for K in (4, 16, 32):
    for ms in (0.3, 0.03, 1e-4, 0.0):
        run(K, ms, trials=60 if K == 32 else 200)
