"""torch <-> C-ABI glue for the pairwise SPD-distance kernels.

torch is used for device memory, streams and autograd bookkeeping only; all arithmetic of
the hot path happens inside libsqfa_hip.so.  The single call site of the library is
``hip_pair_backend`` (bound to the module attribute ``_pair_backend`` that the autograd
functions call through); it refuses CPU tensors and there is no other implementation in
this package.
"""
import ctypes

import torch

from . import _lib

EPSILON = 1e-6  # value added inside square roots (reference: src/sqfa/distances.py:29)


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.SQFA_F32
    if t.dtype == torch.float64:
        return _lib.SQFA_F64
    raise TypeError(f"sqfa_amd kernels support float32 and float64, got {t.dtype}")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def max_dim():
    return _lib.load().sqfa_hip_max_dim()


# Per-call policies of the pair kernels (sqfa_airm_options, include/sqfa_hip.h), read at every launch -- from whichever
# thread launches (the autograd engine runs backward passes on its own).  The library itself holds no policy state.
#   geometry      0 small-launch lane geometries by pair count (default), 1 wherever one exists, -1 never
#   class_factor  0 class factor pass K0b by pair count (default), 1 always, -1 never
#   sweep_counter None, or an int64 CUDA tensor of two elements {sum of Jacobi sweeps, wave rounds} the kernel adds to
#   mean_metric   1 the factor pass works in the metric of the mean class where it can (opt-in: pays for classes that share a
#                 dominant covariance, loses 1-3 % on BASELINE's synthetic generator); 0 (default) plain inner product
POLICY = {"geometry": 0, "class_factor": 0, "sweep_counter": None, "mean_metric": 0}


class policies:
    """with policies(geometry=-1, class_factor=1): ...   (tests, A/B timing; restores the previous values)"""

    def __init__(self, **kw):
        unknown = set(kw) - set(POLICY)
        if unknown:
            raise TypeError(f"unknown policy {sorted(unknown)}")
        self.kw = kw

    def __enter__(self):
        self.saved = dict(POLICY)
        POLICY.update(self.kw)
        return self

    def __exit__(self, *exc):
        POLICY.clear()
        POLICY.update(self.saved)
        return False


def _options():
    cnt = POLICY["sweep_counter"]
    return _lib.AirmOptions(int(POLICY["geometry"]), int(POLICY["class_factor"]),
                            ctypes.c_void_p(cnt.data_ptr()) if cnt is not None else None, int(POLICY["mean_metric"]))


def hip_pair_backend(A, B, *, scale, eps, sqrt_mode, weights, uniform_weight, shard,
                     want_loss, want_grad, want_dist, want_eig, out_loss=None, out_gradA=None):
    """Run sqfa_airm_pairwise on the current stream.  A (nA,m,m); B (nB,m,m) or None (self).
    Returns dict(loss, gradA, gradB, dist, eig, nonfinite) of freshly allocated tensors
    (None where not requested)."""
    lib = _lib.load()
    if not A.is_cuda:
        raise RuntimeError(
            "sqfa_amd computes pairwise SPD distances on the GPU only (no CPU fallback): "
            "move the statistics/model to a HIP device, or pass your own distance_fun."
        )
    if B is not None and (B.device != A.device or B.dtype != A.dtype):
        raise ValueError("A and B must share device and dtype")
    A = A.detach().contiguous()
    nA, m = A.shape[0], A.shape[-1]
    if B is not None:
        B = B.detach().contiguous()
        nB = B.shape[0]
    else:
        nB = 0
    code = _dtype_code(A)
    if m > lib.sqfa_hip_max_dim():
        raise NotImplementedError(
            f"matrix size {m} exceeds the largest size the native kernels handle ({lib.sqfa_hip_max_dim()})"
        )
    opts = _options()
    nbytes = lib.sqfa_airm_workspace_bytes_sharded(nA, nB, m, code, int(shard[1]), opts.geometry_policy)
    if nbytes == 0:
        raise _lib.NativeLibraryError("sqfa_airm_workspace_bytes rejected the problem shape")
    dev = A.device
    nBe = nA if B is None else nB
    with torch.cuda.device(dev):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # out_loss / out_gradA: caller-provided views (e.g. into a fused all-reduce buffer)
        # (loss and the two validity counters are always written by the finalize kernel: no memset)
        loss = (out_loss if out_loss is not None else torch.empty((), dtype=A.dtype, device=dev)) if want_loss else None
        nonfinite = torch.empty(2, dtype=torch.int32, device=dev)
        gradA = (out_gradA if out_gradA is not None else torch.empty_like(A)) if want_grad else None
        gradB = torch.empty_like(B) if (want_grad and B is not None) else None
        if want_dist:
            dist = (torch.zeros if shard[1] > 1 else torch.empty)((nA, nBe), dtype=A.dtype, device=dev)
        else:
            dist = None
        eig = torch.empty((nA, nBe, m), dtype=A.dtype, device=dev) if want_eig else None
        if weights is not None:
            weights = weights.detach().to(dtype=A.dtype, device=dev).contiguous()
            if tuple(weights.shape) != (nA, nBe):
                raise ValueError("pair weights must have shape (nA, nB)")
        stream = torch.cuda.current_stream(dev).cuda_stream
        status = lib.sqfa_airm_pairwise_opt(
            _ptr(A), nA, _ptr(B), nB, m, code,
            float(scale), float(eps), int(bool(sqrt_mode)),
            _ptr(weights), float(uniform_weight),
            int(shard[0]), int(shard[1]),
            _ptr(loss), _ptr(gradA), _ptr(gradB), _ptr(dist), _ptr(eig), _ptr(nonfinite),
            _ptr(ws), nbytes, ctypes.c_void_p(stream), ctypes.byref(opts),
        )
    _lib.check(status, "sqfa_airm_pairwise_opt")
    return {"loss": loss, "gradA": gradA, "gradB": gradB, "dist": dist, "eig": eig, "nonfinite": nonfinite}


_pair_backend = hip_pair_backend


# ------------------------------------------------------------------------------------------
# autograd wrappers


class PairDistanceMatrix(torch.autograd.Function):
    """D[i,j] = dist(A_i, B_j) as a differentiable (nA,nB) matrix.  The backward pass
    re-evaluates the pairs with the incoming gradient as per-pair weights (nothing of
    size nA*nB*m*m is ever stored)."""

    @staticmethod
    def forward(ctx, A, B, scale, eps, sqrt_mode):
        out = _pair_backend(A, B, scale=scale, eps=eps, sqrt_mode=sqrt_mode, weights=None,
                            uniform_weight=0.0, shard=(0, 1), want_loss=False, want_grad=False,
                            want_dist=True, want_eig=False)
        ctx.save_for_backward(A, B if B is not None else A.new_empty(0))
        ctx.self_mode = B is None
        ctx.cfg = (scale, eps, sqrt_mode)
        ctx.mark_non_differentiable(out["nonfinite"])
        return out["dist"], out["nonfinite"]

    @staticmethod
    def backward(ctx, gD, _gflag):
        A, B = ctx.saved_tensors
        B = None if ctx.self_mode else B
        scale, eps, sqrt_mode = ctx.cfg
        out = _pair_backend(A, B, scale=scale, eps=eps, sqrt_mode=sqrt_mode, weights=gD,
                            uniform_weight=0.0, shard=(0, 1), want_loss=False, want_grad=True,
                            want_dist=False, want_eig=False)
        return out["gradA"], out["gradB"], None, None, None


class PairwiseLoss(torch.autograd.Function):
    """Fused closure loss: sum over the unordered pairs i>j of  weight * dist(S_i, S_j)
    (weight = -1/P gives the reference's -mean, src/sqfa/_optim.py:94) together with its
    gradient, in one pass.  `reducer(loss, nonfinite, grad)` combines shards (all-reduce)."""

    @staticmethod
    def forward(ctx, S, scale, eps, sqrt_mode, weight, shard, reducer):
        extra = {}
        fused = None
        owner = getattr(reducer, "__self__", None)
        if (reducer is not None and shard[1] > 1 and _pair_backend is hip_pair_backend
                and hasattr(owner, "reduce_fused")):
            # the kernel writes loss and gradient straight into the all-reduce buffer
            # [loss, nan, inf, grad...]: no packing copies
            fused = torch.empty(S.numel() + 3, dtype=S.dtype, device=S.device)
            extra = {"out_loss": fused[0], "out_gradA": fused[3:].view(S.shape)}
        out = _pair_backend(S, None, scale=scale, eps=eps, sqrt_mode=sqrt_mode, weights=None,
                            uniform_weight=weight, shard=shard, want_loss=True, want_grad=True,
                            want_dist=False, want_eig=False, **extra)
        loss, nonfinite, grad = out["loss"], out["nonfinite"], out["gradA"]
        if fused is not None:
            loss, nonfinite, grad = owner.reduce_fused(fused, nonfinite, S.shape)
        elif reducer is not None:
            loss, nonfinite, grad = reducer(loss, nonfinite, grad)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(nonfinite)
        return loss, nonfinite

    @staticmethod
    def backward(ctx, gloss, _gflag):
        (grad,) = ctx.saved_tensors
        return grad * gloss, None, None, None, None, None, None


def hip_eigenvalues_backward(A, B, eig_weights):
    """sqfa_airm_eigenvalues_backward on the current stream: gradients of sum(eig_weights * eig)
    wrt A (nA,m,m) and B (nB,m,m), `eig_weights` (nA,nB,m) in the kernel's unsorted column order."""
    lib = _lib.load()
    if not A.is_cuda:
        raise RuntimeError("sqfa_amd computes generalized eigenvalues on the GPU only (no CPU fallback)")
    A = A.detach().contiguous()
    B = B.detach().contiguous()
    W = eig_weights.detach().to(A.dtype).contiguous()
    nA, nB, m = A.shape[0], B.shape[0], A.shape[-1]
    code = _dtype_code(A)
    nbytes = lib.sqfa_airm_workspace_bytes(nA, nB, m, code)
    if nbytes == 0:
        raise _lib.NativeLibraryError("sqfa_airm_workspace_bytes rejected the problem shape")
    with torch.cuda.device(A.device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
        gA, gB = torch.empty_like(A), torch.empty_like(B)
        stream = torch.cuda.current_stream(A.device).cuda_stream
        opts = _options()   # the policies of the forward call: the kernel's eigenvalue order depends on the lane geometry
        status = lib.sqfa_airm_eigenvalues_backward(_ptr(A), nA, _ptr(B), nB, m, code, _ptr(W), _ptr(gA), _ptr(gB),
                                                    _ptr(ws), nbytes, ctypes.c_void_p(stream), ctypes.byref(opts))
    _lib.check(status, "sqfa_airm_eigenvalues_backward")
    return gA, gB


_eig_backward_backend = hip_eigenvalues_backward


class GeneralizedEigenvalues(torch.autograd.Function):
    """(nA,nB,m) generalized eigenvalues of every pair (A_i, B_j), descending, differentiable
    (the reference's generalized_eigenvalues is autograd-transparent, src/sqfa/linalg.py:48-70).
    The kernel returns the eigenvalues in its own (deterministic) column order; the sort
    permutation is kept and the upstream gradient is scattered back through it, then one
    more launch evaluates sum_k w_k dlambda_k/d(A,B) in closed form (u u^T, -lambda u u^T)."""

    @staticmethod
    def forward(ctx, A, B):
        out = _pair_backend(A, B, scale=1.0, eps=EPSILON, sqrt_mode=False, weights=None,
                            uniform_weight=0.0, shard=(0, 1), want_loss=False, want_grad=False,
                            want_dist=False, want_eig=True)
        lam, order = torch.sort(out["eig"], dim=-1, descending=True)
        ctx.save_for_backward(A, B, order)
        return lam

    @staticmethod
    def backward(ctx, g_lam):
        A, B, order = ctx.saved_tensors
        w = torch.zeros_like(g_lam).scatter_(-1, order, g_lam)
        gA, gB = _eig_backward_backend(A, B, w)
        return gA, gB


def generalized_eigenvalues_raw(A, B):
    """(nA,nB,m) generalized eigenvalues of (A_i, B_j), descending; differentiable wrt A and B."""
    return GeneralizedEigenvalues.apply(A, B)


# ------------------------------------------------------------------------------------------
# Gaussian pair terms (bhattacharyya / mahalanobis / hellinger / fisher_rao_same_cov)


def hip_gauss_terms(muA, covA, muB, covB, gQ=None, gLD=None, want_outputs=True, want_grad=False):
    """sqfa_gauss_pair_terms on the current stream.  Returns (Q, LD, gmuA, gcovA) (None where not requested)."""
    lib = _lib.load()
    if not covA.is_cuda:
        raise RuntimeError("sqfa_amd's native Gaussian pair terms run on the GPU only")
    muA, covA, muB, covB = (t.detach().contiguous() for t in (muA, covA, muB, covB))
    nA, nB, m = covA.shape[0], covB.shape[0], covA.shape[-1]
    m_true = m
    if m < 16 and m % 4 != 0 and nA * nB >= 20000:
        # the one-pair-per-lane kernel reads whole 16-byte rows when the matrices ARE its padded size (4 / 8 / 12 / 16):
        # pad with an identity block here (exact: logdet + log 1, Q unchanged) instead of letting it pad with scalar
        # loads and selects (K=15 at C=1000: 1.11 -> 0.55 ms forward + backward)
        M = (m + 3) // 4 * 4

        def pad(mu, cov):
            cov_p = torch.eye(M, dtype=cov.dtype, device=cov.device).repeat(cov.shape[0], 1, 1)
            cov_p[:, :m, :m] = cov
            return torch.nn.functional.pad(mu, (0, M - m)), cov_p

        same = muB.data_ptr() == muA.data_ptr() and covB.data_ptr() == covA.data_ptr() and nA == nB
        muA, covA = pad(muA, covA)
        muB, covB = (muA, covA) if same else pad(muB, covB)
        m = M
    code = _dtype_code(covA)
    dev, dt = covA.device, covA.dtype
    with torch.cuda.device(dev):
        Q = torch.empty((nA, nB), dtype=dt, device=dev) if want_outputs else None
        LD = torch.empty((nA, nB), dtype=dt, device=dev) if want_outputs else None
        gmu = torch.empty((nA, m), dtype=dt, device=dev) if want_grad else None
        gcov = torch.empty((nA, m, m), dtype=dt, device=dev) if want_grad else None
        gQ = gQ.detach().to(dt).contiguous() if gQ is not None else None
        gLD = gLD.detach().to(dt).contiguous() if gLD is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        status = lib.sqfa_gauss_pair_terms(_ptr(muA), _ptr(covA), nA, _ptr(muB), _ptr(covB), nB, m, code,
                                           _ptr(gQ), _ptr(gLD), _ptr(Q), _ptr(LD), _ptr(gmu), _ptr(gcov), stream)
    _lib.check(status, "sqfa_gauss_pair_terms")
    if m != m_true and want_grad:
        gmu, gcov = gmu[:, :m_true].contiguous(), gcov[:, :m_true, :m_true].contiguous()
    return Q, LD, gmu, gcov


GAUSS_MAX_DIM = 64


class GaussPairTerms(torch.autograd.Function):
    """Q_ij = (mu_i-mu_j)^T Sbar_ij^-1 (mu_i-mu_j) and LD_ij = logdet Sbar_ij with
    Sbar_ij = (Sigma_i + Sigma_j)/2, as differentiable (nA,nB) matrices -- the pair-dependent part of
    the reference's bhattacharyya / mahalanobis / hellinger / fisher_rao_same_cov
    (src/sqfa/distances.py:240-432), without the (nA,nB,K,K) tensor those build.  `same` marks the
    self case (B is A): one backward launch with symmetrised upstream gradients."""

    @staticmethod
    def forward(ctx, muA, covA, muB, covB, same):
        Q, LD, _, _ = hip_gauss_terms(muA, covA, muB, covB)
        ctx.save_for_backward(muA, covA, muB, covB)
        ctx.same = same
        return Q, LD

    @staticmethod
    def backward(ctx, gQ, gLD):
        muA, covA, muB, covB = ctx.saved_tensors
        if ctx.same:
            _, _, gmu, gcov = hip_gauss_terms(muA, covA, muA, covA, gQ + gQ.t(), gLD + gLD.t(),
                                              want_outputs=False, want_grad=True)
            return gmu, gcov, None, None, None
        _, _, gmuA, gcovA = hip_gauss_terms(muA, covA, muB, covB, gQ, gLD, want_outputs=False, want_grad=True)
        _, _, gmuB, gcovB = hip_gauss_terms(muB, covB, muA, covA, gQ.t(), gLD.t(), want_outputs=False, want_grad=True)
        return gmuA, gcovA, gmuB, gcovB, None


# ------------------------------------------------------------------------------------------
# per-class matrix functions (spd_log / spd_sqrt: log_euclidean's building block)

SPD_LOG, SPD_SQRT, SPD_INV_SQRT = 0, 1, 2


def spd_function_supported(M):
    return (M.is_cuda and M.dtype in (torch.float32, torch.float64) and M.dim() >= 2 and M.shape[-1] == M.shape[-2]
            and 1 <= M.shape[-1] <= max_dim() and M.numel() > 0)


class SpdFunction(torch.autograd.Function):
    """f(S) = Q f(Lambda) Q^T for a batch (n,m,m) of SPD matrices through sqfa_spd_function (Cholesky + one-sided Jacobi
    per class, double inside), differentiable through sqfa_spd_function_backward (Daleckii-Krein).  Replaces the
    reference's torch.linalg.eigh + einsum (src/sqfa/linalg.py:121-141, 165-183) and eigh's autograd backward."""

    @staticmethod
    def forward(ctx, S, kind):
        lib = _lib.load()
        S3 = S.detach().contiguous()
        n, m = S3.shape[0], S3.shape[-1]
        code = _dtype_code(S3)
        nbytes = lib.sqfa_spd_function_workspace_bytes(n, m, code)
        if nbytes == 0:
            raise _lib.NativeLibraryError("sqfa_spd_function_workspace_bytes rejected the problem shape")
        dev = S3.device
        with torch.cuda.device(dev):
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            F = torch.empty_like(S3)
            U = torch.empty((n, m, m), dtype=torch.float64, device=dev)
            lam = torch.empty((n, m), dtype=torch.float64, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sqfa_spd_function(_ptr(S3), n, m, code, int(kind), _ptr(F), _ptr(U), _ptr(lam), _ptr(ws), nbytes, stream),
                       "sqfa_spd_function")
        ctx.save_for_backward(U, lam)
        ctx.kind = int(kind)
        return F

    @staticmethod
    def backward(ctx, gF):
        U, lam = ctx.saved_tensors
        lib = _lib.load()
        G = gF.detach().contiguous()
        n, m = G.shape[0], G.shape[-1]
        with torch.cuda.device(G.device):
            gS = torch.empty_like(G)
            stream = ctypes.c_void_p(torch.cuda.current_stream(G.device).cuda_stream)
            _lib.check(lib.sqfa_spd_function_backward(_ptr(U), _ptr(lam), _ptr(G), n, m, _dtype_code(G), ctx.kind, _ptr(gS), stream),
                       "sqfa_spd_function_backward")
        return gS, None


def spd_function(M, kind):
    """(..., m, m) -> (..., m, m) on the native kernels (the caller has checked spd_function_supported)."""
    lead = M.shape[:-2]
    out = SpdFunction.apply(M.reshape(-1, M.shape[-2], M.shape[-1]), kind)
    return out.reshape(*lead, M.shape[-2], M.shape[-1])


# ------------------------------------------------------------------------------------------
# projection of the class scatter matrices (the HBM-bound stage around the pair kernel)


_symmetry_check_warned = False
_symmetry_checked = {}   # id(tensor) -> (weakref to the tensor, _version, verdict): dies with the tensor, never reused by another


def _is_symmetric_batch(scatters):
    """The streaming kernel forms T = Psi^T F^T, which equals the reference's Psi F^T only for
    symmetric Psi (covariance / second-moment matrices are; conjugate_matrix itself is general,
    src/sqfa/linalg.py:19-45).  Checked ONCE per tensor object and version (one pass over Psi in class chunks of
    at most ~256 MB -- the c4 statistics are 16.8 GB, whole-tensor temporaries would triple that -- and one host
    read), outside any graph capture; asymmetry at rounding level (a GEMM-built X^T X) passes.  The verdict is
    keyed on the tensor OBJECT (weak reference), not on its address: the caching allocator hands a freed address
    to the next tensor.  Any failure of the check itself (e.g. out of memory) selects the general torch expression."""
    import weakref
    hit = _symmetry_checked.get(id(scatters))
    if hit is not None and hit[0]() is scatters and hit[1] == scatters._version:
        return hit[2]
    if torch.cuda.is_current_stream_capturing():
        return False  # never seen outside a capture: take the general torch expression
    tol = 1e-5 if scatters.dtype == torch.float32 else 1e-12
    try:
        with torch.no_grad():
            per_class = scatters[0].numel() * scatters.element_size()
            step = max(1, (256 << 20) // max(per_class, 1))
            asym = scatters.new_zeros(())
            peak = scatters.new_zeros(())
            for c0 in range(0, scatters.shape[0], step):
                blk = scatters[c0:c0 + step]
                asym = torch.maximum(asym, (blk - blk.transpose(-2, -1)).abs().amax())
                peak = torch.maximum(peak, blk.abs().amax())
            ok = bool(asym <= tol * peak)
    except Exception as err:   # the check must never take the fit down: the torch expression handles every input
        ok = False
        global _symmetry_check_warned
        if not _symmetry_check_warned:
            _symmetry_check_warned = True
            import warnings
            warnings.warn(f"sqfa_amd: the symmetry check of the scatter matrices itself failed ({err!r}); this tensor keeps the "
                          "general torch projection (slower, same results)")
    key = id(scatters)
    try:
        ref = weakref.ref(scatters, lambda _r, k=key: _symmetry_checked.pop(k, None))
    except TypeError:
        return ok
    _symmetry_checked[key] = (ref, scatters._version, ok)
    return ok


def native_projection_supported(scatters, filters):
    """The streaming kernel handles SYMMETRIC float32/float64 (C,D,D) scatters on the GPU with
    D % 4 == 0, 16-byte aligned storage and up to 64 filters; anything else keeps the plain torch
    expression (conjugate_matrix)."""
    return (
        scatters.is_cuda and filters.is_cuda and scatters.dtype in (torch.float32, torch.float64)
        and filters.dtype == scatters.dtype
        and scatters.dim() == 3 and filters.dim() == 2 and scatters.shape[-1] % 4 == 0
        and filters.shape[0] <= 64 and filters.shape[0] <= filters.shape[1]
        and scatters.shape[-1] == scatters.shape[-2] == filters.shape[1] and not scatters.requires_grad
        and scatters.is_contiguous() and scatters.data_ptr() % 16 == 0
        and _is_symmetric_batch(scatters)
    )


# ---- block-triangular packed statistics (sqfa_pack_scatters / sqfa_project_scatters_packed) ------------------------------
# The symmetric (C,D,D) statistics are packed ONCE per prepared tensor (model._prepare_statistics) into their lower block
# triangle; every closure then streams 51-54 % of the bytes.  The packed copy lives beside the caller's tensor (which the
# caller owns and may free: statistics.pack_scatters hands the packed form out for that case).
# Measured (tools/time_projection_packed.py, profiles/r4_projection_packed.txt; full tensor -> packed, two boxes): c3 (C=1000,
# D=784, K=16) 0.43-0.45 -> 0.35-0.37 ms, K=8 0.43 -> 0.33-0.35, D=1024 0.63-0.67 -> 0.55-0.58, D=2048 / K=16 2.70 -> 2.29-2.40;
# it LOSES where the walk is short or the chip is not full: D=512 0.169 -> 0.186, C=256 0.121 -> 0.143, C=500 / D=3072 a tie;
# K=32 (c4) 3.02 -> 4.4-4.6 ms (two filter blocks: twice the exact-f32 MFMA work per byte, MFMA-bound); C=100 (c5) 0.58 -> 1.1-1.2 ms
# (one workgroup per class leaves 60 % of the CUs idle; splitting a class over several workgroups with a second pass over
# their partial results was built and measured at 1.08 ms: the per-row-block barrier of the walk dominates there; removed).
PACKED_PROJECTION = True
PACKED_MIN_CLASSES = 512      # one workgroup per class, two resident per CU
PACKED_MIN_DIM = 768
PACKED_MAX_FILTERS = 16       # one 16-filter block: beyond it the packed kernel is MFMA-bound and loses to the full-tensor stream
_packed_cache = {}            # id(scatters) -> (weakref, _version, packed tensor)


def packed_supported(scatters):
    return (PACKED_PROJECTION and scatters.is_cuda and scatters.dtype == torch.float32 and scatters.dim() == 3
            and scatters.shape[-1] == scatters.shape[-2] and scatters.shape[-1] % 16 == 0 and PACKED_MIN_DIM <= scatters.shape[-1] <= 4096
            and scatters.shape[0] >= PACKED_MIN_CLASSES and scatters.is_contiguous() and scatters.data_ptr() % 16 == 0
            and not scatters.requires_grad)


def pack_scatters(scatters):
    """(C,D,D) symmetric float32 scatters -> (C, packed_elems) block-triangular packed form (sqfa_pack_scatters)."""
    lib = _lib.load()
    C, D = scatters.shape[0], scatters.shape[-1]
    n = lib.sqfa_packed_scatter_elems(D)
    if n == 0:
        raise ValueError(f"n_dim = {D} has no packed form (needs D % 16 == 0)")
    with torch.cuda.device(scatters.device):
        out = torch.empty((C, n), dtype=scatters.dtype, device=scatters.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(scatters.device).cuda_stream)
        _lib.check(lib.sqfa_pack_scatters(_ptr(scatters), C, D, _dtype_code(scatters), _ptr(out), stream), "sqfa_pack_scatters")
    return out


def prepare_packed(scatters, n_filters):
    """Pack `scatters` once (outside any graph capture) if the packed projection applies to it and to this filter count;
    later projections of the same tensor object / version find the packed copy through packed_for."""
    import weakref
    if (not torch.is_tensor(scatters) or n_filters > PACKED_MAX_FILTERS or not packed_supported(scatters)
            or torch.cuda.is_current_stream_capturing()):
        return None
    hit = packed_for(scatters)
    if hit is not None:
        return hit
    if not _is_symmetric_batch(scatters):
        return None
    packed = pack_scatters(scatters)
    key = id(scatters)
    try:
        ref = weakref.ref(scatters, lambda _r, k=key: _packed_cache.pop(k, None))
    except TypeError:
        return None
    _packed_cache[key] = (ref, scatters._version, packed)
    return packed


def packed_for(scatters, n_filters=1):
    hit = _packed_cache.get(id(scatters))
    if (hit is not None and hit[0]() is scatters and hit[1] == scatters._version and PACKED_PROJECTION
            and n_filters <= PACKED_MAX_FILTERS):
        return hit[2]
    return None


def _launch_projection(lib, F, Psi, T, K, D, C, code, stream):
    """T = Psi F^T: from the packed copy when one was prepared for this tensor, else from the full tensor."""
    packed = packed_for(Psi, K)
    if packed is not None:
        _lib.check(lib.sqfa_project_scatters_packed(_ptr(F), K, D, _ptr(packed), C, code, _ptr(T), stream),
                   "sqfa_project_scatters_packed")
    else:
        _lib.check(lib.sqfa_project_scatters(_ptr(F), K, D, _ptr(Psi), C, code, _ptr(T), stream), "sqfa_project_scatters")


def backward_groups(C, D, cap=64):
    """Class groups of sqfa_feature_scatters_backward: the kernel wants ~3000 waves in flight (one per 16-column
    block and group; it is latency-bound), while the (groups, K, D) partial sums are re-read by the reduction that
    follows, whose time grows with the group count.  Measured (tools/time_feature_backward.py, backward + reduction):
    c3 (D=784) 64 groups 34 us (16: 43), c4 (D=2048) 32: 128 us (64: 139, 16: 151), c5 (D=3072, C=100) 16: 28 us
    (64: 56)."""
    blocks = (D + 15) // 16
    want = max(1, -(-3072 // blocks))
    groups = 8
    while groups < want:
        groups *= 2
    return max(1, min(groups, cap, C))


class ProjectScatters(torch.autograd.Function):
    """S_c = F Psi_c F^T for symmetric Psi_c, reading Psi (C,D,D) from HBM once.

    forward : T = Psi F^T through sqfa_project_scatters (streaming MFMA kernel), S = F T through
              sqfa_feature_scatters
    backward: dL/dF = sum_c (G_c + G_c^T) T_c^T through sqfa_feature_scatters_backward -- needs
              T (C,D,K) only, not Psi.
    Replaces conjugate_matrix (reference src/sqfa/linalg.py:19-45) in transform_scatters
    (src/sqfa/model.py:172-188), whose autograd reads Psi a second time in the backward."""

    # class groups of the backward kernel (one wave per 16-column block and group): see backward_groups
    BACKWARD_GROUPS = 64
    # S = F T and dL/dF go through the HIP kernels from this many classes on (measured, closure at
    # C=1000, D=784, K=16: 1.61 -> 1.56 ms); below, torch's batched GEMMs are as fast (both are
    # launch/latency-bound there).  A huge value selects the torch expressions everywhere.
    NATIVE_PRODUCTS_MIN_CLASSES = 256

    @staticmethod
    def forward(ctx, filters, scatters):
        lib = _lib.load()
        F = filters.detach().contiguous()
        Psi = scatters.detach().contiguous()
        K, D = F.shape
        C = Psi.shape[0]
        code = _dtype_code(Psi)
        with torch.cuda.device(Psi.device):
            T = torch.empty((C, D, K), dtype=Psi.dtype, device=Psi.device)
            S = torch.empty((C, K, K), dtype=Psi.dtype, device=Psi.device)
            stream = ctypes.c_void_p(torch.cuda.current_stream(Psi.device).cuda_stream)
            _launch_projection(lib, F, scatters if scatters.is_contiguous() else Psi, T, K, D, C, code, stream)
            if C >= ProjectScatters.NATIVE_PRODUCTS_MIN_CLASSES:
                _lib.check(lib.sqfa_feature_scatters(_ptr(F), K, D, _ptr(T), C, code, _ptr(S), stream),
                           "sqfa_feature_scatters")
            else:
                S = torch.matmul(F.unsqueeze(0), T)
        ctx.save_for_backward(T)
        return S

    @staticmethod
    def backward(ctx, gS):
        (T,) = ctx.saved_tensors
        C, D, K = T.shape
        if C >= ProjectScatters.NATIVE_PRODUCTS_MIN_CLASSES:
            lib = _lib.load()
            G = gS.contiguous()
            groups = backward_groups(C, D, ProjectScatters.BACKWARD_GROUPS)
            with torch.cuda.device(T.device):
                partial = torch.empty((groups, K, D), dtype=T.dtype, device=T.device)
                stream = ctypes.c_void_p(torch.cuda.current_stream(T.device).cuda_stream)
                _lib.check(lib.sqfa_feature_scatters_backward(_ptr(G), _ptr(T), C, D, K, _dtype_code(T), groups,
                                                              _ptr(partial), stream), "sqfa_feature_scatters_backward")
            return partial.sum(dim=0), None
        # few classes (launch-bound either way) or a filter count that is not a multiple of 4:
        # per-class (K,K)@(K,D) products, then the sum over classes.  (One flat GEMM (K, C*K)@(C*K, D) is the same arithmetic but hipBLASLt is
        # erratic for that skinny shape: 8 ms at C=1000, K=16 in isolation.)
        sym = gS + gS.transpose(1, 2)
        return torch.bmm(sym, T.transpose(1, 2)).sum(dim=0), None


def project_scatters(scatters, filters):
    """(C,D,D) x (K,D) -> (C,K,K); native streaming kernel when supported, torch otherwise."""
    if native_projection_supported(scatters, filters):
        return ProjectScatters.apply(filters, scatters)
    return None


# ------------------------------------------------------------------------------------------
# the whole closure as one autograd node


def fused_closure_supported(raw_filters, scatters, means):
    """Conditions of the single-node closure: the streaming projection's (symmetric (C,D,D) scatters,
    D % 4 == 0, K <= 64, 16-byte aligned), float32/float64 on the GPU; the means (SQFA) on the same
    device/dtype."""
    K = raw_filters.shape[0]
    if not (raw_filters.is_cuda and raw_filters.dim() == 2 and raw_filters.is_contiguous()):
        return False
    if means is not None and not (means.is_cuda and means.dtype == scatters.dtype and means.dim() == 2
                                  and not means.requires_grad):
        return False
    return native_projection_supported(scatters, raw_filters)


def closure_stage_project(raw, scatters, means, noise, sphere, out_S=None):
    """First half of a closure evaluation, no autograd: sphere -> T = Psi F^T -> [m = mu F^T] -> S | E
    (feature scatters + noise, or their Calvo-Oller embedding), written into `out_S` when given (a slice of
    the all-gather buffer of a class-sharded evaluation).  Returns what the backward stage needs."""
    lib = _lib.load()
    X = raw.detach().contiguous()
    Psi = scatters.detach()
    K, D = X.shape
    C = Psi.shape[0]
    code = _dtype_code(Psi)
    dev, dt = Psi.device, Psi.dtype
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if sphere:
            F = torch.empty_like(X)
            norms = torch.empty(K, dtype=dt, device=dev)
            _lib.check(lib.sqfa_sphere_forward(_ptr(X), K, D, code, _ptr(F), _ptr(norms), stream), "sqfa_sphere_forward")
        else:
            F, norms = X, None
        T = torch.empty((C, D, K), dtype=dt, device=dev)
        _launch_projection(lib, F, scatters, T, K, D, C, code, stream)   # keyed on the caller's tensor object (packed copy)
        msize = K + 1 if means is not None else K
        m = torch.matmul(means.detach(), F.t()).contiguous() if means is not None else None   # (C,K) projected means
        S = out_S if out_S is not None else torch.empty((C, msize, msize), dtype=dt, device=dev)
        if tuple(S.shape) != (C, msize, msize) or not S.is_contiguous():
            raise ValueError("out_S must be a contiguous (C, m, m) tensor")
        _lib.check(lib.sqfa_feature_scatters_ex(_ptr(F), K, D, _ptr(T), C, code, float(noise), _ptr(m), _ptr(S), stream),
                   "sqfa_feature_scatters_ex")
    return {"X": X, "norms": norms, "T": T, "m": m, "means": means.detach() if means is not None else None,
            "S": S, "S_shape": tuple(S.shape), "sphere": sphere}


def closure_stage_pairs(S, scale, sqrt_mode, weight, shard, fused=None):
    """Second half: K0 / K1 / K2 on the (C,m,m) batch.  `fused` (S.numel() + 3 elements), when given, receives
    [loss, nan, inf, dL/dS...] in place (the all-reduce buffer of a sharded evaluation)."""
    extra = {}
    if fused is not None:
        extra = {"out_loss": fused[0], "out_gradA": fused[3:].view(S.shape)}
    out = _pair_backend(S, None, scale=scale, eps=EPSILON, sqrt_mode=sqrt_mode, weights=None,
                        uniform_weight=weight, shard=shard, want_loss=True, want_grad=True,
                        want_dist=False, want_eig=False, **extra)
    if fused is not None:
        fused[1:3].copy_(out["nonfinite"])   # int32 -> real in the copy itself
    return out["loss"], out["nonfinite"], out["gradA"]


def closure_stage_forward(raw, scatters, means, noise, scale, sqrt_mode, weight, shard, sphere, fused=None):
    """Both halves (closure_stage_project + closure_stage_pairs) for statistics that are not class-sharded."""
    st = closure_stage_project(raw, scatters, means, noise, sphere)
    st["loss"], st["nonfinite"], st["gS"] = closure_stage_pairs(st["S"], scale, sqrt_mode, weight, shard, fused)
    return st


def closure_stage_backward(st, gS, gloss):
    """Kernels from dL/dS (or dL/dE) back to the raw filter parameter: backward product, [means path],
    class reduction + sphere backward (+ multiplication by `gloss`, a device scalar or None)."""
    lib = _lib.load()
    X, norms, T, m, means = st["X"], st["norms"], st["T"], st["m"], st["means"]
    C, D, K = T.shape
    code = _dtype_code(T)
    dev, dt = T.device, T.dtype
    groups = backward_groups(C, D, FusedClosure.BACKWARD_GROUPS)
    gS = gS.contiguous()
    ldg = gS.shape[-1]
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        partial = torch.empty((groups, K, D), dtype=dt, device=dev)
        # gS is the pair kernels' dL/dS (or dL/dE): written symmetric by sqfa_airm_pairwise, sums of shards included
        _lib.check(lib.sqfa_feature_scatters_backward_ex(_ptr(gS), ldg, _ptr(T), C, D, K, code, groups, 1, _ptr(partial), stream),
                   "sqfa_feature_scatters_backward_ex")
        extra = None
        if m is not None:
            gm = torch.empty((C, K), dtype=dt, device=dev)
            _lib.check(lib.sqfa_embed_backward_means(_ptr(gS), _ptr(m), C, K, code, _ptr(gm), stream), "sqfa_embed_backward_means")
            extra = torch.matmul(gm.t(), means).contiguous()          # (K,D): m = mu F^T  =>  dL/dF += g_m^T mu
        grad = torch.empty_like(X)
        gl = gloss.detach().to(dt).contiguous() if gloss is not None else None
        _lib.check(lib.sqfa_sphere_backward(_ptr(X), _ptr(norms if st["sphere"] else None), K, D, code, _ptr(partial), groups,
                                            _ptr(extra), _ptr(gl), _ptr(grad), stream), "sqfa_sphere_backward")
    return grad


class FusedClosure(torch.autograd.Function):
    """loss(raw filters) of one closure evaluation as ONE autograd node, 8 launches for
    SecondMomentsSQFA and 11 for SQFA instead of ~40:

        forward   sphere (or nothing)  ->  T = Psi F^T (streaming)  ->  [m = mu F^T]  ->
                  S = F T + noise I  |  E = [[S + m m^T, m], [m^T, 1]]  ->  K0 / K1 / K2 (loss, dL/dS, flags)
        backward  (G + G^T) T^T partial sums  ->  [g_m, g_m^T mu]  ->  class reduction + sphere backward + gloss

    Same arithmetic as the chain Sphere -> ProjectScatters -> (+noise) -> embed_gaussian -> PairwiseLoss
    it replaces (reference: src/sqfa/constraints.py:37, src/sqfa/model.py:172-188, 508-546,
    src/sqfa/distances.py:141-174, src/sqfa/_optim.py:90-96); the summation order of the row norms
    and of the class reduction differs (fixed, reproducible)."""

    BACKWARD_GROUPS = 64

    @staticmethod
    def forward(ctx, raw, scatters, means, noise, scale, sqrt_mode, weight, shard, reducer, sphere):
        fused = None
        owner = getattr(reducer, "__self__", None)
        if reducer is not None and shard[1] > 1 and _pair_backend is hip_pair_backend and hasattr(owner, "reduce_fused"):
            C, K = scatters.shape[0], raw.shape[0]
            m = K + 1 if means is not None else K
            fused = torch.empty(C * m * m + 3, dtype=scatters.dtype, device=scatters.device)
        st = closure_stage_forward(raw, scatters, means, noise, scale, sqrt_mode, weight, shard, sphere, fused)
        loss, nonfinite, gS = st["loss"], st["nonfinite"], st["gS"]
        if fused is not None:
            loss, nonfinite, gS = owner.reduce_fused(fused, nonfinite, st["S_shape"])
        elif reducer is not None:
            loss, nonfinite, gS = reducer(loss, nonfinite, gS)
        ctx.st = {k: v for k, v in st.items() if k not in ("loss", "nonfinite", "gS", "S")}
        ctx.gS = gS
        ctx.mark_non_differentiable(nonfinite)
        return loss, nonfinite

    @staticmethod
    def backward(ctx, gloss, _gflag):
        grad = closure_stage_backward(ctx.st, ctx.gS, gloss)
        return grad, None, None, None, None, None, None, None, None, None
