"""GPU tests of the block-triangular packed projection (sqfa_pack_scatters / sqfa_project_scatters_packed,
sqfa_amd/csrc/project_packed_kernel.hip): the packed layout itself, T = Psi F^T against the float64 torch expression of
the reference's conjugate_matrix (src/sqfa/linalg.py:19-45), bitwise reproducibility, and a model closure on packed
statistics against the same closure on the full tensor."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sym(C, D, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(C, D, min(D, 96), generator=g, dtype=torch.float64)
    return (A @ A.transpose(1, 2) / A.shape[-1] + 0.05 * torch.eye(D, dtype=torch.float64))


def _packed_reference(Psi):
    """numpy construction of the documented layout: row blocks of 16 rows, each the 16 x 64 tiles of stripes 0 .. R/64."""
    C, D, _ = Psi.shape
    out = []
    for rb in range(D // 16):
        R = 16 * rb
        for s in range(R // 64 + 1):
            tile = np.zeros((C, 16, 64), dtype=Psi.dtype)
            w = min(64, D - 64 * s)
            tile[:, :, :w] = Psi[:, R:R + 16, 64 * s:64 * s + w]
            out.append(tile.reshape(C, -1))
    return np.concatenate(out, axis=1)


@pytest.mark.parametrize("C,D", [(2, 16), (3, 64), (2, 80), (3, 128), (2, 144), (2, 784), (1, 1040)])
def test_packed_layout(C, D):
    from sqfa_amd import _lib, _native
    Psi = _sym(C, D, D).float()
    packed = _native.pack_scatters(Psi.to(DEV))
    ref = _packed_reference(Psi.numpy())
    assert packed.shape == ref.shape == (C, _lib.load().sqfa_packed_scatter_elems(D))
    assert np.array_equal(packed.cpu().numpy(), ref)


@pytest.mark.parametrize("C,D,K", [(3, 64, 2), (5, 128, 16), (4, 784, 16), (2, 2048, 32), (3, 144, 5), (2, 256, 64), (2, 3072, 16),
                                   (2, 1040, 33), (1, 4096, 8), (7, 16, 3), (3, 80, 16), (300, 96, 12), (2, 1024, 48), (2, 2048, 20)])
def test_packed_projection_vs_float64_expression(C, D, K):
    from sqfa_amd import _lib, _native
    lib = _lib.load()
    Psi64 = _sym(C, D, 7 * D + K)
    g = torch.Generator().manual_seed(K)
    F64 = torch.randn(K, D, generator=g, dtype=torch.float64)
    F64 = F64 / F64.norm(dim=1, keepdim=True)
    Psi, F = Psi64.float().to(DEV), F64.float().to(DEV)
    ref = (Psi.double() @ F.double().T).cpu().numpy()              # at the float32-rounded inputs
    packed = _native.pack_scatters(Psi)
    T = torch.full((C, D, K), float("nan"), device=DEV)
    T2 = torch.empty_like(T)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for out in (T, T2):
        assert lib.sqfa_project_scatters_packed(F.data_ptr(), K, D, packed.data_ptr(), C, 0, out.data_ptr(), stream) == 0
    full = torch.empty_like(T)
    assert lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, full.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(T).all()
    e_packed, e_full = rel_err(T.cpu(), ref), rel_err(full.cpu(), ref)
    print(f"C={C} D={D} K={K}: packed {e_packed:.2e}, full-tensor kernel {e_full:.2e}")
    assert e_packed <= 2e-6
    assert torch.equal(T, T2)                                      # bitwise reproducible


def test_unsupported_shapes_are_refused():
    from sqfa_amd import _lib
    lib = _lib.load()
    z = ctypes.c_void_p(4096)
    assert lib.sqfa_packed_scatter_elems(100) == 0 and lib.sqfa_packed_scatter_elems(8) == 0
    assert lib.sqfa_pack_scatters(z, 2, 100, 0, z, None) == -2
    assert lib.sqfa_pack_scatters(z, 2, 128, 1, z, None) == -2          # float64 keeps the full tensor
    assert lib.sqfa_project_scatters_packed(z, 4, 100, z, 2, 0, z, None) == -2
    assert lib.sqfa_project_scatters_packed(z, 65, 128, z, 2, 0, z, None) == -2
    assert lib.sqfa_project_scatters_packed(None, 4, 128, z, 2, 0, z, None) == -1


@pytest.mark.parametrize("model_name", ["smsqfa", "sqfa"])
def test_closure_on_packed_statistics_matches_full_tensor(model_name, monkeypatch):
    """A model closure (loss and gradient wrt the raw filters) with the statistics packed by _prepare_statistics against
    the same closure with the packed path switched off: the two differ by summation order only."""
    import sqfa_amd
    from sqfa_amd import _native
    C, D, K = 300, 128, 8
    g = torch.Generator().manual_seed(1)
    cov = _sym(C, D, 3).float().to(DEV)
    mu = (0.1 * torch.randn(C, D, generator=g)).to(DEV)
    torch.manual_seed(5)
    cls = sqfa_amd.model.SQFA if model_name == "sqfa" else sqfa_amd.model.SecondMomentsSQFA
    model = cls(n_dim=D, n_filters=K, feature_noise=0.01).to(DEV)
    stats = {"means": mu, "covariances": cov}
    out = {}
    monkeypatch.setattr(_native, "PACKED_MIN_CLASSES", 256)
    monkeypatch.setattr(_native, "PACKED_MIN_DIM", 64)
    for use in (True, False):
        monkeypatch.setattr(_native, "PACKED_PROJECTION", use)
        prepared = model._prepare_statistics(stats)
        scat = prepared["covariances"] if isinstance(prepared, dict) else prepared
        assert (_native.packed_for(scat) is not None) == use
        model.zero_grad()
        loss, flags = model._fused_closure_loss(prepared)
        loss.backward()
        out[use] = (loss.item(), model.parametrizations.filters.original.grad.clone(), flags.tolist())
    assert out[True][2] == [0, 0]
    assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])
    assert rel_err(out[True][1].cpu(), out[False][1].cpu().numpy()) <= 1e-5
    assert not torch.equal(out[True][1], out[False][1])               # the packed kernel did run


_prng = np.random.default_rng(2024)
RANDOM_SHAPES = [(int(_prng.integers(1, 7)), 16 * int(_prng.integers(1, 41)), int(_prng.integers(1, 65))) for _ in range(16)]


@pytest.mark.parametrize("C,D,K", [(c, d, min(k, d)) for c, d, k in RANDOM_SHAPES])
def test_packed_projection_random_shapes(C, D, K):
    """Seeded random (C, D, K) with D % 16 == 0: every stripe count / filter-block count / ragged last stripe the launcher's
    workgroup table can be asked for, against the float64 expression."""
    test_packed_projection_vs_float64_expression(C, D, K)
