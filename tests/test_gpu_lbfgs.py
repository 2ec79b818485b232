"""Device-side L-BFGS kernels (sqfa_amd/csrc/lbfgs_kernels.hip) against their torch expressions (GPU only; the CPU tests of
the optimizer itself live in tests/test_lbfgs.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
@pytest.mark.parametrize("n,h,pushes", [(300, 7, 5), (12544, 100, 130), (50000, 20, 45)])
def test_native_lbfgs_direction_matches_torch_compact_form(n, h, pushes, dtype, tol):
    """sqfa_lbfgs_push / sqfa_lbfgs_direction (six launches) against the torch compact form of the same
    recursion: same ring buffers, same SY, same direction, also after the ring has wrapped around."""
    from sqfa_amd._lbfgs import _History
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n + h)
    like = torch.zeros(n, dtype=dtype, device=dev)
    nat, ref = _History(h, like), None
    saved = _History.native
    try:
        _History.native = False
        ref = _History(h, like)
    finally:
        _History.native = saved
    assert nat._lib is not None and ref._lib is None
    B = torch.randn(n, 8, generator=gen, dtype=torch.float64)
    for it in range(pushes):
        s = torch.randn(n, generator=gen, dtype=torch.float64)
        y = s + 0.3 * (B @ (B.T @ s)) / n                  # y = (I + low rank PSD) s: s.y > 0
        s, y = s.to(dtype).to(dev), y.to(dtype).to(dev)
        nat.push(y, s)
        ref.push(y, s)
        if it in (0, 3, pushes - 1):
            g = torch.randn(n, generator=gen, dtype=torch.float64).to(dtype).to(dev)
            H = (s.dot(y) / y.dot(y))
            d_nat, d_ref = nat.direction(g, H), ref.direction(g, H)
            assert torch.linalg.norm(d_nat - d_ref) <= tol * torch.linalg.norm(d_ref)
    assert nat.slots == ref.slots
    idx = torch.as_tensor(nat.slots, device=dev)
    assert torch.allclose(nat.SY.index_select(0, idx).index_select(1, idx), ref.SY.index_select(0, idx).index_select(1, idx),
                          rtol=1e-10 if dtype == torch.float64 else 1e-4, atol=1e-12 if dtype == torch.float64 else 1e-3)
    assert torch.equal(nat.S, ref.S) and torch.equal(nat.Y, ref.Y)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 2e-6)])
@pytest.mark.parametrize("n", [5, 1000, 12544, 49152, 300001])
def test_native_step_stats_matches_torch(n, dtype, tol):
    """sqfa_lbfgs_step_stats: y = g - g_prev, s = t d (exact) and [max|g|, max|s|, y.s, y.y, y.s / y.y]."""
    from sqfa_amd._lbfgs import _History
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n)
    g, gp, d = (torch.randn(n, generator=gen, dtype=torch.float64).to(dev, dtype) for _ in range(3))
    hist = _History(10, g)
    y, s, scal = hist.step_stats(g, gp, d, 0.37)
    assert torch.equal(y, g - gp) and torch.equal(s, d * 0.37)
    y64, s64 = (g - gp).double(), (d * 0.37).double()
    expect = torch.stack([g.abs().max().double(), s.abs().max().double(), y64.dot(s64), y64.dot(y64), y64.dot(s64) / y64.dot(y64)])
    assert torch.allclose(scal.double(), expect, rtol=tol, atol=tol * float(y64.abs().max() * s64.abs().max()) * n ** 0.5)
    assert torch.equal(scal[:2], torch.stack([g.abs().max(), s.abs().max()]))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_native_step_stats_propagates_nan(dtype):
    """torch's abs().max() propagates NaN, fmax drops it: an all-NaN (or partly NaN) gradient must not read as
    max|g| = 0, which would end the LBFGS step as converged."""
    from sqfa_amd._lbfgs import _History
    dev = torch.device("cuda:0")
    for n, bad in ((5, [0, 1, 2, 3, 4]), (12544, [7000]), (300001, [299999])):
        g = torch.randn(n, dtype=dtype, device=dev)
        gp, d = torch.randn_like(g), torch.randn_like(g)
        g[bad] = float("nan")
        hist = _History(10, g)
        _y, _s, scal = hist.step_stats(g, gp, d, 0.5)
        assert torch.isnan(scal[0]), (n, scal)
        assert torch.isnan(g.abs().max())
