"""CompactLBFGS (vectorised two-loop recursion) against torch.optim.LBFGS: same iterates,
same number of closure calls, same state transitions, also when the history wraps around."""
import pytest
import torch

from sqfa_amd._lbfgs import CompactLBFGS


def rosenbrock_like(x):
    return (100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2).sum() * 1e-3 + 0.05 * (x ** 4).sum()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
@pytest.mark.parametrize("history_size", [100, 5])
@pytest.mark.parametrize("fuse_readback", [False, True])
def test_matches_torch_lbfgs(dtype, tol, history_size, fuse_readback, monkeypatch):
    # fuse_readback=True: the path GPU parameters take (decision scalars gathered in one copy,
    # s.y of the next iteration formed ahead), forced here on CPU tensors
    monkeypatch.setattr(CompactLBFGS, "fuse_readback", fuse_readback)
    torch.manual_seed(0)
    x0 = torch.randn(40, dtype=dtype) * 0.5
    runs = []
    for cls in (torch.optim.LBFGS, CompactLBFGS):
        x = torch.nn.Parameter(x0.clone())
        opt = cls([x], lr=0.1, history_size=history_size)
        calls = [0]

        def closure():
            opt.zero_grad()
            calls[0] += 1
            loss = rosenbrock_like(x)
            loss.backward()
            return loss

        # float32 trajectories of a fixed-step LBFGS separate after a few dozen iterations however the
        # direction is summed; compare them over a shorter horizon
        losses = [opt.step(closure).item() for _ in range(12 if dtype == torch.float64 else 2)]
        runs.append((x.detach().clone(), losses, calls[0], opt.state[x]["n_iter"], opt.state[x]["func_evals"]))
    (xa, la, ca, na, fa), (xb, lb, cb, nb, fb) = runs
    assert (ca, na, fa) == (cb, nb, fb)
    assert torch.allclose(torch.tensor(la), torch.tensor(lb), rtol=tol, atol=tol)
    assert torch.linalg.norm(xa - xb) <= tol * torch.linalg.norm(xa)


@pytest.mark.parametrize("deferred", [False, True])
def test_speculative_descent_test_rolls_back_at_convergence(deferred, monkeypatch):
    """Fused read-back path: torch's test g.d > -tolerance_change (stop BEFORE the step) is evaluated after the
    step and its closure; when it fires the parameters must be restored and the extra evaluation must not be
    counted -- iterates, n_iter and func_evals equal torch.optim.LBFGS' through convergence and beyond.  With
    `deferred` the closure hands [loss, nan, inf] over as a tensor and its flags are checked by the optimizer."""
    monkeypatch.setattr(CompactLBFGS, "fuse_readback", True)
    A = torch.diag(torch.linspace(0.5, 3.0, 12, dtype=torch.float64))
    x0 = torch.linspace(-1, 1, 12, dtype=torch.float64)
    runs = []
    for cls in (torch.optim.LBFGS, CompactLBFGS):
        x = torch.nn.Parameter(x0.clone())
        opt = cls([x], lr=1.0, history_size=8)
        calls, checked = [0], [0]

        def closure():
            opt.zero_grad()
            calls[0] += 1
            loss = 0.5 * x @ A @ x
            loss.backward()
            return loss

        if deferred and cls is CompactLBFGS:
            closure.deferred = lambda: torch.cat([closure().detach().reshape(1), torch.zeros(2, dtype=torch.float64)])

            def check_flags(n_nan, n_inf):
                checked[0] += 1
                assert n_nan == 0 and n_inf == 0
            closure.check_flags = check_flags
        losses = [opt.step(closure).item() for _ in range(6)]  # converges inside the second step; later steps stop at once
        runs.append((x.detach().clone(), losses, calls[0], opt.state[x]["n_iter"], opt.state[x]["func_evals"], checked[0]))
    (xa, la, ca, na, fa, _), (xb, lb, cb, nb, fb, checked) = runs
    assert (na, fa) == (nb, fb)
    assert cb >= ca                       # the discarded evaluations are extra closure calls, never counted ones
    assert torch.allclose(torch.tensor(la), torch.tensor(lb), rtol=1e-12, atol=1e-14)
    assert torch.linalg.norm(xa - xb) <= 1e-12
    assert (checked > 0) == deferred


def test_deferred_closure_flags_raise(monkeypatch):
    monkeypatch.setattr(CompactLBFGS, "fuse_readback", True)
    x = torch.nn.Parameter(torch.ones(5, dtype=torch.float64))
    opt = CompactLBFGS([x], lr=0.1)
    n = [0]

    def closure():
        opt.zero_grad()
        loss = (x ** 2).sum()
        loss.backward()
        return loss

    def deferred():
        n[0] += 1
        return torch.cat([closure().detach().reshape(1), torch.tensor([1.0 if n[0] >= 2 else 0.0, 0.0], dtype=torch.float64)])

    def check_flags(n_nan, n_inf):
        if n_nan:
            raise ValueError("nan in distances")
    closure.deferred, closure.check_flags = deferred, check_flags
    with pytest.raises(ValueError, match="nan in distances"):
        opt.step(closure)


def test_line_search_defers_to_torch():
    x = torch.nn.Parameter(torch.tensor([1.5, -0.5], dtype=torch.float64))
    opt = CompactLBFGS([x], lr=1.0, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = rosenbrock_like(x)
        loss.backward()
        return loss

    l0 = opt.step(closure).item()
    l1 = opt.step(closure).item()
    assert l1 < l0


@pytest.mark.gpu
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


@pytest.mark.gpu
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
