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


def test_discarded_speculative_evaluation_does_not_raise(monkeypatch):
    """the evaluation made past torch's `g.d > -tolerance_change` stop is discarded -- torch.optim.LBFGS and the
    reference never evaluate that point -- so non-finite flags coming from it must be discarded with it, not raised.
    Every point torch.optim.LBFGS evaluates is recorded first; CompactLBFGS' closure then reports a NaN distance at
    every OTHER point (= exactly the speculative evaluations)."""
    monkeypatch.setattr(CompactLBFGS, "fuse_readback", True)
    A = torch.diag(torch.linspace(0.5, 3.0, 12, dtype=torch.float64))
    x0 = torch.linspace(-1, 1, 12, dtype=torch.float64)

    def run(cls, known=None):
        x = torch.nn.Parameter(x0.clone())
        opt = cls([x], lr=1.0, history_size=8)
        points, extra = [], [0]

        def closure():
            opt.zero_grad()
            points.append(x.detach().clone())
            loss = 0.5 * x @ A @ x
            loss.backward()
            return loss

        if known is not None:
            def deferred():
                loss = closure().detach()
                here = points[-1]
                seen_by_torch = any(torch.linalg.norm(here - p) <= 1e-9 * (1 + torch.linalg.norm(p)) for p in known)
                extra[0] += 0 if seen_by_torch else 1
                return torch.cat([loss.reshape(1), torch.tensor([0.0 if seen_by_torch else 1.0, 0.0], dtype=torch.float64)])

            def check_flags(n_nan, n_inf):
                if n_nan:
                    raise ValueError("nan in distances")
            closure.deferred, closure.check_flags = deferred, check_flags
        for _ in range(6):      # converges inside the second step; torch stops before each later step
            opt.step(closure)
        return x.detach().clone(), points, extra[0]

    x_torch, pts, _ = run(torch.optim.LBFGS)
    x_compact, _, extra = run(CompactLBFGS, known=pts)      # must not raise
    assert extra >= 1, "the speculative evaluation never happened: the test does not exercise the break"
    assert torch.linalg.norm(x_compact - x_torch) <= 1e-12
