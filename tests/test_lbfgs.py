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
