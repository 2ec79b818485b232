"""LBFGS driver for the SQFA models (reference: src/sqfa/_optim.py).

torch.optim.LBFGS is used unmodified, so epoch semantics are the reference's by
construction (the loss recorded for an epoch is that of the first closure call of the
step).  What changes is the closure body: when the model's ``distance_fun``
is one of the native affine-invariant operators the closure issues ONE fused
loss+gradient launch (no (C,C) matrix, no tril gather, validity flag read with the loss)
instead of the reference's distance-matrix -> guard -> gather -> mean chain.
"""
import time

import torch
from tqdm import tqdm

__all__ = ["fitting_loop"]


def __dir__():
    return __all__


_NAN_MSG = "Some distances between classes are NaN. Try using float64 or a different regularization parameter."
_INF_MSG = "Some distances between classes are inf. Try using float64 or a different regularization parameter."


def check_distances_valid(distances):
    """Raise ValueError when the distance matrix holds NaN or inf.  Like the reference
    (src/sqfa/_optim.py:16-30) the whole matrix is inspected."""
    if torch.isnan(distances).any():
        raise ValueError(_NAN_MSG)
    if torch.isinf(distances).any():
        raise ValueError(_INF_MSG)


def raise_on_flags(flags):
    """flags: int tensor {nan_count, inf_count} produced by the fused kernel."""
    n_nan, n_inf = (int(v) for v in flags.tolist())
    if n_nan:
        raise ValueError(_NAN_MSG)
    if n_inf:
        raise ValueError(_INF_MSG)


def _n_classes(data_statistics):
    if isinstance(data_statistics, dict):
        return data_statistics["means"].shape[0]
    return data_statistics.shape[0]


def fitting_loop(model, data_statistics, max_epochs=200, lr=0.1, atol=1e-6, show_progress=True,
                 return_loss=False, **kwargs):
    """Learn the filters with LBFGS.  Same arguments, stopping rule (|dloss| < atol for three
    consecutive epochs), messages and return value as the reference's fitting_loop
    (src/sqfa/_optim.py:33-145); extra keyword arguments go to torch.optim.LBFGS."""
    optimizer = torch.optim.LBFGS(model.parameters(), lr=lr, **kwargs)
    n_classes = _n_classes(data_statistics)
    if n_classes < 2:
        raise ValueError("At least two classes are needed to fit the filters.")
    prepared = model._prepare_statistics(data_statistics)
    rows, cols = torch.tril_indices(n_classes, n_classes, offset=-1)

    def closure():
        optimizer.zero_grad()
        fused = model._fused_closure_loss(prepared)
        if fused is not None:
            loss, flags = fused
            raise_on_flags(flags)
        else:
            distances = model.get_class_distances(prepared, regularized=True)
            check_distances_valid(distances)
            loss = -distances[rows.to(distances.device), cols.to(distances.device)].mean()
        loss.backward()
        return loss

    losses, times = [], []
    start = time.time()
    previous = 0.0
    streak = 0
    for epoch in tqdm(range(max_epochs), desc="Epochs", unit="epoch", disable=not show_progress):
        value = optimizer.step(closure).item()
        times.append(time.time() - start)
        losses.append(value)
        streak = streak + 1 if abs(previous - value) < atol else 0
        previous = value
        if streak >= 3:
            tqdm.write(
                f"Loss change below {atol} for 3 consecutive epochs. "
                f"Stopping training at epoch {epoch + 1}/{max_epochs}."
            )
            break
    else:
        print(
            f"Reached max_epochs ({max_epochs}) without meeting stopping criteria."
            + "Consider increasing max_epochs, changing initialization or using dtype=torch.float64."
        )
    if return_loss:
        return torch.tensor(losses), torch.tensor(times)
    return None
