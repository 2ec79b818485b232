"""Multi-GPU evaluation of the pairwise loss: class-pair tiles are sharded over ranks
(one process per GPU) and the partial loss / validity flags / gradient are summed with ONE
all-reduce of a fused buffer per closure (RCCL over xGMI: backend "nccl" on ROCm).

The feature scatters S (C,m,m) are tiny (<= 4.4 MB) and replicated: every rank evaluates the
tiles (bi,bj) with (bi+bj) % world == rank of the same S, so no data-path exchange is needed
before the kernel; after the all-reduce every rank holds the identical loss and dL/dS and
performs the identical LBFGS update.
"""
import torch
import torch.distributed as dist

__all__ = ["PairShard"]


class PairShard:
    """Sharding policy + reducer for ``_native.PairwiseLoss``.

    >>> model.pair_shard = PairShard()            # default process group
    """

    def __init__(self, group=None, rank=None, world_size=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size

    @property
    def shard(self):
        return (self.rank, self.world_size)

    def reduce(self, loss, flags, grad):
        """Sum (loss, flags, grad) over the ranks with a single all-reduce."""
        if self.world_size == 1:
            return loss, flags, grad
        n = grad.numel()
        buf = torch.empty(n + 3, dtype=grad.dtype, device=grad.device)
        buf[0] = loss
        buf[1:3] = flags.to(grad.dtype)
        buf[3:] = grad.reshape(-1)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        flags_out = buf[1:3].round().to(torch.int32)
        return buf[0].clone(), flags_out, buf[3:].reshape(grad.shape).clone()
