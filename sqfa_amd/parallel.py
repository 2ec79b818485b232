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

__all__ = ["PairShard", "ClassShard", "replicas_agree"]


def replicas_agree(tensors, group=None):
    """True when `tensors` are BITWISE identical on every rank of `group`.  Every rank of a sharded fit repeats the
    same LBFGS update on the all-reduced gradient, which is only sound while the all-reduce hands every rank the
    same bits (true for RCCL's and gloo's ring / tree reductions: one rank finishes each element and the result is
    distributed).  This is the check that would notice if that ever stopped holding: an exact integer checksum
    of the bit patterns, compared with one MAX all-reduce of [h, -h] (= max and -min of the checksums)."""
    h = None
    views = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
    for t in tensors:
        view = views.get(t.element_size())
        if view is None or t.is_complex():
            import warnings
            warnings.warn(f"sqfa_amd.replicas_agree: skipping a {t.dtype} tensor (no integer view of its bit pattern)")
            continue
        bits = t.detach().contiguous().view(view)
        # position-weighted so that permuted or compensating differences do not cancel
        w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device)
        part = (bits.reshape(-1).to(torch.int64) * w).sum()
        h = part if h is None else h * 1000003 + part
    if h is None:
        return True
    # (int64 wraps around on overflow, identically on every rank: the checksum stays an exact function of the bits)
    both = torch.stack([h, -h])
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    return bool((both[0] == -both[1]).item())


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

    def reduce_fused(self, buf, flags, shape):
        """Same as `reduce` for a buffer [loss, -, -, grad...] the kernel has already filled."""
        buf[1:3].copy_(flags)  # int32 -> real in the copy itself; small counts are exact in float32
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return buf[0], buf[1:3].to(torch.int32), buf[3:].view(shape)

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


class _GatherClasses(torch.autograd.Function):
    """all_gather of the per-rank slices of the feature scatters along the class axis.  In the
    backward pass every rank already holds the FULL dL/dS (PairShard.reduce summed it), so the
    gradient of the local slice is just the matching rows: no second collective."""

    @staticmethod
    def forward(ctx, S_local, shard):
        ctx.shard = shard
        return shard.gather_tensors(S_local)

    @staticmethod
    def backward(ctx, g_full):
        sh = ctx.shard
        return g_full[sh.offset:sh.offset + sh.counts[sh.rank]], None


class ClassShard:
    """Class-sharding of the PROJECTION for large D: rank r holds the
    (C_r, D, D) statistics of its own classes only, projects them (C_r, m, m), and the small
    feature scatters are all-gathered so that every rank can evaluate its tile shard of ALL
    pairs.  After the backward pass the partial filter gradients (sums over local classes)
    are all-reduced.  Use together with PairShard:

    >>> model.pair_shard = PairShard(); model.class_shard = ClassShard(n_local_classes)
    >>> model.fit(data_statistics=local_statistics)
    """

    def __init__(self, n_local, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        counts = [None] * self.world_size
        dist.all_gather_object(counts, int(n_local), group=group)
        self.counts = counts
        self.offset = sum(counts[: self.rank])
        self.n_classes = sum(counts)

    def gather_tensors(self, S_local, recv=None):
        """All ranks' class slices concatenated along dim 0.  Shards may be uneven: every rank sends a
        slice padded to the largest shard (all_gather with equal sizes works on RCCL and gloo alike)."""
        n_max = max(self.counts)
        send = S_local.contiguous()
        if send.shape[0] != n_max:
            padded = send.new_zeros((n_max,) + tuple(send.shape[1:]))
            padded[: send.shape[0]] = send
            send = padded
        if recv is None:
            recv = send.new_empty((self.world_size,) + tuple(send.shape))
        dist.all_gather([recv[r] for r in range(self.world_size)], send, group=self.group)
        if all(n == n_max for n in self.counts):
            return recv.reshape((self.n_classes,) + tuple(send.shape[1:]))
        return torch.cat([recv[r, :n] for r, n in enumerate(self.counts)], dim=0)

    def gather(self, S_local):
        if S_local.shape[0] != self.counts[self.rank]:
            raise ValueError("local statistics do not match the class count announced to ClassShard")
        return _GatherClasses.apply(S_local, self)

    def reduce_gradients(self, parameters):
        for prm in parameters:
            if prm.grad is not None:
                dist.all_reduce(prm.grad, op=dist.ReduceOp.SUM, group=self.group)
