"""Data-parallel plan: one process per GPU, torch.distributed ('nccl' = RCCL over xGMI).

Training: every rank draws the same global batch (same seeds), takes a contiguous
slice of it, computes gradients already scaled by the GLOBAL batch denominator, and
one all-reduce(sum) over the flat fp32 gradient buffer makes every replica apply the
same Adam step.  Sampling: the batch is sliced the same way and needs no collective.
"""
import torch
import torch.distributed as dist


def slice_bounds(n, world, rank):
    """Contiguous slice [lo, hi) of n items for ``rank``; sizes differ by at most one."""
    return (n * rank) // world, (n * (rank + 1)) // world


class DataParallelPlan:
    def __init__(self, global_batch, process_group=None):
        self.group = process_group
        if process_group is None:
            self.world, self.rank = 1, 0
        else:
            if process_group is True:           # "the default group"
                self.group = dist.group.WORLD
            self.world = dist.get_world_size(self.group)
            self.rank = dist.get_rank(self.group)
        self.global_batch = global_batch
        self.lo, self.hi = slice_bounds(global_batch, self.world, self.rank)
        self.local_count = self.hi - self.lo

    def all_reduce(self, *tensors):
        """Sum the given tensors over the ranks (no-op for a single process)."""
        if self.world == 1:
            return
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather_cat(self, t, dim=0):
        """Concatenate per-rank slices (possibly of unequal length) along ``dim`` on every rank."""
        if self.world == 1:
            return t
        sizes = [slice_bounds(self.global_batch, self.world, r) for r in range(self.world)]
        outs = []
        for r, (lo, hi) in enumerate(sizes):
            shape = list(t.shape)
            shape[dim] = hi - lo
            buf = t.contiguous() if r == self.rank else torch.empty(shape, dtype=t.dtype, device=t.device)
            dist.broadcast(buf, src=dist.get_global_rank(self.group, r) if self.group is not dist.group.WORLD else r,
                           group=self.group)
            outs.append(buf)
        return torch.cat(outs, dim=dim)
