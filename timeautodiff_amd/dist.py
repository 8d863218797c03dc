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

    def all_reduce_async(self, t):
        """Start summing ``t`` over the ranks; returns a handle whose wait() orders the current stream after the
        collective (None for a single process).  The collective is ordered after the work enqueued so far on the
        stream that is current at the call."""
        if self.world == 1:
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_gather_cat(self, t):
        """Concatenate the per-rank slices (lengths differ by at most one) along dim 0 on every rank: ONE
        all_gather_into_tensor over slices padded to the longest one, then the padding rows are dropped."""
        if self.world == 1:
            return t
        bounds = [slice_bounds(self.global_batch, self.world, r) for r in range(self.world)]
        longest = max(hi - lo for lo, hi in bounds)
        tail = tuple(t.shape[1:])
        mine = torch.zeros((longest,) + tail, dtype=t.dtype, device=t.device)
        mine[:t.shape[0]].copy_(t)
        allbuf = torch.empty((self.world * longest,) + tail, dtype=t.dtype, device=t.device)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no single-buffer all-gather for device tensors (rehearsals on one GPU): per-rank views of the same buffer
            dist.all_gather(list(allbuf.view((self.world, longest) + tail).unbind(0)), mine, group=self.group)
        else:
            dist.all_gather_into_tensor(allbuf, mine, group=self.group)
        if all(hi - lo == longest for lo, hi in bounds):
            return allbuf
        return torch.cat([allbuf[r * longest:r * longest + (hi - lo)] for r, (lo, hi) in enumerate(bounds)], 0)
