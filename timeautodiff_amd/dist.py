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

    def _through_host(self, t):
        """gloo with device tensors -- rehearsals of the multi-rank logic with several processes on ONE GPU; the product's
        backend is RCCL -- was seen to hang now and then with eight processes on one card (in all_reduce and in
        all_gather alike): there the collective runs on a host copy, ordered after the current stream's work."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def all_reduce(self, *tensors):
        """Sum the given tensors over the ranks (no-op for a single process)."""
        if self.world == 1:
            return
        for t in tensors:
            if self._through_host(t):
                h = t.cpu()                      # (waits for the current stream's work on t)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    class _Done:
        def wait(self):
            return True

    def all_reduce_async(self, t):
        """Start summing ``t`` over the ranks; returns a handle whose wait() orders the current stream after the
        collective (None for a single process).  The collective is ordered after the work enqueued so far on the
        stream that is current at the call."""
        if self.world == 1:
            return None
        if self._through_host(t):
            self.all_reduce(t)
            return self._Done()
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
            host = torch.empty(allbuf.shape, dtype=t.dtype)            # see _through_host
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            allbuf.copy_(host)
        else:
            dist.all_gather_into_tensor(allbuf, mine, group=self.group)
        if all(hi - lo == longest for lo, hi in bounds):
            return allbuf
        return torch.cat([allbuf[r * longest:r * longest + (hi - lo)] for r, (lo, hi) in enumerate(bounds)], 0)
