"""Device-side sliding windows (reference dataprovider_pypots.py:75-87), the window gather of the trainers, and the
TV-MCG metadata-corruption helpers (reference dataprovider_pypots.py:149-229).

The reference cuts a ``[rows, F]`` table into overlapping windows with a Python loop and ``torch.stack``
(``sliding_window``), keeps the stacked ``[N, T, F]`` tensor on the host, and every training iteration copies
it to the device and indexes it with a list of window numbers (DIFF.py:308-320, VAE.py:326-348).  At stride 1
the stacked tensor is T times the table.  Here the table goes to the GPU once; a minibatch of windows is
cut straight from it (``tad_gather_windows``: ``out[b, t, :] = table[idx[b] * stride + t, :]``).

``sliding_window`` keeps the reference's signature and result (the stacked tensor, built by one launch);
``WindowedTable`` is the un-materialised form that ``DIFF.DiffusionTrainer`` accepts for ``time_info`` / ``cond`` and
``VAE.VAETrainer`` for ``response_train`` / ``cond_train`` in place of an ``[N, T, F]`` tensor (masks stay stacked;
validation sets are materialised once).
"""
import torch

from . import ops


def num_windows(rows, seq_len, stride):
    """Windows the reference's loop produces: i = 0, stride, 2 stride ... while i <= rows - seq_len."""
    if seq_len <= 0 or stride <= 0:
        raise ValueError("seq_len and stride must be positive")
    return 0 if rows < seq_len else (rows - seq_len) // stride + 1


class WindowedTable:
    """Windows of a device-resident ``[rows, F]`` float32 table: window i = rows i*stride .. i*stride+T-1.
    Behaves like the reference's stacked ``[N, T, F]`` tensor where the trainers need it (``shape``, ``len``,
    ``gather``), without holding it."""

    def __init__(self, table, seq_len, stride=1, device=None):
        if table.dim() != 2:
            raise ValueError("WindowedTable needs a [rows, features] table")
        dev = torch.device(device) if device is not None else (table.device if table.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        if dev.type != "cuda":
            raise ops._lib.TadError("WindowedTable needs a GPU (no CPU path exists)")
        self.table = table.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.T, self.stride = int(seq_len), int(stride)
        self.N = num_windows(self.table.shape[0], self.T, self.stride)
        if self.N == 0:
            raise RuntimeError("stack expects a non-empty TensorList")      # what torch.stack([]) raises in the reference

    @property
    def shape(self):
        return torch.Size((self.N, self.T, self.table.shape[1]))

    @property
    def device(self):
        return self.table.device

    def __len__(self):
        return self.N

    def gather(self, idx, out=None):
        """Windows ``idx`` (int64 tensor on the device) -> ``[len(idx), T, F]``."""
        idx = idx.to(device=self.table.device, dtype=torch.int64).contiguous()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.N):
            raise IndexError("window index out of range")
        return self.gather_unchecked(idx, out)

    def gather_unchecked(self, idx, out=None):
        """As ``gather`` for indices already known to lie in [0, N) (no host sync)."""
        if out is None:
            out = torch.empty(idx.numel(), self.T, self.table.shape[1], dtype=torch.float32, device=self.table.device)
        ops.gather_windows(self.table, idx, self.T, self.stride, out)
        return out

    def materialize(self):
        """The reference's stacked tensor ``[N, T, F]``."""
        return self.gather_unchecked(torch.arange(self.N, device=self.table.device, dtype=torch.int64))


def sliding_window(ori_data, seq_len, stride):
    """reference dataprovider_pypots.py:75-87: ``[rows, F]`` (or ``[rows]``) -> stacked windows ``[N, seq_len, F]``,
    on the GPU.  A table on the host is copied to the current device first."""
    t = torch.as_tensor(ori_data)
    one_d = t.dim() == 1
    t2 = t.reshape(t.shape[0], -1)
    out = WindowedTable(t2, seq_len, stride).materialize()
    out = out.reshape(out.shape[0], seq_len) if one_d else out.reshape(out.shape[0], seq_len, *t.shape[1:])
    return out.to(t.dtype) if t.dtype.is_floating_point and t.dtype != torch.float32 else out


# ---------------------------------------------------------------------------- TV-MCG metadata corruption
def _on_device(cond):
    if not torch.cuda.is_available():
        raise ops._lib.TadError("libtad_hip ops need a GPU (no CPU path exists)")
    return cond if cond.is_cuda else cond.to("cuda")


def add_continuous_metadata_noise_subset(cond, cont_idx, level, *, noise=None, generator=None):
    """reference dataprovider_pypots.py:149-180: Gaussian noise of ``level`` column standard deviations on the
    continuous metadata columns ``cont_idx`` of cond [B, T, F]; same signature and result (a new tensor on the GPU;
    cond itself when level <= 0 or no column is listed, like the reference).

    Keyword-only extras: noise [B, T, len(cont_idx)] = the standard-normal draws to use (the reference's
    ``randn_like(x_cont)``); generator = the device generator to draw them from otherwise."""
    if level <= 0 or len(cont_idx) == 0:
        return cond
    c = _on_device(cond)
    if c.dtype != torch.float32:
        raise TypeError("cond must be float32 (the reference's cond tensors are)")
    B, T, F = c.shape
    n = len(cont_idx)
    if noise is None:
        noise = torch.randn(B, T, n, device=c.device, generator=generator)
    noise = noise.to(device=c.device, dtype=torch.float32).reshape(B * T, n).contiguous()
    c = c.contiguous()
    out = torch.empty_like(c)
    ops.metadata_noise(c.view(B * T, F), [int(i) for i in cont_idx], noise, level, out.view(B * T, F))
    return out


def flip_categorical_metadata_subset(cond, cat_idx, p, num_classes_dict, *, rand=None, base=None, generator=None):
    """reference dataprovider_pypots.py:182-229: with probability p, the code in each categorical metadata column of
    ``cat_idx`` is replaced by a uniformly drawn OTHER class; columns with fewer than two classes are skipped.  Same
    signature and result (codes come back truncated to integers, flipped or not, like the reference's ``.long()``).

    Keyword-only extras: rand / base = per flipped column (in ``cat_idx`` order, skipped columns left out) the
    [B, T] uniform draws and the [B, T] integer draws from 0 .. classes-2, i.e. the reference's ``rand_like`` and
    ``randint`` results; generator = the device generator to draw them from otherwise."""
    if p <= 0 or len(cat_idx) == 0:
        return cond
    c = _on_device(cond)
    if c.dtype != torch.float32:
        raise TypeError("cond must be float32 (the reference's cond tensors are)")
    B, T, F = c.shape
    cols = [int(col) for col in cat_idx if num_classes_dict[col] > 1]       # KeyError for an unknown column, as the reference
    classes = [int(num_classes_dict[col]) for col in cols]
    n = len(cols)
    dev = c.device
    if rand is None or base is None:
        rand, base = [], []
        for k in classes:                                                   # the reference's order: rand, randint per column
            rand.append(torch.rand(B, T, device=dev, generator=generator))
            base.append(torch.randint(0, k - 1, (B, T), device=dev, generator=generator))
    as_block = lambda seq, dt: (torch.stack([torch.as_tensor(a).to(device=dev, dtype=dt).reshape(B * T) for a in seq])  # noqa: E731
                                if n else torch.empty(0, B * T, device=dev, dtype=dt)).contiguous()
    c = c.contiguous()
    out = torch.empty_like(c)
    ops.flip_categories(c.view(B * T, F), cols, classes, as_block(rand, torch.float32), as_block(base, torch.int64), p, out.view(B * T, F))
    return out
