"""convert_to_tensor (reference process_edited.py:252-301): decoder outputs -> data table tensor.

Oracle code: test infrastructure only (see ``oracle/__init__.py``).  float32 arithmetic in the
reference's order: sigmoid as 1 / (1 + exp(-x)); softmax as exp(x - max) / sum; argmax returns
the first maximal index (torch.argmax)."""
import numpy as np


def convert_to_tensor(gen_output, n_bins, n_cats, n_nums, cards):
    """gen_output: dict with 'bins' [B,T,n_bins], 'cats' list of [B,T,card_j], 'nums' [B,T,n_nums]
    (keys present when that type has columns) -> [B,T,n_bins+n_cats+n_nums] float32."""
    parts = []
    if n_bins:
        x = np.asarray(gen_output["bins"], np.float32)
        s = np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))
        parts.append((s > np.float32(0.5)).astype(np.float32))
    if len(cards):
        cols = []
        for j in range(len(cards)):
            x = np.asarray(gen_output["cats"][j], np.float32)
            e = np.exp(x - x.max(-1, keepdims=True), dtype=np.float32)
            p = e / e.sum(-1, keepdims=True, dtype=np.float32)
            cols.append(np.argmax(p, -1).astype(np.float32))
        parts.append(np.stack(cols, -1))
    if n_nums:
        parts.append(np.asarray(gen_output["nums"], np.float32))
    return np.concatenate(parts, -1)
