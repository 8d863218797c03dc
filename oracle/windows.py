"""CPU restatement of the reference's ``sliding_window`` (dataprovider_pypots.py:75-87) and of the minibatch
index gather the trainers apply to its result (DIFF.py:320, VAE.py:340-348).  Test infrastructure only."""
import numpy as np


def sliding_window(ori_data, seq_len, stride):
    """Loop of dataprovider_pypots.py:79-85: windows start at i = 0, stride, ... while i <= len - seq_len."""
    ori_data = np.asarray(ori_data)
    out, i = [], 0
    while i <= len(ori_data) - seq_len:
        out.append(ori_data[i:i + seq_len])
        i += stride
    return np.stack(out)


def gather_windows(table, idx, seq_len, stride):
    """stacked[idx] without building the stacked tensor."""
    table = np.asarray(table)
    return np.stack([table[int(i) * stride:int(i) * stride + seq_len] for i in idx]) if len(idx) else np.zeros((0, seq_len) + table.shape[1:], table.dtype)
