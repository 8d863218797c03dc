"""CPU restatement of the reference's TV-MCG metadata-corruption helpers
(dataprovider_pypots.py:149-229).  Test infrastructure only: nothing on the product path imports it.
The random draws are arguments (the reference takes them from torch's generator in the order noted)."""
import numpy as np


def add_continuous_metadata_noise_subset(cond, cont_idx, level, noise):
    """dataprovider_pypots.py:149-180.  cond [B,T,F] float32; noise [B,T,len(cont_idx)] = the reference's
    ``torch.randn_like(x_cont)``.  std is the unbiased standard deviation of each selected column over
    (B, T) (torch.std default), taken in float64 and rounded to float32 as torch's CPU kernel does."""
    cond = np.asarray(cond, np.float32)
    if level <= 0 or len(cont_idx) == 0:
        return cond
    x = cond[:, :, list(cont_idx)]
    std = x.astype(np.float64).std(axis=(0, 1), ddof=1, keepdims=True).astype(np.float32)
    std = std + np.float32(1e-8)
    out = cond.copy()
    out[:, :, list(cont_idx)] = x + np.asarray(noise, np.float32) * (np.float32(level) * std)
    return out


def flip_categorical_metadata_subset(cond, cat_idx, p, num_classes_dict, rand, base):
    """dataprovider_pypots.py:182-229.  rand[k], base[k] ([B,T] each) are the draws of the k-th column of
    cat_idx THAT IS NOT SKIPPED (num_classes > 1), in the reference's order: ``torch.rand_like`` then
    ``torch.randint(0, num_classes - 1)`` per column.  Codes are truncated to integers (``.long()``) whether
    flipped or not."""
    cond = np.asarray(cond, np.float32)
    if p <= 0 or len(cat_idx) == 0:
        return cond
    out = cond.copy()
    k = 0
    for col in cat_idx:
        if num_classes_dict[col] <= 1:
            continue
        x = np.trunc(cond[:, :, col]).astype(np.int64)
        flip = np.asarray(rand[k], np.float32) < np.float32(p)
        b = np.asarray(base[k], np.int64)
        new = b + (b >= x).astype(np.int64)
        out[:, :, col] = np.where(flip, new, x).astype(np.float32)
        k += 1
    return out
