"""Shared helpers for golden fixtures: synthetic inputs, schemas, metadata.

Used by tests/golden/make_fixtures.py (build container, with the reference)
and by the tests (anywhere, without it).
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCHEMAS = {
    # all-numeric (stock_data-like)
    "numeric": dict(n_bins=0, n_cats=0, n_nums=6, cards=[]),
    # one binary, two categoricals, three numericals
    "mixed": dict(n_bins=1, n_cats=2, n_nums=3, cards=[3, 5]),
    # Metro_Traffic-like: three categoricals (one of cardinality 1), five numericals
    "metro": dict(n_bins=0, n_cats=3, n_nums=5, cards=[12, 1, 11]),
}


def pack_meta(d):
    return np.frombuffer(json.dumps(d).encode(), dtype=np.uint8).copy()


def unpack_meta(a):
    return json.loads(bytes(a.tolist()).decode())


def make_data(schema, N, T, K, seed, mask_kind="ones"):
    """Seeded synthetic windows of the reference's input contract.

    Returns float32 arrays: data [N,T,F] (codes as floats in the discrete
    columns, numericals in [0,1)), mask [N,T,F] in {0,1}, response = data*mask,
    cond = data*(1-mask), time_info [N,T,8] (sin/cos pairs), latent [N,T,K]."""
    rng = np.random.default_rng(seed)
    nb, nc, nn_ = schema["n_bins"], schema["n_cats"], schema["n_nums"]
    cols = []
    for _ in range(nb):
        cols.append(rng.integers(0, 2, size=(N, T)).astype(np.float32))
    for c in schema["cards"][:nc]:
        cols.append(rng.integers(0, c, size=(N, T)).astype(np.float32))
    for _ in range(nn_):
        cols.append(rng.random((N, T), dtype=np.float32))
    data = np.stack(cols, -1).astype(np.float32)
    F = data.shape[-1]
    if mask_kind == "ones":
        mask = np.ones((N, T, F), np.float32)
    elif mask_kind == "bern":
        mask = (rng.random((N, T, F)) < 0.1).astype(np.float32)
    elif mask_kind == "forecast":
        mask = np.zeros((N, T, F), np.float32)
        mask[:, T // 2:, :] = 1.0
    else:
        raise ValueError(mask_kind)
    ang = rng.random((N, T, 4), dtype=np.float32) * np.float32(2 * np.pi)
    time_info = np.concatenate([np.sin(ang), np.cos(ang)], -1).astype(np.float32)
    latent = rng.standard_normal((N, T, K)).astype(np.float32)
    return dict(data=data, mask=mask, response=data * mask, cond=data * (1 - mask),
                time_info=time_info, latent=latent)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = unpack_meta(z["meta"]) if "meta" in z.files else {}
    return z, meta


def sub(z, prefix):
    """dict of entries under 'prefix/' with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: z[k] for k in z.files if k.startswith(prefix + "/")}


def rel_err(a, b):
    """max|a-b| / max(max|b|, tiny): relative error at the scale of the reference tensor."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30)) if b.size else 0.0


def stats(a):
    a64 = np.asarray(a).astype(np.float64).ravel()
    return np.array([a64.sum(), np.sqrt((a64 ** 2).sum()), a64[:8].sum() if a64.size else 0.0])
