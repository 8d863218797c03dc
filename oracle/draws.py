"""TEST INFRASTRUCTURE ONLY -- numpy restatement of tad_draw_batch (the trainers' draws="device" mode).

The reference draws an iteration's randomness on the CPU (DIFF.py:306 random.sample, :317 torch.rand(...).sort,
:253 torch.randint, :108 torch.randn_like); those generator streams cannot be replayed on the device, so the device
mode has a definition of its own, pinned here: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as
easy as 1, 2, 3", SC'11; checked below against the known-answer vectors of the Random123 distribution), keyed by the
seed, counter words (element lo, element hi, stream, iteration).  Integer outputs (indices, diffusion steps) and the
sorted uniforms are bit-exact against the kernel; the normals go through log / sin / cos and agree to rounding."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
ST_PERM, ST_T, ST_I, ST_NOISE, ST_SAMPLER = 0, 1, 2, 3, 4


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint32 arrays of one shape); returns four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def _draw(seed, counter, stream, e):
    e = np.asarray(e, dtype=np.uint64)
    return philox4x32_10(e & MASK, e >> np.uint64(32), np.uint64(stream), np.uint64(counter), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def _mix32(h):
    h = np.asarray(h, dtype=np.uint64) & MASK
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x85EBCA77)) & MASK
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE3D)) & MASK
    h ^= h >> np.uint64(16)
    return h


def perm_indices(seed, counter, N, B):
    """B distinct indices in [0, N): four keyed Feistel rounds on the smallest even number of bits that covers N,
    cycle-walked into range."""
    bits = 2
    while bits < 62 and (1 << bits) < N:
        bits += 1
    hb = (bits + 1) // 2
    mask = np.uint64((1 << hb) - 1)
    keys = [int(k[()]) for k in _draw(seed, counter, ST_PERM, 0)]
    x = np.arange(B, dtype=np.uint64)
    out = np.empty(B, dtype=np.int64)
    todo = np.ones(B, dtype=bool)
    first = True
    while todo.any():
        cur = x[todo]
        L, R = cur >> np.uint64(hb), cur & mask
        for r in range(4):
            F = _mix32(((R * np.uint64(0x9E3779B1)) & MASK) + np.uint64(keys[r])) & mask
            L, R = R, L ^ F
        cur = (L << np.uint64(hb)) | R
        x[todo] = cur
        done = cur < np.uint64(N)
        ids = np.flatnonzero(todo)
        out[ids[done]] = cur[done].astype(np.int64)
        todo[ids[done]] = False
        first = False
    return out


def u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def u01_open(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)


def draw_batch(seed, counter, N, B, T, K, n_steps):
    """-> dict(idx [B] int64, t [B, T] float32 ascending, i [B] int32, noise [B, T, K] float32)."""
    idx = perm_indices(seed, counter, N, B)
    i = (_draw(seed, counter, ST_I, np.arange(B))[0] % np.uint32(n_steps)).astype(np.int32)
    t = u01(_draw(seed, counter, ST_T, np.arange(B * T))[0]).reshape(B, T)
    t = np.sort(t, axis=1, kind="stable")
    n = T * K
    pairs = (n + 1) // 2
    d = _draw(seed, counter, ST_NOISE, np.arange(B * pairs))
    r = np.sqrt(np.float32(-2.0) * np.log(u01_open(d[0]), dtype=np.float32), dtype=np.float32).reshape(B, pairs)
    th = (np.float32(6.283185307179586) * u01(d[1])).reshape(B, pairs)
    z = np.empty((B, 2 * pairs), dtype=np.float32)
    z[:, 0::2] = r * np.cos(th, dtype=np.float32)
    z[:, 1::2] = r * np.sin(th, dtype=np.float32)
    return dict(idx=idx, t=t, i=i, noise=z[:, :n].reshape(B, T, K))


def sampler_noise(seed, counter, row0, rows, K):
    """-> [rows, K] float32: the sampler's noise of global rows row0 .. row0+rows at step ``counter``."""
    P = (K + 1) // 2
    r = np.arange(rows, dtype=np.uint64)[:, None]
    p = np.arange(P, dtype=np.uint64)[None, :]
    d = _draw(seed, counter, ST_SAMPLER, (np.uint64(row0) + r) * np.uint64(P) + p)
    rad = np.sqrt(np.float32(-2.0) * np.log(u01_open(d[0]), dtype=np.float32), dtype=np.float32)
    th = np.float32(6.283185307179586) * u01(d[1])
    z = np.empty((rows, 2 * P), dtype=np.float32)
    z[:, 0::2] = rad * np.cos(th, dtype=np.float32)
    z[:, 1::2] = rad * np.sin(th, dtype=np.float32)
    return z[:, :K]
