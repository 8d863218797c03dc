"""Elementary ops of the hot path, forward and backward, in numpy.

Oracle code: test infrastructure only (see ``oracle/__init__.py``).
All functions take/return arrays of one floating dtype (float32 mirrors the
reference; float64 gives a tighter truth for tolerance studies).
"""
import math

import numpy as np


# ----------------------------------------------------------------------------
# activations
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def silu(x):
    return x * sigmoid(x)


def silu_grad(x):
    s = sigmoid(x)
    return s * (1.0 + x * (1.0 - s))


# ----------------------------------------------------------------------------
# Matrix products.  With operand rounding switched on (``with bf16_operands():``) every
# product that the HIP path evaluates on bf16 matrix cores rounds BOTH operands to bf16
# (round-to-nearest-even) and accumulates in the working dtype -- the documented
# semantics of the bf16 compute mode.  Off by default (plain float products).
_ROUND_OPERANDS = False
# With storage rounding on as well (``with bf16_operands(storage=True):``) the per-timestep tensors of a
# recurrent layer that the HIP path keeps in HBM as bf16 (input-side pre-activations, the gate stash read by
# the backward, the pre-activation gradients) are rounded to bf16 where the kernels store them.  The HIP
# path does so for every layer with an even hidden size (an input width that is not a multiple of 8 is padded
# with zero columns there, which changes no value).
_ROUND_STORAGE = False
# ... except the input-side pre-activations of a layer whose sequence kernel forms them on chip (the HIP path's first
# encoder layer of the VAE at hidden size 512 reads at most 32 input columns and multiplies them itself): they reach the
# gates unrounded (``with rnn_xproj_on_chip():`` inside the above).
_XPROJ_ON_CHIP = False


def bf16_round(a):
    """Round float32 values to the nearest bfloat16 (ties to even), returned as float32."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32).reshape(a32.shape)


class bf16_operands:
    def __init__(self, storage=False):
        self.storage = storage

    def __enter__(self):
        global _ROUND_OPERANDS, _ROUND_STORAGE
        self.prev = (_ROUND_OPERANDS, _ROUND_STORAGE)
        _ROUND_OPERANDS, _ROUND_STORAGE = True, self.storage

    def __exit__(self, *exc):
        global _ROUND_OPERANDS, _ROUND_STORAGE
        _ROUND_OPERANDS, _ROUND_STORAGE = self.prev


class rnn_xproj_on_chip:
    def __enter__(self):
        global _XPROJ_ON_CHIP
        self.prev = _XPROJ_ON_CHIP
        _XPROJ_ON_CHIP = True

    def __exit__(self, *exc):
        global _XPROJ_ON_CHIP
        _XPROJ_ON_CHIP = self.prev


def rnn_xproj_store_bf16(H, in_dim):
    """True if a recurrent layer of this shape reads its input-side pre-activations back from bf16 storage."""
    return rnn_store_bf16(H, in_dim) and not _XPROJ_ON_CHIP


def rnn_store_bf16(H, in_dim):
    """True if a recurrent layer of this shape keeps its per-timestep tensors as bf16 (see above)."""
    return _ROUND_STORAGE and H % 2 == 0        # any input width: the kernels pad it to a multiple of 8 with zeros


def stored(a, as_bf16):
    """The value a tensor has after a trip through its HBM storage type."""
    return bf16_round(a).astype(a.dtype) if as_bf16 else a


def mm(a, b):
    """a @ b on the (emulated) matrix cores."""
    if _ROUND_OPERANDS:
        dt = np.result_type(a, b)
        return (bf16_round(a).astype(np.float64) @ bf16_round(b).astype(np.float64)).astype(dt)
    return a @ b


# ----------------------------------------------------------------------------
# nn.Linear: y = x W^T + b   (weights are [out, in] as in torch)
def linear_fwd(x, W, b=None):
    y = mm(x, W.T)
    if b is not None:
        y = y + b
    return y


def linear_bwd(x, W, dy, need_dx=True):
    """Returns (dx, dW, db) for y = x W^T + b."""
    K = W.shape[1]
    N = W.shape[0]
    x2 = x.reshape(-1, K)
    dy2 = dy.reshape(-1, N)
    dW = mm(dy2.T, x2)
    db = dy2.sum(0)
    dx = mm(dy2, W).reshape(x.shape) if need_dx else None
    return dx, dW, db


# ----------------------------------------------------------------------------
# compute_sine_cosine -- DIFF.py:14-32, VAE.py:24-42
def sincos_features(v, num_terms=8):
    """v [..., n] -> [..., 16 n]: all sines (feature-major, k-minor), then all
    cosines.  angle = (2^k * pi) * v evaluated in v's dtype, like the reference
    (``2**arange(k).float() * tensor(pi) * v.unsqueeze(-1)``)."""
    dt = v.dtype
    freqs = (2.0 ** np.arange(num_terms)).astype(dt) * np.asarray(math.pi, dtype=dt)
    ang = freqs * v[..., None]                       # [..., n, k]
    s = np.sin(ang).reshape(*v.shape[:-1], -1)
    c = np.cos(ang).reshape(*v.shape[:-1], -1)
    return np.concatenate([s, c], axis=-1)


# ----------------------------------------------------------------------------
# PositionalEncoding -- DIFF.py:129-146
def posenc_tables(dim, max_value, dtype=np.float32):
    """scale/shift tensors of DIFF.py:134-139 (not parameters)."""
    linear_dim = dim // 2
    periodic_dim = dim - linear_dim
    # reference: exp(-2 * arange(P).float() * log(max) / P) in float32
    ar = np.arange(periodic_dim).astype(np.float32)
    scale = np.exp((np.float32(-2.0) * ar * np.float32(math.log(max_value))) / np.float32(periodic_dim))
    shift = np.zeros(periodic_dim, dtype=np.float32)
    shift[::2] = np.float32(0.5 * math.pi)
    return scale.astype(dtype), shift.astype(dtype)


def posenc_fwd(t, W, b, max_value, scale, shift):
    """t [..., 1] -> [..., dim] = cat[Linear(1->dim//2)(t/max), sin(t*scale+shift)]."""
    periodic = np.sin(t * scale + shift)
    linear = (t / np.asarray(max_value, dtype=t.dtype)) @ W.T + b
    return np.concatenate([linear, periodic], axis=-1)


def posenc_bwd(t, W, dout, max_value):
    """Grad wrt the linear_proj parameters only (t carries no gradient)."""
    L = W.shape[0]
    d = dout[..., :L].reshape(-1, L)
    tn = (t / np.asarray(max_value, dtype=t.dtype)).reshape(-1, 1)
    dW = d.T @ tn
    db = d.sum(0)
    return dW, db


# ----------------------------------------------------------------------------
# nn.LayerNorm over the last dim, eps 1e-5, biased variance -- DIFF.py:183,221
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + np.asarray(eps, dtype=x.dtype))
    xhat = xc * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layernorm_bwd(dy, gamma, cache):
    xhat, rstd = cache
    D = xhat.shape[-1]
    dgamma = (dy * xhat).reshape(-1, D).sum(0)
    dbeta = dy.reshape(-1, D).sum(0)
    dxhat = dy * gamma
    dx = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------
# DDPM schedule -- DIFF.py:88-95
def schedule(steps=100):
    """betas = 1e-5*(1-u) + 0.1*u, u = linspace(0,1,steps); alphas = cumprod(1-betas),
    all in float32 like the reference's module globals."""
    u = np.linspace(0.0, 1.0, steps, dtype=np.float64).astype(np.float32)
    # torch.linspace(float32) is computed as start + i*step symmetric halves; the
    # committed golden table pins whichever rounding we produce here.
    betas = (np.float32(1e-5) * (np.float32(1.0) - u) + np.float32(0.1) * u).astype(np.float32)
    alphas = np.cumprod((np.float32(1.0) - betas).astype(np.float32), dtype=np.float32)
    return betas, alphas


# ----------------------------------------------------------------------------
# add_noise -- DIFF.py:102-118 (get_gp_covariance returns the identity, so the
# Cholesky factor is the identity and ``L @ noise`` is the noise itself)
def add_noise(x, i_idx, noise, alphas):
    """x [B,T,K]; i_idx [B] int diffusion step per sample; noise [B,T,K]."""
    a = alphas[i_idx].astype(x.dtype)[:, None, None]
    return np.sqrt(a) * x + np.sqrt(1 - a) * noise


# ----------------------------------------------------------------------------
# one reverse-diffusion update -- DIFF.py:341-351
def sampler_update(x, pred, z, beta, alpha, last):
    """x <- (x - beta*pred/sqrt(1-alpha))/sqrt(1-beta) + sqrt(beta)*z; at the last
    step (diff_step == 0) the mean formula is applied a second time to the
    already-updated x with the same prediction (DIFF.py:350-351)."""
    dt = x.dtype
    beta = np.asarray(beta, dtype=dt)
    alpha = np.asarray(alpha, dtype=dt)
    c = 1.0 / np.sqrt(1 - beta)
    x = c * (x - beta * pred / np.sqrt(1 - alpha)) + np.sqrt(beta) * z
    if last:
        x = c * (x - beta * pred / np.sqrt(1 - alpha))
    return x
