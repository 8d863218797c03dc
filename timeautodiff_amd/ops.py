"""Thin wrappers: torch CUDA(HIP) tensors -> raw pointers -> libtad_hip.so.

torch is used here only as the owner of device memory and of the current HIP
stream; every wrapper enqueues hand-written gfx950 kernels on that stream.
Tensors are float32 with a contiguous last dimension; any leading dimensions
must collapse to rows of a common stride (slices of a wider buffer are fine)."""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, BF16, CELL_GRU, CELL_LSTM, F32, RnnBwdDir, RnnDir  # noqa: F401


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_dev(t):
    if not t.is_cuda:
        raise _lib.TadError("libtad_hip ops need device tensors (no CPU path exists)")


def as2d(t):
    """(ptr, rows, cols, ld) of a float32 tensor viewed as rows x cols."""
    _chk_dev(t)
    assert t.dtype == torch.float32, t.dtype
    cols = t.shape[-1]
    assert t.stride(-1) == 1 or cols == 1, (t.shape, t.stride())
    if t.dim() == 1:
        return t.data_ptr(), 1, cols, cols
    ld = t.stride(-2) if t.shape[-2] > 1 else max(cols, t.stride(-2))
    expect = ld * t.shape[-2]
    for d in range(t.dim() - 3, -1, -1):
        if t.shape[d] > 1:
            assert t.stride(d) == expect, ("leading dims do not collapse", t.shape, t.stride())
        expect *= t.shape[d]
    return t.data_ptr(), t.numel() // cols, cols, ld


def _ptr(t):
    if t is None:
        return None
    _chk_dev(t)
    return t.data_ptr()


class LaunchTrace:
    """Optional per-call device timing: while installed (``with LaunchTrace() as tr``) every
    C-ABI call is bracketed by two events on the current stream.  ``summary()`` synchronises
    once and returns {(entry point, integer args): [calls, total_ms]}."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _TRACE
        self._prev, _TRACE = _TRACE, self
        return self

    def __exit__(self, *exc):
        global _TRACE
        _TRACE = self._prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, _, e0, e1 in self.records:
            ent = out.setdefault(key, [0, 0.0])
            ent[0] += 1
            ent[1] += e0.elapsed_time(e1)
        return out

    def summary_by_kernel(self):
        """{device kernel name (entry point name if the call names no kernel): {key: [calls, total_ms]}}."""
        torch.cuda.synchronize()
        out = {}
        for key, kern, e0, e1 in self.records:
            ent = out.setdefault(kern or key[0], {}).setdefault(key, [0, 0.0])
            ent[0] += 1
            ent[1] += e0.elapsed_time(e1)
        return out


_TRACE = None


def _call(name, *args):
    lib = _lib.load()
    if _TRACE is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        key = (name,) + tuple(a for a in args[:6] if isinstance(a, int) and abs(a) < (1 << 40))
        _TRACE.records.append((key, lib.tad_last_kernel().decode(), e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise _lib.TadError(f"{name}: {lib.tad_last_error().decode()}")


# ------------------------------------------------------------------ linear family
def linear_fwd(prec, x, w, bias, y, act=ACT_NONE, accumulate=False, bias2=None, residual=None):
    """y = act(x w^T + bias + bias2 + residual); accumulate=True uses y itself as the residual."""
    xp, M, K, ldx = as2d(x)
    wp, N, Kw, ldw = as2d(w)
    yp, My, Ny, ldy = as2d(y)
    assert K == Kw and M == My and N == Ny, (x.shape, w.shape, y.shape)
    rp, ldr = None, 0
    if accumulate:
        assert residual is None
        rp, ldr = yp, ldy
    elif residual is not None:
        rp, Mr, Nr, ldr = as2d(residual)
        assert Mr == M and Nr == N
    _call("tad_linear_fwd", prec, M, N, K, xp, ldx, wp, ldw, _ptr(bias), _ptr(bias2), rp, ldr, yp, ldy, act, _stream())


def linear_bwd_data(prec, dy, w, dx, relu_of=None, accumulate=False):
    dyp, M, N, ldy = as2d(dy)
    wp, Nw, K, ldw = as2d(w)
    dxp, Mx, Kx, ldx = as2d(dx)
    assert N == Nw and M == Mx and K == Kx, (dy.shape, w.shape, dx.shape)
    rp, ldr = None, 0
    if relu_of is not None:
        rp, Mr, Kr, ldr = as2d(relu_of)
        assert Mr == M and Kr == K
    _call("tad_linear_bwd_data", prec, M, N, K, dyp, ldy, wp, ldw, dxp, ldx, rp, ldr, int(accumulate), _stream())


def linear_bwd_weight(prec, dy, x, dw, db=None, db2=None, T=0, tshift=0):
    """dw += dy^T x (time-shifted x rows if T > 0); db (and db2) += column sums of dy in the same pass."""
    dyp, M, N, ldy = as2d(dy)
    xp, Mx, K, ldx = as2d(x)
    dwp, Nw, Kw, ldw = as2d(dw)
    assert M == Mx and N == Nw and K == Kw, (dy.shape, x.shape, dw.shape)
    for b in (db, db2):
        assert b is None or (b.is_contiguous() and b.numel() == N), (N, None if b is None else b.shape)
    _call("tad_linear_bwd_weight", prec, M, N, K, dyp, ldy, xp, ldx, dwp, ldw, _ptr(db), _ptr(db2), T, tshift, _stream())


def colsum(dy, out_sum=None, out_sum2=None, row_scale=None, row_scale_stride=1, scale_mul=1.0, out_wsum=None):
    dyp, M, N, ldy = as2d(dy)
    _call("tad_colsum", M, N, dyp, ldy, _ptr(row_scale), row_scale_stride, float(scale_mul), _ptr(out_sum), _ptr(out_sum2),
          _ptr(out_wsum), _stream())


# ------------------------------------------------------------------ recurrent family
def rnn_packed_bytes(prec, H, G):
    return _lib.load().tad_rnn_packed_bytes(prec, H, G)


def rnn_pack_weights(prec, H, G, w_hh, packed_fwd, packed_bwd):
    assert w_hh.is_contiguous() and w_hh.shape == (G * H, H), w_hh.shape
    _call("tad_rnn_pack_weights", prec, H, G, w_hh.data_ptr(), _ptr(packed_fwd), _ptr(packed_bwd), _stream())


def rnn_fwd(cell, prec, B, T, H, dirs):
    """dirs: list of dict(xproj=(tensor2d), w_packed, b_hh, out, stash, reverse)."""
    arr = (RnnDir * len(dirs))()
    for k, d in enumerate(dirs):
        xp, _, _, ldxp = as2d(d["xproj"])
        op, _, _, ldo = as2d(d["out"])
        arr[k] = RnnDir(xp, ldxp, d["w_packed"].data_ptr(), _ptr(d.get("b_hh")), op, ldo, _ptr(d.get("stash")),
                        int(d["reverse"]), 0)
    _call("tad_rnn_fwd", cell, prec, B, T, H, len(dirs), arr, _stream())


def rnn_bwd(cell, prec, B, T, H, dirs):
    """dirs: list of dict(dout, stash, out, w_packed_t, dgates, dgates_h, reverse)."""
    arr = (RnnBwdDir * len(dirs))()
    for k, d in enumerate(dirs):
        dp, _, _, ldd = as2d(d["dout"])
        op, _, _, ldo = as2d(d["out"])
        gp, _, _, ldg = as2d(d["dgates"])
        ghp = None
        if d.get("dgates_h") is not None:
            ghp, _, _, ldg2 = as2d(d["dgates_h"])
            assert ldg2 == ldg
        arr[k] = RnnBwdDir(dp, ldd, d["stash"].data_ptr(), op, ldo, d["w_packed_t"].data_ptr(), gp, ghp, ldg,
                           int(d["reverse"]), 0)
    _call("tad_rnn_bwd", cell, prec, B, T, H, len(dirs), arr, _stream())


# ------------------------------------------------------------------ elementwise family
def add_noise(x0, noise, i_idx, alphas, x_noisy, i_f):
    B, T, K = x0.shape
    assert x0.is_contiguous() and noise.is_contiguous() and x_noisy.is_contiguous()
    assert i_idx.dtype == torch.int32 and i_idx.is_cuda
    _call("tad_add_noise", B, T, K, x0.data_ptr(), noise.data_ptr(), i_idx.data_ptr(), alphas.data_ptr(),
          x_noisy.data_ptr(), _ptr(i_f), _stream())


def sampler_update(x, pred, z, c1, c2, c3, last):
    assert x.is_contiguous() and pred.is_contiguous() and z.is_contiguous()
    _call("tad_sampler_update", x.numel(), x.data_ptr(), pred.data_ptr(), z.data_ptr(), float(c1), float(c2), float(c3),
          int(last), _stream())


def mse_loss(pred, target, grad_scale, dpred, loss_sum):
    assert pred.is_contiguous() and target.is_contiguous()
    _call("tad_mse_loss", pred.numel(), pred.data_ptr(), target.data_ptr(), float(grad_scale), _ptr(dpred), _ptr(loss_sum),
          _stream())


def sincos_features(x_nums, out):
    xp, rows, n, ldx = as2d(x_nums)
    op, rows2, cols, ldo = as2d(out)
    assert rows == rows2 and cols == 16 * n
    _call("tad_sincos_features", rows, n, xp, ldx, op, ldo, _stream())


def _i32(lst):
    return (C.c_int32 * len(lst))(*lst)


def embed_fwd(x_disc, tables, tab_off, cards, E, out):
    xp, rows, nd, ldx = as2d(x_disc)
    op, rows2, cols, ldo = as2d(out)
    assert rows == rows2 and cols == nd * E and len(cards) == nd
    _call("tad_embed_fwd", rows, nd, E, xp, ldx, tables.data_ptr(), _i32(tab_off), _i32(cards), op, ldo, _stream())


def embed_bwd(x_disc, dout, dtables, tab_off, cards, E):
    xp, rows, nd, ldx = as2d(x_disc)
    dp, rows2, cols, ldo = as2d(dout)
    assert rows == rows2 and cols == nd * E
    _call("tad_embed_bwd", rows, nd, E, xp, ldx, dp, ldo, dtables.data_ptr(), _i32(tab_off), _i32(cards), _stream())


def posenc_fwd(t, t_stride, rows, H, inv_max, w, b, scale, shift, out):
    op, rows2, cols, ldo = as2d(out)
    assert rows == rows2 and cols == H
    _call("tad_posenc_fwd", rows, H, t.data_ptr(), t_stride, float(inv_max), w.data_ptr(), b.data_ptr(), scale.data_ptr(),
          shift.data_ptr(), op, ldo, _stream())


def silu_fwd(z, a):
    assert z.is_contiguous() and a.is_contiguous()
    _call("tad_silu_fwd", z.numel(), z.data_ptr(), a.data_ptr(), _stream())


def silu_bwd(z, da, dz):
    assert z.is_contiguous() and da.is_contiguous() and dz.is_contiguous()
    _call("tad_silu_bwd", z.numel(), z.data_ptr(), da.data_ptr(), dz.data_ptr(), _stream())


def layernorm_fwd(x, gamma, beta, y, mean_rstd, eps=1e-5):
    xp, rows, D, ldx = as2d(x)
    yp, _, _, ldy = as2d(y)
    _call("tad_layernorm_fwd", rows, D, xp, ldx, gamma.data_ptr(), beta.data_ptr(), float(eps), yp, ldy, _ptr(mean_rstd), _stream())


def layernorm_bwd(dy, x, gamma, mean_rstd, dx, dgamma, dbeta):
    dyp, rows, D, lddy = as2d(dy)
    xp, _, _, ldx = as2d(x)
    dxp, _, _, lddx = as2d(dx)
    _call("tad_layernorm_bwd", rows, D, dyp, lddy, xp, ldx, gamma.data_ptr(), mean_rstd.data_ptr(), dxp, lddx,
          dgamma.data_ptr(), dbeta.data_ptr(), _stream())


def gather_rows(src, idx, out):
    """out[r] = src[idx[r]] over the leading dimension; idx int64 on device."""
    assert src.is_contiguous() and out.is_contiguous() and idx.dtype == torch.int64 and idx.is_cuda
    rows = idx.numel()
    row_len = src[0].numel()
    assert out.numel() == rows * row_len
    _call("tad_gather_rows", rows, row_len, src.data_ptr(), idx.data_ptr(), out.data_ptr(), _stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    _call("tad_adam_step", n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), float(lr), float(beta1), float(beta2),
          float(eps), float(weight_decay), int(step), _stream())


# ------------------------------------------------------------------ VAE family
def vae_reparam_fwd(mu, lv, eps, z, kld_sum):
    _call("tad_vae_reparam_fwd", mu.numel(), mu.data_ptr(), lv.data_ptr(), eps.data_ptr(), z.data_ptr(), _ptr(kld_sum), _stream())


def vae_reparam_bwd(dz, mu, lv, eps, kld_scale, dmu, dlv):
    _call("tad_vae_reparam_bwd", dz.numel(), dz.data_ptr(), mu.data_ptr(), lv.data_ptr(), eps.data_ptr(), float(kld_scale),
          dmu.data_ptr(), dlv.data_ptr(), _stream())


def vae_heads_loss(heads, mask, target, nb, nc, nn, cards, s_bce, s_ce, s_mse, dheads, loss_parts):
    hp, rows, NH, ldh = as2d(heads)
    mp, rows2, F, ldf = as2d(mask)
    tp, rows3, F2, ldf2 = as2d(target)
    assert rows == rows2 == rows3 and F == F2 and ldf == ldf2
    if dheads is not None:
        dp, _, _, ldd = as2d(dheads)
        assert ldd == ldh
    else:
        dp = None
    _call("tad_vae_heads_loss", rows, nb, nc, nn, _i32(list(cards)) if nc else None, hp, ldh, mp, tp, ldf, float(s_bce),
          float(s_ce), float(s_mse), dp, _ptr(loss_parts), _stream())


def vae_prepare(resp, cond, mask, inp, target):
    _call("tad_vae_prepare", resp.numel(), resp.data_ptr(), _ptr(cond), _ptr(mask), _ptr(inp), _ptr(target), _stream())
