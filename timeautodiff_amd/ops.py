"""Thin wrappers: torch CUDA(HIP) tensors -> raw pointers -> libtad_hip.so.

torch is used here only as the owner of device memory and of the current HIP
stream; every wrapper enqueues hand-written gfx950 kernels on that stream.
Tensors are float32 with a contiguous last dimension; any leading dimensions
must collapse to rows of a common stride (slices of a wider buffer are fine)."""
import ctypes as C
import os

import weakref

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, ACT_DSILU, ACT_SILU_Z32, BF16, CELL_GRU, CELL_LSTM, F32, RnnBwdDir, RnnDir, RnnXin  # noqa: F401


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current HIP stream on the current device (the fast C accessor when this torch has
    it: the Python-level lookup costs several microseconds per launch, 84 launches per training step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def norm_device(dev=None):
    """An indexed device: 'cuda' / None -> the current device, so that 'cuda' and 'cuda:0' compare equal."""
    dev = torch.device("cuda" if dev is None else dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def device_scope(pick):
    """Decorator for the package's entry points: the call runs with the device ``pick(*args, **kwargs)`` names as
    the current one (None or the device that is current already: no switch), because every launch goes to the
    current device's stream."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrap(*args, **kwargs):
            dev = pick(*args, **kwargs)
            if dev is not None:
                dev = torch.device(dev)
            if dev is None or dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        return wrap
    return deco


def _first_cuda(*cands):
    """Device of the first CUDA tensor among the candidates, else of the first object that knows its engine."""
    for c in cands:
        if isinstance(c, torch.Tensor) and c.is_cuda:
            return c.device
    for c in cands:
        eng = getattr(c, "_eng", None) or getattr(c, "eng", None)
        if eng is not None:
            return eng.device
    return None


def _chk_dev(t):
    """Launches go to the current stream of the CURRENT device: an operand on another device would hand that
    stream pointers of a GPU it cannot reach (a memory fault), so it is rejected here."""
    if not t.is_cuda:
        raise _lib.TadError("libtad_hip ops need device tensors (no CPU path exists)")
    if t.device.index != torch.cuda.current_device():
        raise _lib.TadError(f"operand on {t.device} while cuda:{torch.cuda.current_device()} is the current device: "
                            f"call under torch.cuda.device({t.device.index}) (the engines and trainers of this package do)")


def as2d(t, dtype=torch.float32):
    """(ptr, rows, cols, ld) of a float32 (or ``dtype``) tensor viewed as rows x cols."""
    _chk_dev(t)
    assert t.dtype == dtype, (t.dtype, dtype)
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

    def _aggregate(self, group):
        """{group(record): {key: [calls, total_ms, mean algorithmic FLOPs per call, mean algorithmic HBM bytes per call,
        mean issued FLOPs per call]}}: every record adds its OWN cost (calls that share a key -- the same entry point
        with the same leading integer arguments -- may still differ in the work they do: the weight-gradient launches
        of the four recurrent layers of a step do)."""
        torch.cuda.synchronize()
        out = {}
        for rec in self.records:
            key, _, e0, e1, cost = rec
            ent = out.setdefault(group(rec), {}).setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
            ent[0] += 1
            ent[1] += e0.elapsed_time(e1)
            ent[2] += cost[0]
            ent[3] += cost[1]
            ent[4] += cost[2] if len(cost) > 2 else cost[0]
        for keys in out.values():
            for ent in keys.values():
                ent[2] /= ent[0]
                ent[3] /= ent[0]
                ent[4] /= ent[0]
        return out

    def summary(self):
        """{(entry point, integer args): [calls, total_ms, algorithmic FLOPs per call, algorithmic HBM bytes per call,
        issued FLOPs per call] (means over the calls of the key)}."""
        return self._aggregate(lambda rec: None).get(None, {})

    def summary_by_kernel(self):
        """{device kernel name (entry point name if the call names no kernel): {key: [calls, total_ms, flops, bytes, issued flops]}}."""
        return self._aggregate(lambda rec: rec[1] or rec[0][0])


_TRACE = None


def tracing():
    """The installed LaunchTrace, or None."""
    return _TRACE


def _call(name, *args, cost=(0.0, 0.0)):
    """Calls a C-ABI entry point.  ``cost`` = (algorithmic FLOPs, algorithmic HBM bytes: every operand and result
    once at its storage width[, FLOPs the kernel issues, padding of its tiles included]) of the call, recorded by an
    installed LaunchTrace."""
    lib = _lib.load()
    if _TRACE is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        key = (name,) + tuple(a for a in args[:6] if isinstance(a, int) and abs(a) < (1 << 40))
        _TRACE.records.append((key, lib.tad_last_kernel().decode(), e0, e1, cost))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise _lib.TadError(f"{name}: {lib.tad_last_error().decode()}")


# ------------------------------------------------------------------ linear family
def _cost(flops, nbytes, useful):
    """(algorithmic FLOPs, bytes, issued FLOPs): ``useful`` < 1 where a share of the operand is structural padding (the
    spare fourth slot of a GRU's gate-interleaved tensors meets zero weights: 3/4 of the issued product is the model's)."""
    return (flops * useful, nbytes, flops)


def linear_fwd(prec, x, w, bias, y, act=ACT_NONE, accumulate=False, bias2=None, residual=None, useful=1.0):
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
    _call("tad_linear_fwd", prec, M, N, K, xp, ldx, wp, ldw, _ptr(bias), _ptr(bias2), rp, ldr, yp, ldy, act, _stream(),
          cost=_cost(2.0 * M * N * K, 4.0 * (M * K + N * K + M * N + (M * N if rp else 0)), useful))


def linear_bwd_data(prec, dy, w, dx, relu_of=None, accumulate=False, useful=1.0):
    dyp, M, N, ldy = as2d(dy)
    wp, Nw, K, ldw = as2d(w)
    dxp, Mx, Kx, ldx = as2d(dx)
    assert N == Nw and M == Mx and K == Kx, (dy.shape, w.shape, dx.shape)
    rp, ldr = None, 0
    if relu_of is not None:
        rp, Mr, Kr, ldr = as2d(relu_of)
        assert Mr == M and Kr == K
    _call("tad_linear_bwd_data", prec, M, N, K, dyp, ldy, wp, ldw, dxp, ldx, rp, ldr, int(accumulate), _stream(),
          cost=_cost(2.0 * M * N * K, 4.0 * (M * N + N * K + M * K * (1 + bool(rp) + bool(accumulate))), useful))


def _gate_map(gate_map):
    """(H, G, (slot0..slot3)) -> the host int32[6] array of the C ABI (None passes NULL)."""
    if gate_map is None:
        return None
    H, G, slots = gate_map
    return (C.c_int32 * 6)(H, G, *slots)


def linear_bwd_weight(prec, dy, x, dw, db=None, db2=None, T=0, tshift=0, gate_map=None):
    """dw += dy^T x (time-shifted x rows if T > 0); db (and db2) += column sums of dy in the same pass.
    dy may be a bfloat16 tensor (bf16 precision only).  gate_map=(H, G, slots): dy is gate-interleaved
    (column blk*4H + 4u + s) and its product rows go to torch's gate-major rows blk*G*H + slots[s]*H + u of dw/db."""
    b16 = dy.dtype == torch.bfloat16
    dyp, M, N, ldy = as2d(dy, dy.dtype)
    xp, Mx, K, ldx = as2d(x)
    dwp, Nw, Kw, ldw = as2d(dw)
    if gate_map is None:
        n_out = N
    else:
        assert N % (4 * gate_map[0]) == 0, (N, gate_map)
        n_out = N // 4 * gate_map[1]
    assert M == Mx and n_out == Nw and K == Kw, (dy.shape, x.shape, dw.shape, gate_map)
    for b in (db, db2):
        assert b is None or (b.is_contiguous() and b.numel() == n_out), (n_out, None if b is None else b.shape)
    cost = (2.0 * M * n_out * K, dy.element_size() * M * N + 4.0 * M * K + 4.0 * n_out * K, 2.0 * M * N * K)
    if b16:
        assert prec == BF16
        _call("tad_linear_bwd_weight_b16", M, N, K, dyp, ldy, xp, ldx, dwp, ldw, _ptr(db), _ptr(db2), T, tshift,
              _gate_map(gate_map), _stream(), cost=cost)
    else:
        _call("tad_linear_bwd_weight", prec, M, N, K, dyp, ldy, xp, ldx, dwp, ldw, _ptr(db), _ptr(db2), T, tshift,
              _gate_map(gate_map), _stream(), cost=cost)


def linear_fwd_b16(x, w16, bias, y, act=ACT_NONE, useful=1.0):
    """y = act(x w16^T + bias): x fp32, w16 bfloat16 [N,K], y float32 or bfloat16 (bf16 matrix cores)."""
    xp, M, K, ldx = as2d(x)
    wp, N, Kw, ldw = as2d(w16, torch.bfloat16)
    yp, My, Ny, ldy = as2d(y, y.dtype)
    assert K == Kw and M == My and N == Ny, (x.shape, w16.shape, y.shape)
    _call("tad_linear_fwd_b16", M, N, K, xp, ldx, wp, ldw, _ptr(bias), yp, BF16 if y.dtype == torch.bfloat16 else F32, ldy,
          act, _stream(), cost=_cost(2.0 * M * N * K, 4.0 * M * K + 2.0 * N * K + y.element_size() * M * N, useful))


def linear_fwd_b16x(x16, w16, bias, y, act=ACT_NONE, useful=1.0):
    """y = act(x16 w16^T + bias): x16 bfloat16 [M,K], w16 bfloat16 [N,K] (any row stride that is a multiple of 8), y float32 or
    bfloat16; the square-tiled kernel, for weights too large for mm16_nt's scheme."""
    xp, M, K, ldx = as2d(x16, torch.bfloat16)
    wp, N, Kw, ldw = as2d(w16, torch.bfloat16)
    yp, My, Ny, ldy = as2d(y, y.dtype)
    assert K == Kw and M == My and N == Ny, (x16.shape, w16.shape, y.shape)
    _call("tad_linear_fwd_b16x", M, N, K, xp, ldx, wp, ldw, _ptr(bias), yp, BF16 if y.dtype == torch.bfloat16 else F32, ldy,
          act, _stream(), cost=_cost(2.0 * M * N * K, 2.0 * M * K + 2.0 * N * K + y.element_size() * M * N, useful))


def linear_bwd_data_b16(dy16, w16, dx, relu_of=None, accumulate=False, useful=1.0):
    """dx = dy16 w16 (+ dx), masked where relu_of <= 0: dy16 bfloat16 [M,N], w16 bfloat16 [N,K], dx fp32."""
    dyp, M, N, ldy = as2d(dy16, torch.bfloat16)
    wp, Nw, K, ldw = as2d(w16, torch.bfloat16)
    dxp, Mx, Kx, ldx = as2d(dx)
    assert N == Nw and M == Mx and K == Kx, (dy16.shape, w16.shape, dx.shape)
    rp, ldr = None, 0
    if relu_of is not None:
        rp, Mr, Kr, ldr = as2d(relu_of)
        assert Mr == M and Kr == K
    _call("tad_linear_bwd_data_b16", M, N, K, dyp, ldy, wp, ldw, dxp, ldx, rp, ldr, int(accumulate), _stream(),
          cost=_cost(2.0 * M * N * K, 2.0 * (M * N + N * K) + 4.0 * M * K * (1 + bool(rp) + bool(accumulate)), useful))


def linear_bwd_data_b16o(dy16, w16, dx16, relu_of16=None, useful=1.0):
    """dx16 = dy16 w16, masked where relu_of16 <= 0: every operand bfloat16 (a dx that only matrix products read)."""
    b16 = torch.bfloat16
    dyp, M, N, ldy = as2d(dy16, b16)
    wp, Nw, K, ldw = as2d(w16, b16)
    dxp, Mx, Kx, ldx = as2d(dx16, b16)
    assert N == Nw and M == Mx and K == Kx, (dy16.shape, w16.shape, dx16.shape)
    rp, ldr = None, 0
    if relu_of16 is not None:
        rp, Mr, Kr, ldr = as2d(relu_of16, b16)
        assert Mr == M and Kr == K
    _call("tad_linear_bwd_data_b16o", M, N, K, dyp, ldy, wp, ldw, dxp, ldx, rp, ldr, _stream(),
          cost=_cost(2.0 * M * N * K, 2.0 * (M * N + N * K) + 2.0 * M * K * (1 + bool(rp)), useful))


def _dt(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def _sfx(out):
    """Entry-point suffix by the element type of a result: '' float32, '_b16' bfloat16."""
    assert out.dtype in (torch.float32, torch.bfloat16), out.dtype
    return "_b16" if out.dtype == torch.bfloat16 else ""


def w16_image(n, k, device):
    """Zeroed storage of a padded bf16 weight image for mm16_nt: [n rounded up to 64, k rounded up to 128]."""
    return torch.zeros(-(-n // 64) * 64, -(-k // 128) * 128, dtype=torch.bfloat16, device=device)


LN_FC_MAX_D = int(os.environ.get("TAD_LN_FC_MAX_D", "1024"))     # widest row of the fused LayerNorm + fc launches (experiments: 512)
MM16_NT_MAX_K = 1024        # tad_mm16_nt keeps a wave's 32 rows of A in registers: K at most 1024
# plain products (bias, ReLU) that leave tad_mm16_nt for the square-tiled kernel: K from NT_TILED_K on, or from
# NT_TILED_K_NARROW on with at most NT_TILED_N_NARROW result columns (TAD_NT_TILED=0: none)
NT_TILED = os.environ.get("TAD_NT_TILED", "1") != "0"
NT_TILED_K = int(os.environ.get("TAD_NT_TILED_K", "768"))
NT_TILED_K_NARROW = int(os.environ.get("TAD_NT_TILED_K_NARROW", "384"))
NT_TILED_N_NARROW = int(os.environ.get("TAD_NT_TILED_N_NARROW", "512"))


def mm16_nt(a16, w16, bias, c, act=ACT_NONE, residual=None, relu_of=None, z_out=None, useful=1.0):
    """c = act(a16 w16[:N, :K]^T + bias + residual), zeroed where relu_of <= 0 (act=ACT_DSILU: multiplied by silu'(relu_of) instead): a16 [M,K] bfloat16, w16 a padded
    bfloat16 weight image (``w16_image``: rows >= N rounded up to 64, columns K rounded up to 128, zeros beyond K),
    c [M,N] float32 or bfloat16, residual / relu_of float32 or bfloat16 [M,N]; z_out (bfloat16) receives the value
    before ``act``."""
    ap, M, K, lda = as2d(a16, torch.bfloat16)
    cp, Mc, N, ldc = as2d(c, c.dtype)
    wp, Nw, Kw, ldw = as2d(w16, torch.bfloat16)
    assert M == Mc and Nw >= -(-N // 64) * 64 and Kw >= -(-K // 128) * 128 and Kw - K < 128, (a16.shape, w16.shape, c.shape)
    plain = residual is None and relu_of is None and z_out is None and act in (ACT_NONE, ACT_RELU)
    if K > MM16_NT_MAX_K:
        # more of A than a wave keeps in registers (proj and cond_output + input_proj at hidden size 512): the square-tiled
        # kernel on the same operands, for the plain epilogues it has
        assert plain, "mm16_nt: K > 1024 with a fused operand"
        return linear_fwd_b16x(a16, w16[:N, :K], bias, c, act=act, useful=useful)
    if plain and NT_TILED and (K >= NT_TILED_K or (K >= NT_TILED_K_NARROW and N <= NT_TILED_N_NARROW)):
        # long reductions: tad_mm16_nt holds 32 rows x K of A per wave in registers, so from K = 592 on a CU runs four waves
        # and the loads of A, the ring and the stores follow each other; the square-tiled kernel measured 15-45 % faster
        # for K >= 800 at every N and 3-30 % for K = 400-512 up to N = 512 (98 304 rows; scripts/nt_vs_tiled_k.py).  The
        # choice depends on N and K only: a batch split differently meets the same kernels.
        return linear_fwd_b16x(a16, w16[:N, :K], bias, c, act=act, useful=useful)
    rp = ldr = up = ldu = zp = ldz = 0
    rdt = udt = F32
    if residual is not None:
        rp, Mr, Nr, ldr = as2d(residual, residual.dtype)
        rdt = _dt(residual)
        assert Mr == M and Nr == N
    if relu_of is not None:
        up, Mu, Nu, ldu = as2d(relu_of, relu_of.dtype)
        udt = _dt(relu_of)
        assert Mu == M and Nu == N
    if z_out is not None:
        zp, Mz, Nz, ldz = as2d(z_out, torch.float32 if act == ACT_SILU_Z32 else torch.bfloat16)
        assert Mz == M and Nz == N
    _call("tad_mm16_nt", M, N, K, ap, lda, wp, ldw, _ptr(bias), rp or None, rdt, ldr, up or None, udt, ldu, cp, _dt(c), ldc,
          zp or None, ldz, act, _stream(),
          cost=_cost(2.0 * M * N * K, 2.0 * (M * K + N * K) + c.element_size() * M * N
                     + (residual.element_size() * M * N if residual is not None else 0)
                     + (relu_of.element_size() * M * N if relu_of is not None else 0)
                     + (z_out.element_size() * M * N if z_out is not None else 0), useful))


TN_MAXJOBS = 12
TN_TILE_A = 256          # columns of A per workgroup tile of tad_mm16_tn (B is multiplied in 32-column blocks)


def mm16_tn(a16, jobs, T=0):
    """Weight gradients from bf16-stored operands, one launch for all ``jobs`` (at most ``TN_MAXJOBS`` = 12):
    job = dict(b=bfloat16 [M, nb], c=float32 [rows, nb] (accumulated into), a=None (own A operand, bfloat16 [M, lda];
    default: the shared ``a16``, which may be None when every job brings one), a0=0, na=None (columns of the A operand),
    shift=0, db=None, db2=None, gate_map=None).  c += A[:, a0:a0+na]^T b', b' = b with rows shifted by ``shift`` inside
    windows of T rows (zeros outside the window).  A contiguous c up to 7 columns narrower than b takes the leading
    columns of the product (b's extra columns are zero padding)."""
    ap, M, Na, lda = as2d(a16, torch.bfloat16) if a16 is not None else (None, None, 0, 0)
    arr = (_lib.TnJob * len(jobs))()
    flops = bytes_ = issued = 0.0
    own_a = set()
    for k, j in enumerate(jobs):
        b, c = j["b"], j["c"]
        bp, Mb, nb, ldb = as2d(b, torch.bfloat16)
        cp, rows, nbc, ldc = as2d(c)
        ja = j.get("a")
        if ja is not None:
            jap, Mj, Naj, jlda = as2d(ja, torch.bfloat16)
            if ja.data_ptr() not in own_a:
                own_a.add(ja.data_ptr())
                bytes_ += 2.0 * Mj * Naj
        else:
            assert a16 is not None, "a job without its own A operand needs the shared one"
            jap, Mj, Naj, jlda = None, M, Na, 0
        M = Mj if M is None else M
        a0 = j.get("a0", 0)
        na = j.get("na") or (Naj - a0)
        gm = j.get("gate_map")
        n_out = na if gm is None else na // 4 * gm[1]
        assert Mb == M and Mj == M and rows == n_out and (nbc == nb or (nb - 8 < nbc < nb and ldc == nbc)), (b.shape, c.shape, a0, na, gm)
        for d in (j.get("db"), j.get("db2")):
            assert d is None or (d.is_contiguous() and d.numel() == n_out)
        g6 = (C.c_int32 * 6)(*( (gm[0], gm[1], *gm[2]) if gm is not None else (0, 0, 0, 0, 0, 0)))
        arr[k] = _lib.TnJob(bp, ldb, nb, j.get("shift", 0), cp, ldc, _ptr(j.get("db")), _ptr(j.get("db2")), a0, na, g6, jap, jlda)
        # algorithmic: the rows and columns of C the product fills (a GRU's spare slot and B's zero padding produce
        # nothing); issued: what the kernel multiplies (whole 32-column blocks of both operands)
        flops += 2.0 * M * n_out * nbc
        issued += 2.0 * M * (-(-na // 32) * 32) * (-(-nb // 32) * 32)
        bytes_ += 2.0 * M * nb + 4.0 * n_out * nbc
    if a16 is not None:
        bytes_ += 2.0 * M * Na
    _call("tad_mm16_tn", M, T, ap, lda, len(jobs), arr, _stream(), cost=(flops, bytes_, issued))


def cvt_bf16(src, dst16, relu_of=None):
    """dst16 (bfloat16) = src (float32), zero where relu_of (bfloat16) <= 0; all [rows, cols] row-strided views."""
    sp, rows, cols, lds = as2d(src)
    dp, rd, cd, ldd = as2d(dst16, torch.bfloat16)
    assert rows == rd and cols == cd, (src.shape, dst16.shape)
    mp, ldm = None, 0
    if relu_of is not None:
        mp, rm, cm, ldm = as2d(relu_of, torch.bfloat16)
        assert rm == rows and cm == cols
    _call("tad_cvt_bf16", rows, cols, sp, lds, dp, ldd, mp, ldm, _stream(), cost=(0.0, (6.0 + 2.0 * (mp is not None)) * rows * cols))


def pack_w16(jobs):
    """jobs: list of (w float32 [n, k] view, image bfloat16 2-d, transpose): image[:n, :k] = w (or image[:k, :n] = w^T)."""
    if not jobs:
        return
    arr = (_lib.PackW16Job * len(jobs))()
    for i, (w, img, tr) in enumerate(jobs):
        wp, n, k, ldw = as2d(w)
        ip, ir, ic, ldi = as2d(img, torch.bfloat16)
        assert (ir >= k and ic >= n) if tr else (ir >= n and ic >= k), (w.shape, img.shape, tr)
        arr[i] = _lib.PackW16Job(wp, ldw, n, k, ip, ldi, int(bool(tr)), 0)
    _call("tad_pack_w16", len(jobs), arr, _stream())


def enc_grad_b16(g16, s1, s2, W, col1, col2, L, mul1, mul2, sums, gb1, gw1, gb2, gw2, round_w=True):
    """Bias / weight gradients of two Linear(1 -> L) maps feeding columns col1.. / col2.. of a Linear with weight W, from
    the bf16 gradient g16 of that Linear's pre-activations and the per-row scalars s1 / s2 (see tad_enc_grad_b16)."""
    gp, M, N, ldg = as2d(g16, torch.bfloat16)
    assert W.dim() == 2 and W.shape[0] == N and W.stride(1) == 1 and W.dtype == torch.float32
    assert s1.numel() == M and s2.numel() == M and s1.is_contiguous() and s2.is_contiguous()
    assert sums.numel() >= 3 * N and sums.is_contiguous() and sums.dtype == torch.float32
    for t in (gb1, gw1, gb2, gw2):
        assert t.numel() == L and t.is_contiguous() and t.dtype == torch.float32
    _call("tad_enc_grad_b16", M, N, gp, ldg, _ptr(s1), 1, _ptr(s2), 1, _ptr(W), W.stride(0), col1, col2, L, float(mul1), float(mul2),
          1 if round_w else 0, _ptr(sums), _ptr(gb1), _ptr(gw1), _ptr(gb2), _ptr(gw2), _stream())


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


def rnn_pack_weights_multi(prec, H, G, jobs):
    """jobs: list of (w_hh [G*H, H], packed_fwd or None, packed_bwd or None): every image in one launch per 16 jobs."""
    for lo in range(0, len(jobs), 16):
        part = jobs[lo:lo + 16]
        arr = (_lib.RnnPackJob * len(part))()
        for k, (w, pf, pb) in enumerate(part):
            assert w.is_contiguous() and w.shape == (G * H, H), w.shape
            arr[k].w_hh, arr[k].packed_fwd, arr[k].packed_bwd = w.data_ptr(), _ptr(pf), _ptr(pb)
        _call("tad_rnn_pack_weights_multi", prec, H, G, len(part), arr, _stream())


def _store_of(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def rnn_pack_input(cell, H, w_ih, b_ih, b_hh, w_gl, bias_gl):
    """Gate-interleaved image of the glued input-side parameters of nblk directions/heads:
    w_ih [nblk*G*H, in] -> w_gl [nblk*4H, ld >= in] (float32 or bfloat16), bias_gl [nblk*4H]."""
    G = 4 if cell == CELL_LSTM else 3
    wp, rows, in_dim, ldw = as2d(w_ih)
    assert ldw == in_dim and rows % (G * H) == 0, (w_ih.shape, w_ih.stride())
    nblk = rows // (G * H)
    gp, grows, gcols, ldg = as2d(w_gl, w_gl.dtype)
    assert grows == nblk * 4 * H and gcols >= in_dim and ldg == gcols and bias_gl.numel() == nblk * 4 * H and bias_gl.is_contiguous()
    assert b_ih.is_contiguous() and (b_hh is None or b_hh.is_contiguous())
    _call("tad_rnn_pack_input", cell, H, in_dim, nblk, wp, b_ih.data_ptr(), _ptr(b_hh), gp, _store_of(w_gl), ldg,
          bias_gl.data_ptr(), _stream())


def rnn_workspace_bytes(cell, prec, store, B, T, H, ndir):
    """Device memory the forms of rnn_fwd that split a batch tile's hidden units over several workgroups need (0: none)."""
    return _lib.load().tad_rnn_workspace_bytes(cell, prec, store, B, T, H, ndir)


def rnn_ws_status(workspace):
    """Raises if a launch that used ``workspace`` gave up waiting (synchronises the stream).  The word it reads stays set
    until the caller zeroes the workspace."""
    _lib.check(_lib.load().tad_rnn_ws_status(workspace.data_ptr(), _stream()))


_RNN_WS = {}      # id -> weak reference: workspaces that launches of rnn_fwd have used (dead ones are dropped at the next check)


def check_rnn_workspaces(device=None):
    """rnn_ws_status for every live workspace a launch has used on ``device`` (default: the current one).  The trainers
    and the sampler call it where they synchronise anyway: a launch whose workgroups gave up waiting for each other has
    written results that must not be used."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    for key, ref in list(_RNN_WS.items()):
        w = ref()
        if w is None:
            del _RNN_WS[key]
        elif w.device.index == dev:
            rnn_ws_status(w)


def rnn_fwd(cell, prec, B, T, H, dirs, workspace=None, xin=None):
    """dirs: list of dict(xproj (2-d view, float32 or bfloat16, gate-interleaved), w_packed, b_hh, out,
    gates (same dtype as xproj, or None = inference), cstate (LSTM training), reverse).  workspace: a zero-initialised
    uint8 tensor of rnn_workspace_bytes(...) bytes that no other launch in flight uses, or None.
    xin (needs a workspace; bfloat16 storage): per direction dict(x16 [B*T, kin] bfloat16 with kin in (8, 16, 24, 32),
    w_ih16 = the direction's rows of rnn_pack_input's bfloat16 image (row stride >= 32, zeros beyond kin), bias = its rows
    of bias_gl): the sequence kernel forms the input-side pre-activations itself and the dirs carry no xproj."""
    arr = (RnnDir * len(dirs))()
    xarr = None
    if xin is not None:
        assert workspace is not None and len(xin) == len(dirs)
        xarr = (RnnXin * len(dirs))()
        for k, xi in enumerate(xin):
            x16, _, kin, ldx = as2d(xi["x16"], torch.bfloat16)
            w16, _, kw, ldw = as2d(xi["w_ih16"], torch.bfloat16)
            assert kw >= kin and xi["bias"].dtype == torch.float32 and xi["bias"].is_contiguous() and xi["bias"].numel() == 4 * H
            xarr[k] = RnnXin(x16, ldx, w16, ldw, xi["bias"].data_ptr(), kin, 0)
    store = BF16 if xin is not None else _store_of(dirs[0]["xproj"])
    for k, d in enumerate(dirs):
        dt = torch.bfloat16 if xin is not None else d["xproj"].dtype
        xp, ldxp = None, 0
        if xin is None:
            assert _store_of(d["xproj"]) == store
            xp, _, _, ldxp = as2d(d["xproj"], dt)
        op, ldo, o16, ldo16 = None, 0, None, 0
        if d.get("out") is not None:
            op, _, _, ldo = as2d(d["out"])
        if d.get("out16") is not None:
            o16, _, _, ldo16 = as2d(d["out16"], torch.bfloat16)
        assert op is not None or o16 is not None
        gp, ldg, cp, ldc = None, 0, None, 0
        if d.get("gates") is not None:
            gp, _, _, ldg = as2d(d["gates"], dt)
            if d.get("cstate") is not None:
                cp, _, _, ldc = as2d(d["cstate"])
        arr[k] = RnnDir(xp, ldxp, d["w_packed"].data_ptr(), _ptr(d.get("b_hh")), op, ldo, gp, ldg, cp, ldc,
                        int(d["reverse"]), 0, o16, ldo16)
    G, nd, esz = (4 if cell == CELL_LSTM else 3), len(dirs), (2 if xin is not None else dirs[0]["xproj"].element_size())
    train = dirs[0].get("gates") is not None
    h_bytes = (4 if dirs[0].get("out") is not None else 0) + (2 if dirs[0].get("out16") is not None else 0)
    per_elem = (0 if xin is not None else 4 * esz) + h_bytes + ((4 * esz + (4 if cell == CELL_LSTM else 0)) if train else 0)     # xproj in, h out, stash out
    cost = (2.0 * B * T * nd * G * H * H, float(B * T * nd * H * per_elem))
    if xin is not None:
        kin = int(xarr[0].kin)
        cost = (cost[0] + 2.0 * B * T * nd * G * H * kin, cost[1] + float(B * T * nd * kin * 2))
        assert workspace.dtype == torch.uint8 and workspace.is_contiguous()
        _RNN_WS[id(workspace)] = weakref.ref(workspace)
        _call("tad_rnn_fwd_ws_x", cell, prec, store, B, T, H, nd, arr, xarr, workspace.data_ptr(), workspace.numel(), _stream(), cost=cost)
    elif workspace is not None:
        assert workspace.dtype == torch.uint8 and workspace.is_contiguous()
        _RNN_WS[id(workspace)] = weakref.ref(workspace)
        _call("tad_rnn_fwd_ws", cell, prec, store, B, T, H, nd, arr, workspace.data_ptr(), workspace.numel(), _stream(), cost=cost)
    else:
        _call("tad_rnn_fwd", cell, prec, store, B, T, H, nd, arr, _stream(), cost=cost)


def rnn_bwd(cell, prec, B, T, H, dirs):
    """dirs: list of dict(dout, gates, cstate (LSTM), out (GRU), w_packed_t, dgates, reverse)."""
    arr = (RnnBwdDir * len(dirs))()
    store = _store_of(dirs[0]["gates"])
    for k, d in enumerate(dirs):
        dt = d["gates"].dtype
        assert _store_of(d["gates"]) == store and d["dgates"].dtype == dt
        dp, _, _, ldd = as2d(d["dout"])
        sp, _, _, lds = as2d(d["gates"], dt)
        cp, ldc, op, ldo = None, 0, None, 0
        if d.get("cstate") is not None:
            cp, _, _, ldc = as2d(d["cstate"])
        if d.get("out") is not None:
            op, _, _, ldo = as2d(d["out"])
        gp, _, _, ldg = as2d(d["dgates"], dt)
        arr[k] = RnnBwdDir(dp, ldd, sp, lds, cp, ldc, op, ldo, d["w_packed_t"].data_ptr(), gp, ldg, int(d["reverse"]), 0)
    G, nd, esz = (4 if cell == CELL_LSTM else 3), len(dirs), dirs[0]["gates"].element_size()
    per_elem = 4 + 4 * esz + 4 + 4 * esz                                    # dout, stash, c or h_prev, dgates out
    _call("tad_rnn_bwd", cell, prec, store, B, T, H, nd, arr, _stream(),
          cost=(2.0 * B * T * nd * G * H * H, float(B * T * nd * H * per_elem)))


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


def l1_loss(pred, target, grad_scale, dpred, loss_sum):
    """loss_sum[0] += sum |pred - target|; dpred = grad_scale * sign(pred - target) (all contiguous float32)."""
    assert pred.is_contiguous() and target.is_contiguous() and pred.numel() == target.numel()
    assert dpred is None or (dpred.is_contiguous() and dpred.numel() == pred.numel())
    _call("tad_l1_loss", pred.numel(), _ptr(pred), _ptr(target), float(grad_scale), _ptr(dpred), _ptr(loss_sum), _stream())


def grad_norm_sq(g, out_sum):
    """out_sum[0] += sum of squares of the contiguous float32 buffer g (one launch)."""
    assert g.is_contiguous() and g.dtype == torch.float32 and out_sum.dtype == torch.float32
    _call("tad_grad_norm_sq", g.numel(), _ptr(g), _ptr(out_sum), _stream(), cost=(2.0 * g.numel(), 4.0 * g.numel()))


def sincos_features(x_nums, out):
    xp, rows, n, ldx = as2d(x_nums)
    op, rows2, cols, ldo = as2d(out, out.dtype)
    assert rows == rows2 and cols == 16 * n
    _call("tad_sincos_features" + _sfx(out), rows, n, xp, ldx, op, ldo, _stream())


def _i32(lst):
    return (C.c_int32 * len(lst))(*lst)


def embed_fwd(x_disc, tables, tab_off, cards, E, out):
    xp, rows, nd, ldx = as2d(x_disc)
    op, rows2, cols, ldo = as2d(out, out.dtype)
    assert rows == rows2 and cols == nd * E and len(cards) == nd
    _call("tad_embed_fwd" + _sfx(out), rows, nd, E, xp, ldx, tables.data_ptr(), _i32(tab_off), _i32(cards), op, ldo, _stream())


def embed_bwd(x_disc, dout, dtables, tab_off, cards, E):
    xp, rows, nd, ldx = as2d(x_disc)
    dp, rows2, cols, ldo = as2d(dout)
    assert rows == rows2 and cols == nd * E
    _call("tad_embed_bwd", rows, nd, E, xp, ldx, dp, ldo, dtables.data_ptr(), _i32(tab_off), _i32(cards), _stream())


def posenc_fwd(t, t_stride, rows, H, inv_max, w, b, scale, shift, out):
    op, rows2, cols, ldo = as2d(out, out.dtype)
    assert rows == rows2 and cols == H
    _call("tad_posenc_fwd" + _sfx(out), rows, H, t.data_ptr(), t_stride, float(inv_max), w.data_ptr(), b.data_ptr(), scale.data_ptr(),
          shift.data_ptr(), op, ldo, _stream())


def silu_fwd(z, a):
    assert z.is_contiguous() and a.is_contiguous()
    _call("tad_silu_fwd" + _sfx(a), z.numel(), _ptr(z), _ptr(a), _stream())


def silu_bwd(z, da, dz):
    assert z.is_contiguous() and da.is_contiguous() and dz.is_contiguous()
    _call("tad_silu_bwd" + _sfx(dz), z.numel(), _ptr(z), _ptr(da), _ptr(dz), _stream())


def layernorm_fwd(x, gamma, beta, y, mean_rstd, eps=1e-5):
    xp, rows, D, ldx = as2d(x)
    yp, _, _, ldy = as2d(y)
    _call("tad_layernorm_fwd", rows, D, xp, ldx, gamma.data_ptr(), beta.data_ptr(), float(eps), yp, ldy, _ptr(mean_rstd), _stream())


def ln_fc_supported(D, K):
    return D <= LN_FC_MAX_D and K <= 8


def ln_fc_fwd(prec, x, gamma, beta, fc_w, fc_b, mean_rstd, pred, y=None, eps=1e-5):
    """pred = fc(LayerNorm(x)) in one launch (K = fc_w.shape[0] <= 8 outputs per row); y receives the normalised rows
    only if given."""
    xp, rows, D, ldx = as2d(x)
    K = fc_w.shape[0]
    assert fc_w.shape == (K, D) and fc_w.is_contiguous() and pred.is_contiguous() and pred.numel() == rows * K
    yp, ldy = (None, 0) if y is None else (as2d(y)[0], as2d(y)[3])
    _call("tad_ln_fc_fwd", prec, rows, D, K, xp, ldx, _ptr(gamma), _ptr(beta), float(eps), _ptr(fc_w), _ptr(fc_b), yp, ldy,
          _ptr(mean_rstd), _ptr(pred), _stream(), cost=(2.0 * rows * D * K, 4.0 * rows * (D + K + 2)))


def ln_fc_bwd(prec, dpred, x, gamma, beta, mean_rstd, fc_w, dx, dgamma, dbeta, dfc_w, dfc_b):
    """Backward of ln_fc_fwd in one launch: dx, and += into dgamma, dbeta, dfc_w, dfc_b."""
    xp, rows, D, ldx = as2d(x)
    dxp, rows2, D2, lddx = as2d(dx)
    K = fc_w.shape[0]
    assert rows == rows2 and D == D2 and dpred.is_contiguous() and dpred.numel() == rows * K and dfc_w.is_contiguous()
    _call("tad_ln_fc_bwd", prec, rows, D, K, _ptr(dpred), xp, ldx, _ptr(gamma), _ptr(beta), _ptr(mean_rstd), _ptr(fc_w), dxp, lddx,
          _ptr(dgamma), _ptr(dbeta), _ptr(dfc_w), _ptr(dfc_b), _stream(), cost=(4.0 * rows * D * K, 4.0 * rows * (2 * D + K + 2)))


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


def draw_batch(seed, counter, N, B, T, K, n_steps, idx=None, t=None, i=None, noise=None):
    """The draws of one training iteration on the device (reference DIFF.py:306, 317, 253, 108; tad_draw_batch):
    idx [B] int64 distinct window indices in [0, N), t [B, T(, 1)] float32 ascending uniforms, i [B] int32 diffusion
    steps, noise [B, T, K] float32 normals -- functions of (seed, counter) alone.  Outputs left None are not drawn."""
    for a, dt, n in ((idx, torch.int64, B), (t, torch.float32, B * T), (i, torch.int32, B), (noise, torch.float32, B * T * K)):
        assert a is None or (a.is_cuda and a.is_contiguous() and a.dtype == dt and a.numel() == n), (dt, n)
    _call("tad_draw_batch", int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), N, B, T, K, n_steps, _ptr(idx), _ptr(t), _ptr(i), _ptr(noise),
          _stream())


def sampler_noise(seed, counter, row0, rows, K, out):
    """out [rows, K] float32 = the standard normals of rows row0 .. row0+rows of the global latent tensor at step
    ``counter`` (tad_sampler_noise: a function of (seed, counter, global row, k) alone)."""
    assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == rows * K
    _call("tad_sampler_noise", int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), int(row0), int(rows), int(K), _ptr(out), _stream())


def zero_(t):
    """t[...] = 0 for a contiguous tensor or a 2-D view with unit column stride (float32 / bfloat16 / int32)."""
    if t.numel() == 0:
        return t
    _chk_dev(t)
    esz = t.element_size()
    if t.is_contiguous():
        rows, row_bytes, ld = 1, t.numel() * esz, t.numel() * esz
    else:
        assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
        rows, row_bytes, ld = t.shape[0], t.shape[1] * esz, t.stride(0) * esz
    _call("tad_zero_2d", rows, row_bytes, t.data_ptr(), ld, _stream(), cost=(0.0, float(rows * row_bytes)))
    return t


def fill_(t, value):
    """t[...] = value for a contiguous float32 tensor (tad_fill_f32)."""
    assert t.is_contiguous() and t.dtype == torch.float32
    if t.numel():
        _call("tad_fill_f32", t.numel(), _ptr(t), float(value), _stream(), cost=(0.0, 4.0 * t.numel()))
    return t


def sigmoid_(t):
    """t = 1 / (1 + exp(-t)) in place for a contiguous float32 tensor (tad_sigmoid_f32; torch.sigmoid's fp32 formula)."""
    assert t.is_contiguous() and t.dtype == torch.float32
    if t.numel():
        _call("tad_sigmoid_f32", t.numel(), _ptr(t), _stream(), cost=(0.0, 8.0 * t.numel()))
    return t


def copy_2d(src, dst):
    """dst[:, :cols] = src (float32, unit column stride)."""
    sp, rows, cols, lds_ = as2d(src)
    dp, rows2, cols2, ldd = as2d(dst)
    assert rows == rows2 and cols == cols2
    _call("tad_copy_2d", rows, cols, sp, lds_, dp, ldd, _stream(), cost=(0.0, 8.0 * rows * cols))


def add_vec(a, b, out):
    """out = a + b (contiguous float32 vectors)."""
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and a.numel() == b.numel() == out.numel()
    assert a.dtype == b.dtype == out.dtype == torch.float32
    _call("tad_add_vec", a.numel(), _ptr(a), _ptr(b), _ptr(out), _stream())


def gather_windows(table, idx, T, stride, out):
    """out[b, t, :] = table[idx[b] * stride + t, :]: a minibatch of sliding windows cut on the fly from the
    [rows, F] table (reference dataprovider_pypots.py:75-87 + the trainers' index gather); idx int64 on device."""
    assert table.dim() == 2 and table.stride(1) == 1 and out.is_contiguous() and idx.dtype == torch.int64 and idx.is_cuda
    B, F = idx.numel(), table.shape[1]
    assert out.numel() == B * T * F
    _call("tad_gather_windows", B, T, F, stride, table.data_ptr(), table.stride(0), idx.data_ptr(), out.data_ptr(), _stream())


def metadata_noise(cond, cols, noise, level, out):
    """out = cond with level * (column std + 1e-8) * noise added on the columns ``cols`` (reference
    dataprovider_pypots.py:149-180).  cond/out [..., F] float32, noise [rows, len(cols)] float32."""
    cp, rows, F, ld = as2d(cond)
    op, rows2, F2, ldo = as2d(out)
    n = len(cols)
    assert rows == rows2 and F == F2 and noise.is_contiguous() and noise.numel() == rows * n and noise.dtype == torch.float32
    stats = torch.empty(2 * max(n, 1), dtype=torch.float64, device=cond.device)
    _call("tad_metadata_noise", rows, F, n, _i32(cols), cp, ld, _ptr(noise), float(level), stats.data_ptr(), op, ldo, _stream())


def flip_categories(cond, cols, classes, rnd, base, p, out):
    """out = cond with the codes of the columns ``cols`` flipped to another class where rnd < p (reference
    dataprovider_pypots.py:182-229).  rnd [len(cols), rows] float32, base [len(cols), rows] int64."""
    cp, rows, F, ld = as2d(cond)
    op, rows2, F2, ldo = as2d(out)
    n = len(cols)
    assert rows == rows2 and F == F2 and len(classes) == n
    assert rnd.is_contiguous() and rnd.dtype == torch.float32 and rnd.numel() == n * rows
    assert base.is_contiguous() and base.dtype == torch.int64 and base.numel() == n * rows
    _call("tad_flip_categories", rows, F, n, _i32(cols), _i32(classes), cp, ld, _ptr(rnd), _ptr(base), float(p), op, ldo, _stream())


def sampler_tail_supported(D, K, H):
    nc = -(-D // 64)
    nc = 2 if nc <= 2 else 8 if nc <= 8 else 16
    return D <= 1024 and K <= 8 and H <= 1024 and 10 * H + 3 + 640 * nc <= 16384


def sampler_tail(prec, h, gamma, beta, fc_w, fc_b, x, z, c1, c2, c3, last, w_in, b_in, xh, eps=1e-5):
    """LayerNorm(h) -> fc -> sampler update of x (in place, with noise z) -> xh = input_proj(updated x), one launch
    (reference DIFF.py:221-223, 348-351, 198)."""
    hp, rows, D, ldh = as2d(h)
    xp, rows2, H, ldxh = as2d(xh, xh.dtype)
    K = fc_w.shape[0]
    assert rows == rows2 and fc_w.shape == (K, D) and w_in.shape == (H, K) and fc_w.is_contiguous() and w_in.is_contiguous()
    assert x.is_contiguous() and z.is_contiguous() and x.numel() == rows * K and z.numel() == rows * K
    _call("tad_sampler_tail" + _sfx(xh), prec, rows, D, K, H, hp, ldh, _ptr(gamma), _ptr(beta), float(eps), _ptr(fc_w), _ptr(fc_b), _ptr(x),
          _ptr(z), float(c1), float(c2), float(c3), int(bool(last)), _ptr(w_in), _ptr(b_in), xp, ldxh, _stream())


def sampler_tail_proj(prec, h, gamma, beta, fc_w, fc_b, x, z, c1, c2, c3, last, w_fold, b_fold, vrow, pinv, pa16, eps=1e-5):
    """sampler_tail with the next step's x share of `proj` folded in: pa16 = relu(x_new w_fold^T + b_fold + vrow + pinv)
    as bfloat16 (reference DIFF.py:221-223, 348-351, then 203 and 218 of the next reverse step)."""
    hp, rows, D, ldh = as2d(h)
    pp, rows2, H, ldpa = as2d(pa16, torch.bfloat16)
    ip, rows3, H3, ldp = as2d(pinv)
    K = fc_w.shape[0]
    assert rows == rows2 == rows3 and H3 == H and fc_w.shape == (K, D) and w_fold.shape == (H, K) and fc_w.is_contiguous()
    assert w_fold.is_contiguous() and b_fold.numel() == H and vrow.numel() == H and vrow.is_contiguous()
    assert x.is_contiguous() and z.is_contiguous() and x.numel() == rows * K and z.numel() == rows * K
    _call("tad_sampler_tail_proj_b16", prec, rows, D, K, H, hp, ldh, _ptr(gamma), _ptr(beta), float(eps), _ptr(fc_w), _ptr(fc_b),
          _ptr(x), _ptr(z), float(c1), float(c2), float(c3), int(bool(last)), _ptr(w_fold), _ptr(b_fold), _ptr(vrow), ip, ldp, pp, ldpa,
          _stream())


def sampler_tail_proj_cn(prec, h, gamma, beta, fc_w, fc_b, x, seed, counter, row0, c1, c2, c3, last, w_fold, b_fold, vrow, pinv, pa16, eps=1e-5):
    """sampler_tail_proj that draws the step's noise inside the launch: the values ``sampler_noise(seed, counter, row0, ...)``
    writes for these rows, without the tensor and without its launch."""
    hp, rows, D, ldh = as2d(h)
    pp, rows2, H, ldpa = as2d(pa16, torch.bfloat16)
    ip, rows3, H3, ldp = as2d(pinv)
    K = fc_w.shape[0]
    assert rows == rows2 == rows3 and H3 == H and fc_w.shape == (K, D) and w_fold.shape == (H, K) and fc_w.is_contiguous()
    assert w_fold.is_contiguous() and b_fold.numel() == H and vrow.numel() == H and vrow.is_contiguous()
    assert x.is_contiguous() and x.numel() == rows * K
    _call("tad_sampler_tail_proj_cn_b16", prec, rows, D, K, H, hp, ldh, _ptr(gamma), _ptr(beta), float(eps), _ptr(fc_w), _ptr(fc_b),
          _ptr(x), int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), int(row0), float(c1), float(c2), float(c3), int(bool(last)),
          _ptr(w_fold), _ptr(b_fold), _ptr(vrow), ip, ldp, pp, ldpa, _stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    _call("tad_adam_step", n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), float(lr), float(beta1), float(beta2),
          float(eps), float(weight_decay), int(step), _stream())


def adam_coefficients(lr, beta1, beta2, step):
    """(lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) as float32, computed the way tad_adam_step does."""
    out = (C.c_float * 2)()
    _lib.load().tad_adam_coefficients(float(lr), float(beta1), float(beta2), int(step), out)
    return float(out[0]), float(out[1])


def adam_tick(coef4, lr, beta1, beta2):
    """Advance the device-side step counter of a captured optimizer step (coef4[2], an int32 stored in the float32
    buffer) and refresh the two factors tad_adam_step_dev reads from coef4[0:2] -- one single-thread launch that a HIP
    graph captures in front of the Adam launch, so that a replay needs no host-to-device copy."""
    assert coef4.numel() >= 4 and coef4.dtype == torch.float32 and coef4.is_contiguous()
    _call("tad_adam_tick", float(lr), float(beta1), float(beta2), _ptr(coef4), _stream())


def adam_step_dev(p, g, m, v, beta1, beta2, eps, weight_decay, coef2):
    """Adam step whose step-dependent factors come from the device tensor coef2 ([0:2]; graph replays)."""
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and coef2.numel() >= 2 and coef2.dtype == torch.float32
    _call("tad_adam_step_dev", n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), float(beta1), float(beta2), float(eps),
          float(weight_decay), _ptr(coef2), _stream())


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


def heads_to_tensor(heads, nb, nc, nn, cards, out):
    """Decoder head outputs [rows, NH] (bins | cats | nums) -> table rows [rows, F] (thresholded bins, argmax codes, nums)."""
    hp, rows, NH, ldh = as2d(heads)
    op, rows2, F, ldo = as2d(out)
    assert rows == rows2 and F == nb + nc + nn and NH == nb + sum(cards[:nc]) + nn, (heads.shape, out.shape)
    _call("tad_heads_to_tensor", rows, nb, nc, nn, _i32(list(cards)) if nc else None, hp, ldh, op, ldo, _stream(),
          cost=(0.0, 4.0 * rows * (NH + F)))
