"""Parity of every C-ABI kernel family against the numpy oracle on seeded inputs (GPU only).

fp32 MFMA path: relative error <= 1e-4 (north-star tolerance; observed ~1e-6).
bf16 MFMA path: compared against the oracle evaluated with both operands of every matrix
product rounded to bf16 (the kernels' rounding points): GEMMs <= 1e-4, recurrences
<= 5e-3 (an accumulation-order difference can flip a bf16 rounding of h)."""
import contextlib

import numpy as np
import pytest
import torch

import fixture_util as fu
from oracle import ops as oops
from oracle import rnn as ornn

pytestmark = pytest.mark.gpu

F32_TOL = 1e-4
BF16_EMU_TOL = 5e-3      # vs the oracle with operands rounded to bf16 at the kernels' rounding points


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(lib):
    from timeautodiff_amd import ops as _ops
    return _ops


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16().float().numpy()


def D(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ linear family
SHAPES = [(5, 7, 3), (130, 70, 33), (257, 200, 6), (100, 800, 200), (64, 6, 400), (33, 100, 1), (300, 128, 448)]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_fwd(ops, dev, prec, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    ldx, ldy = K + 3, N + 5
    X = rnd(rng, M, ldx)
    W = rnd(rng, N, K, scale=0.3)
    b, b2 = rnd(rng, N), rnd(rng, N)
    R = rnd(rng, M, N)
    Xd, Wd, Yd = D(X, dev), D(W, dev), torch.zeros(M, ldy, device=dev)
    Xr, Wr = (bf16_round(X[:, :K]), bf16_round(W)) if prec else (X[:, :K], W)
    ref = Xr.astype(np.float64) @ Wr.astype(np.float64).T
    # plain
    ops.linear_fwd(prec, Xd[:, :K], Wd, None, Yd[:, :N])
    assert fu.rel_err(Yd[:, :N].cpu().numpy(), ref) < F32_TOL
    assert float(Yd[:, N:].abs().max()) == 0.0          # nothing written outside the slice
    # bias + bias2 + residual + relu
    ops.linear_fwd(prec, Xd[:, :K], Wd, D(b, dev), Yd[:, :N], act=ops.ACT_RELU, bias2=D(b2, dev), residual=D(R, dev))
    assert fu.rel_err(Yd[:, :N].cpu().numpy(), np.maximum(ref + b + b2 + R, 0)) < F32_TOL
    # accumulate into y
    Yd.fill_(1.5)
    ops.linear_fwd(prec, Xd[:, :K], Wd, None, Yd[:, :N], accumulate=True)
    assert fu.rel_err(Yd[:, :N].cpu().numpy(), ref + 1.5) < F32_TOL


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_bwd_data(ops, dev, prec, M, N, K):
    rng = np.random.default_rng(M + N + K)
    dY = rnd(rng, M, N)
    W = rnd(rng, N, K, scale=0.3)
    act = rnd(rng, M, K)
    dYr, Wr = (bf16_round(dY), bf16_round(W)) if prec else (dY, W)
    ref = dYr.astype(np.float64) @ Wr.astype(np.float64)
    dX = torch.zeros(M, K + 2, device=dev)
    ops.linear_bwd_data(prec, D(dY, dev), D(W, dev), dX[:, :K])
    assert fu.rel_err(dX[:, :K].cpu().numpy(), ref) < F32_TOL
    ops.linear_bwd_data(prec, D(dY, dev), D(W, dev), dX[:, :K], relu_of=D(act, dev))
    assert fu.rel_err(dX[:, :K].cpu().numpy(), ref * (act > 0)) < F32_TOL
    dX.fill_(2.0)
    ops.linear_bwd_data(prec, D(dY, dev), D(W, dev), dX[:, :K], accumulate=True)
    assert fu.rel_err(dX[:, :K].cpu().numpy(), ref + 2.0) < F32_TOL


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", SHAPES + [(4096, 96, 40), (8192, 8, 400), (8191, 200, 6), (3, 2, 1100)])
def test_linear_bwd_weight(ops, dev, prec, M, N, K):
    rng = np.random.default_rng(M + 7 * N + K)
    dY = rnd(rng, M, N)
    X = rnd(rng, M, K)
    dYr, Xr = (bf16_round(dY), bf16_round(X)) if prec else (dY, X)
    ref = dYr.astype(np.float64).T @ Xr.astype(np.float64)
    dW = torch.full((N, K), 0.25, device=dev)
    ops.linear_bwd_weight(prec, D(dY, dev), D(X, dev), dW)
    assert fu.rel_err(dW.cpu().numpy(), ref + 0.25) < F32_TOL
    # bias gradients in the same pass: unrounded fp32 column sums of dY in both precisions
    dW.fill_(0.25)
    db, db2 = torch.full((N,), 0.5, device=dev), torch.full((N,), -1.0, device=dev)
    ops.linear_bwd_weight(prec, D(dY, dev), D(X, dev), dW, db=db, db2=db2)
    cs = dY.astype(np.float64).sum(0)
    assert fu.rel_err(dW.cpu().numpy(), ref + 0.25) < F32_TOL
    assert fu.rel_err(db.cpu().numpy(), cs + 0.5) < F32_TOL
    assert fu.rel_err(db2.cpu().numpy(), cs - 1.0) < F32_TOL


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("shift", [-1, 1])
def test_linear_bwd_weight_timeshift(ops, dev, prec, shift):
    rng = np.random.default_rng(5)
    B, T, N, K = 7, 6, 72, 24
    dY = rnd(rng, B, T, N)
    X = rnd(rng, B, T, K)
    Xs = np.zeros_like(X)
    if shift == -1:
        Xs[:, 1:] = X[:, :-1]
    else:
        Xs[:, :-1] = X[:, 1:]
    dYr, Xr = (bf16_round(dY), bf16_round(Xs)) if prec else (dY, Xs)
    ref = dYr.reshape(B * T, N).astype(np.float64).T @ Xr.reshape(B * T, K).astype(np.float64)
    dW = torch.zeros(N, K, device=dev)
    ops.linear_bwd_weight(prec, D(dY, dev).view(B * T, N), D(X, dev).view(B * T, K), dW, T=T, tshift=shift)
    assert fu.rel_err(dW.cpu().numpy(), ref) < F32_TOL


def test_colsum(ops, dev):
    rng = np.random.default_rng(9)
    M, N = 1000, 150
    dY = rnd(rng, M, N + 4)
    s = rnd(rng, M)
    o1 = torch.zeros(N, device=dev)
    o2 = torch.ones(N, device=dev)
    ow = torch.zeros(N, device=dev)
    ops.colsum(D(dY, dev)[:, :N], out_sum=o1, out_sum2=o2, row_scale=D(s, dev), scale_mul=0.01, out_wsum=ow)
    ref = dY[:, :N].astype(np.float64).sum(0)
    assert fu.rel_err(o1.cpu().numpy(), ref) < F32_TOL
    assert fu.rel_err(o2.cpu().numpy(), ref + 1) < F32_TOL
    assert fu.rel_err(ow.cpu().numpy(), (dY[:, :N].astype(np.float64) * s[:, None]).sum(0) * 0.01) < F32_TOL


@pytest.mark.parametrize("M,N,K", [(5, 8, 8), (130, 72, 40), (257, 200, 16), (100, 800, 200), (300, 128, 448), (4100, 1600, 400)])
def test_linear_b16_family(ops, dev, M, N, K):
    """The three products with bf16-stored operands: exact products of the stored bf16 values, fp32 accumulation."""
    rng = np.random.default_rng(M + 3 * N + K)
    X = rnd(rng, M, K)
    W16 = torch.from_numpy(rnd(rng, N, K, scale=0.3)).bfloat16()
    b = rnd(rng, N)
    Xd, Wd = D(X, dev), W16.to(dev)
    Wf = W16.float().numpy().astype(np.float64)
    ref = bf16_round(X).astype(np.float64) @ Wf.T + b
    # forward: fp32 result, then bf16 result (+ relu)
    Y = torch.zeros(M, N + 8, device=dev)
    ops.linear_fwd_b16(Xd, Wd, D(b, dev), Y[:, :N])
    assert fu.rel_err(Y[:, :N].cpu().numpy(), ref) < F32_TOL
    assert float(Y[:, N:].abs().max()) == 0.0
    Y16 = torch.zeros(M, N + 8, dtype=torch.bfloat16, device=dev)
    ops.linear_fwd_b16(Xd, Wd, D(b, dev), Y16[:, :N], act=ops.ACT_RELU)
    got, want = Y16[:, :N].float().cpu().numpy(), np.maximum(ref, 0)
    # one bf16 rounding of the result: at most half a unit in the last of 8 significant bits (2^-8 relative)
    # (plus the fp32 accumulation noise of the unrounded sum, which matters where the terms cancel)
    assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -8 + 1e-5 * np.abs(want).max())
    assert float(Y16[:, N:].float().abs().max()) == 0.0
    # the same forward with the input stored as bf16 too (a strided weight view, as the engine passes its padded images)
    X16 = torch.from_numpy(X).bfloat16().to(dev)
    Wpad = torch.zeros(N + 3, K + 8, dtype=torch.bfloat16, device=dev)
    Wpad[:N, :K] = Wd
    Yx = torch.zeros(M, N + 8, device=dev)
    ops.linear_fwd_b16x(X16, Wpad[:N, :K], D(b, dev), Yx[:, :N])
    assert fu.rel_err(Yx[:, :N].cpu().numpy(), ref) < F32_TOL and float(Yx[:, N:].abs().max()) == 0.0
    Yx16 = torch.zeros(M, N + 8, dtype=torch.bfloat16, device=dev)
    ops.linear_fwd_b16x(X16, Wpad[:N, :K], D(b, dev), Yx16[:, :N], act=ops.ACT_RELU)
    assert torch.equal(Yx16, Y16)
    # backward wrt the input: bf16 gradient x bf16 weights
    dY16 = torch.from_numpy(rnd(rng, M, N)).bfloat16().to(dev)
    dYf = dY16.float().cpu().numpy().astype(np.float64)
    act = rnd(rng, M, K)
    dX = torch.zeros(M, K + 4, device=dev)
    ops.linear_bwd_data_b16(dY16, Wd, dX[:, :K], relu_of=D(act, dev))
    assert fu.rel_err(dX[:, :K].cpu().numpy(), (dYf @ Wf) * (act > 0)) < F32_TOL
    dX.fill_(2.0)
    ops.linear_bwd_data_b16(dY16, Wd, dX[:, :K], accumulate=True)
    assert fu.rel_err(dX[:, :K].cpu().numpy(), dYf @ Wf + 2.0) < F32_TOL
    assert float((dX[:, K:] - 2.0).abs().max()) == 0.0
    # backward wrt the weight: bf16 gradient x fp32 activations, bias sums over the bf16 values
    dW = torch.full((N, K), 0.25, device=dev)
    db, db2 = torch.full((N,), 0.5, device=dev), torch.full((N,), -1.0, device=dev)
    ops.linear_bwd_weight(1, dY16, Xd, dW, db=db, db2=db2)
    assert fu.rel_err(dW.cpu().numpy(), dYf.T @ bf16_round(X).astype(np.float64) + 0.25) < F32_TOL
    assert fu.rel_err(db.cpu().numpy(), dYf.sum(0) + 0.5) < F32_TOL
    assert fu.rel_err(db2.cpu().numpy(), dYf.sum(0) - 1.0) < F32_TOL


@pytest.mark.parametrize("b16", [False, True])
@pytest.mark.parametrize("G,slots", [(4, (0, 1, 2, 3)), (3, (0, 1, 2, -1)), (3, (0, 1, -1, 2))])
def test_linear_bwd_weight_gate_map(ops, dev, b16, G, slots):
    """Gate-interleaved gradient columns land in torch's gate-major rows (dropped slots contribute nothing)."""
    rng = np.random.default_rng(17 + G)
    M, H, K, nblk = 520, 50, 24, 2
    dY = rnd(rng, M, nblk * 4 * H)
    X = rnd(rng, M, K)
    dYd = D(dY, dev).bfloat16() if b16 else D(dY, dev)
    prec = 1 if b16 else 0
    dYr = dYd.float().cpu().numpy().astype(np.float64)
    Xr = (bf16_round(X) if b16 else X).astype(np.float64)
    full = dYr.T @ Xr                                    # [nblk*4H, K], rows blk*4H + 4u + s
    ref = np.zeros((nblk * G * H, K))
    refb = np.zeros(nblk * G * H)
    for blk in range(nblk):
        for s_, g in enumerate(slots):
            if g < 0:
                continue
            src = blk * 4 * H + 4 * np.arange(H) + s_
            ref[blk * G * H + g * H + np.arange(H)] = full[src]
            refb[blk * G * H + g * H + np.arange(H)] = dYr[:, src].sum(0)
    dW = torch.zeros(nblk * G * H, K, device=dev)
    db = torch.zeros(nblk * G * H, device=dev)
    ops.linear_bwd_weight(prec, dYd, D(X, dev), dW, db=db, gate_map=(H, G, slots))
    assert fu.rel_err(dW.cpu().numpy(), ref) < F32_TOL
    assert fu.rel_err(db.cpu().numpy(), refb) < F32_TOL


# ------------------------------------------------------------------ recurrent family
IN_SLOTS = {0: (0, 1, 2, 3), 1: (0, 1, 2, -1)}
HID_SLOTS = {0: (0, 1, 2, 3), 1: (0, 1, -1, 2)}


def _run_rnn(ops, dev, cell, prec, B, T, H, rev_flags, rng, IN=11, cluster=False, xin=False):
    """One layer, len(rev_flags) directions sharing the input, through the C ABI exactly as the engine drives
    it: pack -> input GEMM -> sequence kernel -> BPTT kernel -> weight / bias / input gradients.  bf16 storage
    of the per-timestep tensors is used when the shapes allow it (H even, IN a multiple of 8).
    Returns device results (numpy, torch's gate-major order) and the oracle's."""
    G = 4 if cell == 0 else 3
    nd = len(rev_flags)
    s16 = bool(prec) and H % 2 == 0 and IN % 8 == 0
    sdt = torch.bfloat16 if s16 else torch.float32
    BT, H4, GH = B * T, 4 * H, G * H
    x = rnd(rng, B, T, IN)
    dout = rnd(rng, B, T, nd * H)
    k = 1.0 / np.sqrt(H)
    Wih = rng.uniform(-k, k, (nd, GH, IN)).astype(np.float32)
    Whh = rng.uniform(-k, k, (nd, GH, H)).astype(np.float32)
    bih = rng.uniform(-k, k, (nd, GH)).astype(np.float32)
    bhh = rng.uniform(-k, k, (nd, GH)).astype(np.float32)
    oracle = []
    fwd, bwd = (ornn.lstm_dir_fwd, ornn.lstm_dir_bwd) if cell == 0 else (ornn.gru_dir_fwd, ornn.gru_dir_bwd)
    with (oops.bf16_operands(storage=True) if prec else contextlib.nullcontext()), \
            (oops.rnn_xproj_on_chip() if xin else contextlib.nullcontext()):
        for d, rev in enumerate(rev_flags):
            f64 = lambda a: a.astype(np.float64)
            hs, cache = fwd(f64(x), f64(Wih[d]), f64(Whh[d]), f64(bih[d]), f64(bhh[d]), rev)
            dx, dWih, dWhh, dbih, dbhh = bwd(f64(Wih[d]), f64(Whh[d]), cache, f64(dout[..., d * H:(d + 1) * H]))
            oracle.append(dict(hs=hs, cache=cache, dWih=dWih, dWhh=dWhh, dbih=dbih, dbhh=dbhh, dx=dx))
    # device
    xd, doutd = D(x, dev).view(BT, IN), D(dout, dev).view(BT, nd * H)
    Wihd, Whhd, bihd, bhhd = D(Wih, dev), D(Whh, dev), D(bih, dev), D(bhh, dev)
    # (xin: the sequence kernel reads 32 columns of the image whatever the input width)
    w_gl = torch.empty(nd * H4, 128 if xin else IN, dtype=sdt, device=dev)
    bias_gl = torch.empty(nd * H4, device=dev)
    ops.rnn_pack_input(cell, H, Wihd.view(nd * GH, IN), bihd.view(-1), bhhd.view(-1), w_gl, bias_gl)
    w_gl = w_gl[:, :IN]
    xproj = torch.zeros(BT, nd * H4, dtype=sdt, device=dev)
    if s16:
        ops.linear_fwd_b16(xd, w_gl, bias_gl, xproj)
    else:
        ops.linear_fwd(prec, xd, w_gl, bias_gl, xproj)
    out = torch.zeros(BT, nd * H, device=dev)
    gates = torch.zeros(BT, nd * H4, dtype=sdt, device=dev)
    cst = torch.zeros(BT, nd * H, device=dev)
    dg = torch.zeros(BT, nd * H4, dtype=sdt, device=dev)
    nbytes = ops.rnn_packed_bytes(prec, H, G)
    pf = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
    pb = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
    fdirs, bdirs = [], []
    for d, rev in enumerate(rev_flags):
        ops.rnn_pack_weights(prec, H, G, Whhd[d], pf[d], pb[d])
        sl, hl = slice(d * H4, (d + 1) * H4), slice(d * H, (d + 1) * H)
        fdirs.append(dict(xproj=xproj[:, sl], w_packed=pf[d], b_hh=bhhd[d] if cell == 1 else None, out=out[:, hl],
                          gates=gates[:, sl], cstate=cst[:, hl] if cell == 0 else None, reverse=rev))
        bdirs.append(dict(dout=doutd[:, hl], gates=gates[:, sl], cstate=cst[:, hl] if cell == 0 else None,
                          out=out[:, hl] if cell == 1 else None, w_packed_t=pb[d], dgates=dg[:, sl], reverse=rev))
    fwd_kernel = None
    if cluster:
        # the form that splits a batch tile's hidden units over eight workgroups: needs device memory for the exchange
        nws = ops.rnn_workspace_bytes(cell, prec, 1 if s16 else 0, B, T, H, nd)
        assert nws > 0
        wsp = torch.zeros(nws, dtype=torch.uint8, device=dev)
        xi = None
        if xin:
            # the layer's input as bf16 (a padded row stride), W_ih and the bias from the gate-interleaved images; no xproj
            x16 = torch.zeros(BT, IN + 8, dtype=torch.bfloat16, device=dev)[:, :IN]
            x16.copy_(xd)
            xi = [dict(x16=x16, w_ih16=w_gl[d * H4:(d + 1) * H4], bias=bias_gl[d * H4:(d + 1) * H4]) for d in range(nd)]
            stored_h = None
            if B * T <= 4096:
                ops.rnn_fwd(cell, prec, B, T, H, fdirs, workspace=wsp)       # from the stored pre-activations, for comparison
                stored_h = out.clone()
                out.zero_(); gates.zero_(); cst.zero_()
            fdirs = [dict(d, xproj=None) for d in fdirs]
        ops.rnn_fwd(cell, prec, B, T, H, fdirs, workspace=wsp, xin=xi)
        fwd_kernel = ops._lib.load().tad_last_kernel().decode()
        ops.rnn_ws_status(wsp)
        if xin and stored_h is not None:
            assert fu.rel_err(out.cpu().numpy(), stored_h.cpu().numpy()) < BF16_EMU_TOL
        # and once more without the stash (inference): the same h
        out_inf = torch.zeros_like(out)
        idirs = [dict(d, gates=None, cstate=None, out=out_inf[:, k * H:(k + 1) * H]) for k, d in enumerate(fdirs)]
        ops.rnn_fwd(cell, prec, B, T, H, idirs, workspace=wsp, xin=xi)
        ops.rnn_ws_status(wsp)
        assert torch.equal(out_inf, out)
    else:
        ops.rnn_fwd(cell, prec, B, T, H, fdirs)
    ops.rnn_bwd(cell, prec, B, T, H, bdirs)
    dWih = torch.zeros(nd * GH, IN, device=dev)
    dbih, dbhh = torch.zeros(nd * GH, device=dev), torch.zeros(nd * GH, device=dev)
    dWhh = torch.zeros(nd, GH, H, device=dev)
    ops.linear_bwd_weight(prec, dg, xd, dWih, db=dbih, db2=dbhh if cell == 0 else None, gate_map=(H, G, IN_SLOTS[cell]))
    for d, rev in enumerate(rev_flags):
        ops.linear_bwd_weight(prec, dg[:, d * H4:(d + 1) * H4], out[:, d * H:(d + 1) * H], dWhh[d],
                              db=None if cell == 0 else dbhh[d * GH:(d + 1) * GH], T=T, tshift=1 if rev else -1,
                              gate_map=(H, G, HID_SLOTS[cell]))
    dx = torch.zeros(BT, IN, device=dev)
    if s16:
        ops.linear_bwd_data_b16(dg, w_gl, dx)
    else:
        ops.linear_bwd_data(prec, dg, w_gl, dx)
    torch.cuda.synchronize()
    unglue = lambda t: t.float().view(BT, nd, H, 4).permute(0, 1, 3, 2).cpu().numpy()      # -> [BT, nd, slot, H]
    return dict(out=out.cpu().numpy(), gates=unglue(gates), cst=cst.cpu().numpy(), xproj=unglue(xproj), dg=unglue(dg),
                dWih=dWih.view(nd, GH, IN).cpu().numpy(), dWhh=dWhh.cpu().numpy(), dbih=dbih.view(nd, GH).cpu().numpy(),
                dbhh=dbhh.view(nd, GH).cpu().numpy(), dx=dx.cpu().numpy(), oracle=oracle, x=x, dout=dout, G=G, s16=s16,
                fwd_kernel=fwd_kernel)


def _check_rnn(r, cell, B, T, H, tol):
    BT = B * T
    dx_ref = 0
    for d, o in enumerate(r["oracle"]):
        assert fu.rel_err(r["out"][:, d * H:(d + 1) * H], o["hs"].reshape(BT, H)) < tol, ("h", d)
        cache = o["cache"]
        if cell == 0:
            _, _, cs, gates, _ = cache
            assert fu.rel_err(r["cst"][:, d * H:(d + 1) * H], cs.reshape(BT, H)) < tol, ("c", d)
        else:
            gates = cache[2]
        # a bf16-stored stash may differ from the oracle's by one unit in the last place where a value sits on a
        # rounding boundary (2^-7 relative at most)
        gtol = 2.0 ** -7 if r["s16"] else tol
        assert fu.rel_err(r["gates"][:, d].reshape(BT, 4 * H), gates.reshape(BT, 4 * H)) < gtol, ("gates", d)
        assert fu.rel_err(r["dWih"][d], o["dWih"]) < 4 * tol, ("dWih", d)
        assert fu.rel_err(r["dWhh"][d], o["dWhh"]) < 4 * tol, ("dWhh", d)
        assert fu.rel_err(r["dbih"][d], o["dbih"]) < 4 * tol, ("dbih", d)
        assert fu.rel_err(r["dbhh"][d], o["dbhh"]) < 4 * tol, ("dbhh", d)
        dx_ref = dx_ref + o["dx"].reshape(BT, -1)
    assert fu.rel_err(r["dx"], dx_ref) < 4 * tol, "dx"


RNN_CASES = [(5, 6, 24, 11), (37, 5, 40, 16), (70, 4, 200, 24), (33, 3, 272, 11), (9, 4, 30, 8), (6, 3, 25, 8), (6, 3, 512, 16)]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("cell", [0, 1])
@pytest.mark.parametrize("B,T,H,IN", RNN_CASES)
def test_rnn_fwd_bwd(ops, dev, cell, prec, B, T, H, IN):
    """Both directions of one layer: outputs, stash, and every gradient the layer produces, against the oracle
    (bf16: the oracle rounds the matrix operands and, where the kernels store bf16, the stored tensors)."""
    rng = np.random.default_rng(100 * cell + 10 * prec + H)
    r = _run_rnn(ops, dev, cell, prec, B, T, H, [False, True], rng, IN=IN)
    assert r["s16"] == (bool(prec) and H % 2 == 0 and IN % 8 == 0)
    _check_rnn(r, cell, B, T, H, F32_TOL if prec == 0 else BF16_EMU_TOL)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("H", [200, 512])
@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_full_batch_forms(ops, dev, cell, H, prec):
    """At 4096+ windows per launch the sequence kernels take other forms than at the small batches above (bf16, H = 200:
    the LSTM forward as two rounds of 16-row workgroups with resident W_hh, the GRU forward at 32 rows with resident W_hh,
    the backward at 32 rows with most of the transposed W_hh in LDS and registers; H = 512, the hidden size of every
    reference VAE and of the Metro / AirQuality denoisers: 32 rows per workgroup with two chunks of hidden units per wave,
    the GRU's backward gate by gate): the same outputs, stash and gradients against the oracle, on a batch that leaves
    the last workgroup ragged.  fp32 at the same sizes: 1e-4."""
    B, T, IN = 4100, 2, 24
    rng = np.random.default_rng(4100 + cell + H + prec)
    r = _run_rnn(ops, dev, cell, prec, B, T, H, [False, True], rng, IN=IN)
    assert r["s16"] == bool(prec)
    _check_rnn(r, cell, B, T, H, BF16_EMU_TOL if prec else F32_TOL)


@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_heads_layout_at_hidden_512(ops, dev, cell):
    """The VAE encoders' launch shape (reference VAE.py:119-120, 154-155): two heads that both run forward in time over
    one shared input, hidden 512, full 32-row workgroups, three steps (so that the recurrent gradient crosses two
    step boundaries), against the oracle."""
    B, T, H, IN = 4100, 3, 512, 32
    rng = np.random.default_rng(512 + cell)
    r = _run_rnn(ops, dev, cell, 1, B, T, H, [False, False], rng, IN=IN)
    assert r["s16"]
    _check_rnn(r, cell, B, T, H, BF16_EMU_TOL)


@pytest.mark.parametrize("B,T,revs", [(4100, 3, (False, True)), (4100, 2, (False, False)), (100, 6, (False, True)), (60, 4, (False,)),
                                      (70, 5, (True,))])
@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_cluster_form_at_hidden_512(ops, dev, cell, B, T, revs):
    """tad_rnn_fwd_ws at H = 512, bf16: eight workgroups per 128 batch rows, each with 64 hidden units' rows of W_hh in
    registers, exchanging h_t through the workspace every step.  Outputs and stash (and the gradients the backward makes of
    them) against the oracle as for the other forms; 4100 rows = 33 tiles x 2 directions is more work items than groups
    fit the chip at once (a group walks several items and re-uses its exchange buffers), with a ragged last tile; 100, 70 and
    60 rows fill the second half of a tile partly / not at all.  The launch without a stash returns the same h bit for bit."""
    H, IN = 512, 16
    rng = np.random.default_rng(5120 + cell + B + T)
    r = _run_rnn(ops, dev, cell, 1, B, T, H, list(revs), rng, IN=IN, cluster=True)
    assert r["s16"] and r["fwd_kernel"].startswith("rnn_cluster_fwd_kernel"), r["fwd_kernel"]
    _check_rnn(r, cell, B, T, H, BF16_EMU_TOL)


@pytest.mark.parametrize("B,T,IN,revs", [(4100, 3, 8, (False, False)), (300, 5, 32, (False, False)), (100, 6, 24, (False, True)),
                                         (70, 4, 16, (True,)), (1, 2, 8, (False,)), (129, 2, 32, (False, False))])
@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_cluster_form_multiplies_a_narrow_input_itself(ops, dev, cell, B, T, IN, revs):
    """tad_rnn_fwd_ws_x: a layer input of 8 to 32 columns (the first layer of the VAE's encoder GRUs at hidden size 512,
    reference VAE.py:119-120 -> 154-155, reads the table's features) is multiplied by W_ih inside the cluster form, from
    the bf16 image of W_ih and the bias that tad_rnn_pack_input writes; no pre-activation tensor exists.  Outputs, stash
    and the gradients that the backward makes of them against the oracle whose pre-activations skip the bf16 storage
    rounding; h against the launch that reads stored pre-activations to the bf16 tolerance; the launch without a stash
    gives the same h bit for bit."""
    H = 512
    rng = np.random.default_rng(777 + cell + B + IN)
    r = _run_rnn(ops, dev, cell, 1, B, T, H, list(revs), rng, IN=IN, cluster=True, xin=True)
    assert r["s16"] and r["fwd_kernel"].startswith("rnn_cluster_fwd_kernel") and r["fwd_kernel"].endswith("true>"), r["fwd_kernel"]
    _check_rnn(r, cell, B, T, H, BF16_EMU_TOL)


def test_rnn_fwd_ws_x_rejects_what_the_cluster_form_cannot_take(ops, dev):
    """No silent other path: an input wider than 32 columns, a hidden size without the form, or no usable workspace is
    an error that names the entry point."""
    H, B, T = 512, 40, 2
    ws = torch.zeros(ops.rnn_workspace_bytes(1, 1, 1, B, T, H, 1), dtype=torch.uint8, device=dev)
    whh = torch.zeros(3 * H, H, device=dev)
    pf = torch.empty(ops.rnn_packed_bytes(1, H, 3), dtype=torch.uint8, device=dev)
    ops.rnn_pack_weights(1, H, 3, whh, pf, None)
    out = torch.zeros(B * T, H, device=dev)
    dirs = [dict(xproj=None, w_packed=pf, b_hh=torch.zeros(3 * H, device=dev), out=out, reverse=False)]

    def xin(k):
        return [dict(x16=torch.zeros(B * T, k, dtype=torch.bfloat16, device=dev), w_ih16=torch.zeros(4 * H, 128, dtype=torch.bfloat16, device=dev)[:, :k],
                     bias=torch.zeros(4 * H, device=dev))]
    ops.rnn_fwd(1, 1, B, T, H, dirs, workspace=ws, xin=xin(32))
    with pytest.raises(ops._lib.TadError, match="tad_rnn_fwd_ws_x"):
        ops.rnn_fwd(1, 1, B, T, H, dirs, workspace=ws, xin=xin(40))
    with pytest.raises(ops._lib.TadError, match="tad_rnn_fwd_ws_x"):
        ops.rnn_fwd(1, 1, B, T, H, dirs, workspace=torch.zeros(4096, dtype=torch.uint8, device=dev), xin=xin(32))
    torch.cuda.synchronize()


def _cluster_problem(ops, dev, cell, B, T, nd, seed, xin=False):
    """Random operands of one forward launch at H = 512 in bf16 storage, and a function that runs it into fresh outputs.
    xin: the launch multiplies a 16-column input by W_ih itself instead of reading pre-activations."""
    H, G = 512, (4 if cell == 0 else 3)
    g = torch.Generator(device="cpu").manual_seed(seed)
    BT, H4 = B * T, 4 * H
    xproj = (torch.randn(BT, nd * H4, generator=g) * 0.7).to(dev).to(torch.bfloat16)
    xi = None
    if xin:
        x16 = torch.randn(BT, 16, generator=g).to(dev).to(torch.bfloat16)
        w16 = torch.zeros(nd * H4, 128, dtype=torch.bfloat16, device=dev)
        w16[:, :16] = (torch.randn(nd * H4, 16, generator=g) * 0.2).to(dev).to(torch.bfloat16)
        bias = (torch.randn(nd * H4, generator=g) * 0.1).to(dev)
        xi = [dict(x16=x16, w_ih16=w16[d * H4:(d + 1) * H4, :16], bias=bias[d * H4:(d + 1) * H4]) for d in range(nd)]
        # the same pre-activations as a stored tensor (rounded to bf16), for run(..., use_xin=False)
        xproj = (x16.float() @ w16[:, :16].float().T + bias).to(torch.bfloat16)
    whh = ((torch.rand(nd, G * H, H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    bhh = ((torch.rand(nd, G * H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    nbytes = ops.rnn_packed_bytes(1, H, G)
    pf = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
    for d in range(nd):
        ops.rnn_pack_weights(1, H, G, whh[d], pf[d], None)

    def run(workspace, stash=True, use_xin=xin):
        out = torch.zeros(BT, nd * H, device=dev)
        out16 = torch.zeros(BT, nd * H, dtype=torch.bfloat16, device=dev)
        gates = torch.zeros(BT, nd * H4, dtype=torch.bfloat16, device=dev)
        cst = torch.zeros(BT, nd * H, device=dev)
        dirs = [dict(xproj=None if use_xin else xproj[:, d * H4:(d + 1) * H4], w_packed=pf[d], b_hh=bhh[d] if cell == 1 else None,
                     out=out[:, d * H:(d + 1) * H], out16=out16[:, d * H:(d + 1) * H],
                     gates=gates[:, d * H4:(d + 1) * H4] if stash else None,
                     cstate=cst[:, d * H:(d + 1) * H] if (stash and cell == 0) else None, reverse=bool(d & 1)) for d in range(nd)]
        ops.rnn_fwd(cell, 1, B, T, H, dirs, workspace=workspace, xin=xi if use_xin else None)
        return out, out16, gates, cst
    return run


@pytest.mark.parametrize("xin", [False, True])
@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_cluster_form_beside_other_work_and_replayed(ops, dev, cell, xin):
    """The eight workgroups of a group hand h_t to each other through device memory every step.  The launch must give the
    same bits (a) alone on the chip, (b) while another stream keeps the chip busy with memory-bound kernels, so that the
    workgroups of a group start at different times and meet loaded memory paths, (c) many times in a row on one workspace
    (arrival counters are re-zeroed by every launch; the consumers' caches have seen the exchange buffers before), and
    (d) replayed from a captured HIP graph (the memset of the counters is a node of the graph).  Against the forms that
    need no workspace: to the tolerance of the bf16 mode (another summation order).  xin: the same for the launch that
    multiplies a narrow layer input itself (tad_rnn_fwd_ws_x; it has no form without a workspace to compare with)."""
    B, T, nd = 700, 7, 2
    run = _cluster_problem(ops, dev, cell, B, T, nd, 77 + cell, xin=xin)
    nws = ops.rnn_workspace_bytes(cell, 1, 1, B, T, 512, nd)
    assert nws > 0
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    ref = run(ws)
    assert ops._lib.load().tad_last_kernel().decode().startswith("rnn_cluster_fwd_kernel")
    ops.rnn_ws_status(ws)
    if not xin:
        plain = run(None)
        assert not ops._lib.load().tad_last_kernel().decode().startswith("rnn_cluster_fwd_kernel")
        for a, b in zip(ref, plain):
            assert fu.rel_err(a.float().cpu().numpy(), b.float().cpu().numpy()) < BF16_EMU_TOL
    # (b) + (c): a side stream streams 256 MB copies all the while
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device=dev)
    big2 = torch.empty_like(big)
    main = torch.cuda.current_stream()
    for rep in range(6):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.copy_2d(big.view(-1, 4096), big2.view(-1, 4096))
        got = run(ws)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), rep
    main.wait_stream(side)
    ops.rnn_ws_status(ws)
    # (d) captured and replayed
    g = torch.cuda.HIPGraph() if hasattr(torch.cuda, "HIPGraph") else torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(main)
    with torch.cuda.stream(cap):
        warm = run(ws)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            held = run(ws)
    for rep in range(12):
        # (other fills and small copies of the process between the replays: a memset node replayed after them was seen to
        # write another value than the captured zero, which is why the launch zeroes its counters by a kernel of its own)
        assert int((ref[0] > 0.05 * rep).nonzero().shape[0]) > 0
        for t in held:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(held, ref):
            assert torch.equal(a, b), rep
        # every arrival counter of the first group: 8 workgroups x T steps, counted from zero
        assert ws[:512].view(torch.int32)[::32].tolist() == [8 * T] * 4, rep
    ops.rnn_ws_status(ws)


@pytest.mark.parametrize("xin", [False, True])
@pytest.mark.parametrize("B,T,nd", [(1, 1, 1), (3, 1, 2), (33, 2, 1), (129, 3, 2)])
def test_rnn_cluster_form_small_shapes(ops, dev, B, T, nd, xin):
    """One row, one step (nothing is ever handed over), a tile whose last three quarters are empty, a second tile of one row:
    the form with the workspace against the forms without, with and without the stash.  xin: the launch that multiplies
    a 16-column input itself against the forms without a workspace fed the same pre-activations as a stored tensor."""
    for cell in (0, 1):
        run = _cluster_problem(ops, dev, cell, B, T, nd, 5 + B, xin=xin)
        ws = torch.zeros(ops.rnn_workspace_bytes(cell, 1, 1, B, T, 512, nd), dtype=torch.uint8, device=dev)
        got = run(ws)
        assert ops._lib.load().tad_last_kernel().decode().startswith("rnn_cluster_fwd_kernel")
        ops.rnn_ws_status(ws)
        ref = run(None, use_xin=False)
        # (xin: the reference's pre-activations went through bf16 storage, the launch's did not: results stored as bf16
        # may then sit one bf16 step apart, 2^-7 relative at most)
        tol = 2.0 ** -7 if xin else BF16_EMU_TOL
        for a, b in zip(got, ref):
            assert fu.rel_err(a.float().cpu().numpy(), b.float().cpu().numpy()) < tol
        inf = run(ws, stash=False)
        assert torch.equal(inf[0], got[0]) and torch.equal(inf[1], got[1])


def test_rnn_workspace_status_is_reported(ops, dev):
    """The word a launch sets when its workgroups give up waiting for each other: ops.check_rnn_workspaces (called by the
    trainers and the sampler where they synchronise) raises on it, and keeps quiet on workspaces of clean launches."""
    run = _cluster_problem(ops, dev, 0, 64, 2, 1, 21)
    ws = torch.zeros(ops.rnn_workspace_bytes(0, 1, 1, 64, 2, 512, 1), dtype=torch.uint8, device=dev)
    run(ws)
    ops.check_rnn_workspaces()
    ws[16384:16388] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8, device=dev)       # what a launch that timed out leaves
    with pytest.raises(ops._lib.TadError, match="ran out of time"):
        ops.check_rnn_workspaces()
    ws.zero_()
    ops.check_rnn_workspaces()


def test_rnn_fwd_ws_falls_back_where_the_cluster_form_does_not_apply(ops, dev):
    """Other hidden sizes, fp32, a workspace that is too small or views the form cannot address: tad_rnn_fwd_ws is
    tad_rnn_fwd, bit for bit."""
    assert ops.rnn_workspace_bytes(0, 1, 1, 64, 4, 200, 2) == 0 and ops.rnn_workspace_bytes(0, 0, 0, 64, 4, 512, 2) == 0
    run = _cluster_problem(ops, dev, 0, 40, 3, 2, 9)
    small = torch.zeros(4096, dtype=torch.uint8, device=dev)
    a = run(small)
    assert not ops._lib.load().tad_last_kernel().decode().startswith("rnn_cluster_fwd_kernel")
    b = run(None)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_rnn_bwd_phased_path(ops, dev):
    """fp32 LSTM with H = 320: the gate-gradient tile exceeds LDS and the backward kernel
    stages it gate by gate."""
    rng = np.random.default_rng(77)
    B, T, H = 9, 3, 320
    r = _run_rnn(ops, dev, 0, 0, B, T, H, [False], rng)
    _check_rnn(r, 0, B, T, H, F32_TOL)


def test_rnn_inference_matches_training(ops, dev):
    """gates = NULL (inference) produces the same h as the training launch."""
    rng = np.random.default_rng(3)
    B, T, H, G = 17, 5, 40, 4
    xproj = D(rnd(rng, B * T, 4 * H), dev)
    Whh = D(rnd(rng, G * H, H, scale=0.2), dev)
    nbytes = ops.rnn_packed_bytes(0, H, G)
    pf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ops.rnn_pack_weights(0, H, G, Whh, pf, None)
    o1 = torch.zeros(B * T, H, device=dev)
    o2 = torch.zeros(B * T, H, device=dev)
    st = torch.zeros(B * T, 4 * H, device=dev)
    cs = torch.zeros(B * T, H, device=dev)
    ops.rnn_fwd(0, 0, B, T, H, [dict(xproj=xproj, w_packed=pf, out=o1, gates=st, cstate=cs, reverse=False)])
    ops.rnn_fwd(0, 0, B, T, H, [dict(xproj=xproj, w_packed=pf, out=o2, gates=None, reverse=False)])
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("cell", [0, 1])
@pytest.mark.parametrize("B,T,H", [(19, 5, 200), (8, 3, 168), (33, 4, 256), (5, 6, 160), (12, 3, 196), (9, 4, 40), (4100, 3, 200)])
def test_rnn_inference_16_row_form(ops, dev, cell, B, T, H):
    """bf16 launches at 8 rows per workgroup take the 16-row MFMA form of the forward kernel (W_hh in registers + LDS
    where it fits; H = 40 is below its register-resident k-steps and falls back), with or without a gate stash; full
    batches (B = 4100) keep the 32-row forms: h against the oracle (which rounds the operands and stored tensors as the kernels do), the bf16
    copy of h against h, and the launches with and without a stash against each other on the same inputs."""
    rng = np.random.default_rng(1000 * cell + H)
    G, nd, IN = (4 if cell == 0 else 3), 2, 8
    BT, H4, GH = B * T, 4 * H, G * H
    x = rnd(rng, B, T, IN)
    k = 1.0 / np.sqrt(H)
    Wih = rng.uniform(-k, k, (nd, GH, IN)).astype(np.float32)
    Whh = rng.uniform(-k, k, (nd, GH, H)).astype(np.float32)
    bih = rng.uniform(-k, k, (nd, GH)).astype(np.float32)
    bhh = rng.uniform(-k, k, (nd, GH)).astype(np.float32)
    fwd = ornn.lstm_dir_fwd if cell == 0 else ornn.gru_dir_fwd
    f64 = lambda a: a.astype(np.float64)
    with oops.bf16_operands(storage=True):
        ref = [fwd(f64(x), f64(Wih[d]), f64(Whh[d]), f64(bih[d]), f64(bhh[d]), bool(d))[0].reshape(BT, H) for d in range(nd)]
    xd = D(x, dev).view(BT, IN)
    Wihd, Whhd, bihd, bhhd = D(Wih, dev), D(Whh, dev), D(bih, dev), D(bhh, dev)
    w_gl = torch.empty(nd * H4, IN, dtype=torch.bfloat16, device=dev)
    bias_gl = torch.empty(nd * H4, device=dev)
    ops.rnn_pack_input(cell, H, Wihd.view(nd * GH, IN), bihd.view(-1), bhhd.view(-1), w_gl, bias_gl)
    xproj = torch.zeros(BT, nd * H4, dtype=torch.bfloat16, device=dev)
    ops.linear_fwd_b16(xd, w_gl, bias_gl, xproj)
    nbytes = ops.rnn_packed_bytes(1, H, G)
    pf = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
    for d in range(nd):
        ops.rnn_pack_weights(1, H, G, Whhd[d], pf[d], None)
    res = {}
    for form in ("plain", "stash"):
        out = torch.full((BT, nd * H), float("nan"), device=dev)
        out16 = torch.full((BT, nd * H), float("nan"), device=dev, dtype=torch.bfloat16)
        gates = torch.zeros(BT, nd * H4, dtype=torch.bfloat16, device=dev) if form == "stash" else None
        cst = torch.zeros(BT, nd * H, device=dev) if (form == "stash" and cell == 0) else None
        dirs = []
        for d in range(nd):
            sl, hl = slice(d * H4, (d + 1) * H4), slice(d * H, (d + 1) * H)
            dirs.append(dict(xproj=xproj[:, sl], w_packed=pf[d], b_hh=bhhd[d] if cell == 1 else None, out=out[:, hl], out16=out16[:, hl],
                             gates=gates[:, sl] if gates is not None else None, cstate=cst[:, hl] if cst is not None else None,
                             reverse=bool(d)))
        ops.rnn_fwd(cell, 1, B, T, H, dirs)
        torch.cuda.synchronize()
        res[form] = (out.cpu().numpy(), out16.float().cpu().numpy())
    o, o16 = res["plain"]
    assert not np.isnan(o).any() and not np.isnan(o16).any()
    for d in range(nd):
        assert fu.rel_err(o[:, d * H:(d + 1) * H], ref[d]) < BF16_EMU_TOL, ("h", d)
    assert np.array_equal(o16, torch.from_numpy(o).bfloat16().float().numpy())          # the bf16 copy is h rounded
    assert fu.rel_err(o, res["stash"][0]) < BF16_EMU_TOL


# ------------------------------------------------------------------ elementwise family
def test_add_noise_and_loss(ops, dev):
    rng = np.random.default_rng(1)
    B, T, K = 9, 7, 6
    x0, noise = rnd(rng, B, T, K), rnd(rng, B, T, K)
    i_idx = rng.integers(0, 100, B).astype(np.int32)
    _, alphas = oops.schedule(100)
    xn = torch.zeros(B, T, K, device=dev)
    i_f = torch.zeros(B, T, device=dev)
    ops.add_noise(D(x0, dev), D(noise, dev), D(i_idx, dev), D(alphas, dev), xn, i_f)
    assert fu.rel_err(xn.cpu().numpy(), oops.add_noise(x0, i_idx, noise, alphas)) < 1e-6
    assert np.array_equal(i_f.cpu().numpy(), np.repeat(i_idx[:, None], T, 1).astype(np.float32))
    pred = rnd(rng, B, T, K)
    dp = torch.zeros(B, T, K, device=dev)
    ls = torch.zeros(1, device=dev)
    ops.mse_loss(D(pred, dev), D(noise, dev), 2.0 / pred.size, dp, ls)
    assert abs(ls.item() - float(((pred - noise).astype(np.float64) ** 2).sum())) < 1e-5 * ls.item()
    assert fu.rel_err(dp.cpu().numpy(), 2.0 / pred.size * (pred - noise)) < 1e-6


def test_sampler_update(ops, dev):
    rng = np.random.default_rng(2)
    x, pred, z = rnd(rng, 4, 5, 6), rnd(rng, 4, 5, 6), rnd(rng, 4, 5, 6)
    betas, alphas = oops.schedule(100)
    for s, last in ((57, False), (0, True)):
        b, a = np.float32(betas[s]), np.float32(alphas[s])
        xd = D(x, dev)
        ops.sampler_update(xd, D(pred, dev), D(z, dev), 1 / np.sqrt(1 - b), b / np.sqrt(1 - a), np.sqrt(b), last)
        assert fu.rel_err(xd.cpu().numpy(), oops.sampler_update(x, pred, z, b, a, last)) < 1e-6


def test_sincos_embed_posenc_silu(ops, dev):
    rng = np.random.default_rng(4)
    rows, nn_, nd, E = 50, 5, 3, 16
    cards = [2, 7, 1]
    x = np.concatenate([np.stack([rng.integers(0, c, rows) for c in cards], 1).astype(np.float32),
                        rng.random((rows, nn_), dtype=np.float32)], 1)
    xd = D(x, dev)
    out = torch.zeros(rows, 16 * nn_, device=dev)
    ops.sincos_features(xd[:, nd:], out)
    assert np.max(np.abs(out.cpu().numpy() - oops.sincos_features(x[:, nd:]))) < 2e-6
    tables = rnd(rng, sum(cards), E)
    off = [0, 2, 9]
    eo = torch.zeros(rows, nd * E + 3, device=dev)
    ops.embed_fwd(xd[:, :nd], D(tables, dev), off, cards, E, eo[:, :nd * E])
    ref = np.concatenate([tables[off[d] + x[:, d].astype(int)] for d in range(nd)], 1)
    assert np.array_equal(eo[:, :nd * E].cpu().numpy(), ref)
    dout = rnd(rng, rows, nd * E)
    dt = torch.zeros(sum(cards), E, device=dev)
    ops.embed_bwd(xd[:, :nd], D(dout, dev), dt, off, cards, E)
    gref = np.zeros_like(tables, dtype=np.float64)
    for d in range(nd):
        np.add.at(gref, off[d] + x[:, d].astype(int), dout[:, d * E:(d + 1) * E])
    assert fu.rel_err(dt.cpu().numpy(), gref) < 1e-5
    # positional encodings (t_enc: max 1; i_enc: max 100)
    H = 24
    for mx, tv in ((1.0, rng.random(rows, dtype=np.float32)), (100.0, rng.integers(0, 100, rows).astype(np.float32))):
        sc, sh = oops.posenc_tables(H, mx)
        w, b = rnd(rng, H // 2, 1), rnd(rng, H // 2)
        po = torch.zeros(rows, H, device=dev)
        ops.posenc_fwd(D(tv, dev), 1, rows, H, 1.0 / mx, D(w, dev), D(b, dev), D(sc, dev), D(sh, dev), po)
        assert np.max(np.abs(po.cpu().numpy() - oops.posenc_fwd(tv[:, None], w, b, mx, sc, sh))) < 1e-5
    zt = rnd(rng, 40, 33)
    a = torch.zeros(40, 33, device=dev)
    ops.silu_fwd(D(zt, dev), a)
    assert fu.rel_err(a.cpu().numpy(), oops.silu(zt.astype(np.float64))) < 1e-6
    da = rnd(rng, 40, 33)
    dz = torch.zeros(40, 33, device=dev)
    ops.silu_bwd(D(zt, dev), D(da, dev), dz)
    assert fu.rel_err(dz.cpu().numpy(), da * oops.silu_grad(zt.astype(np.float64))) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_feature_kernels_wide_and_narrow_forms(ops, dev, dtype):
    """sincos / embedding lookup / positional encoding pick a form with 16-byte stores when the rows of the result are
    aligned for it and an element-per-thread form otherwise: the two must agree bit for bit (same arithmetic)."""
    rng = np.random.default_rng(11)
    rows, nn_, nd, E, H = 1000, 27, 2, 64, 200
    cards = [8, 3]
    off = [0, 8]
    x = np.concatenate([np.stack([rng.integers(0, c, rows) for c in cards], 1).astype(np.float32),
                        rng.random((rows, nn_), dtype=np.float32)], 1)
    xd = D(x, dev)
    tables = D(rnd(rng, sum(cards), E), dev)

    def both(width, call):
        wide = torch.zeros(rows, width + 8, device=dev, dtype=dtype)         # row pitch a multiple of 16 bytes
        narrow = torch.zeros(rows, width + 9, device=dev, dtype=dtype)       # odd pitch: rows are not 16-byte aligned
        call(wide[:, :width]); call(narrow[:, :width])
        assert torch.equal(wide[:, :width], narrow[:, :width])
        assert not wide[:, width:].any() and not narrow[:, width:].any()
        return wide[:, :width].float().cpu().numpy()

    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
    sc = both(16 * nn_, lambda o: ops.sincos_features(xd[:, nd:], o))
    assert np.max(np.abs(sc - oops.sincos_features(x[:, nd:]))) < tol
    em = both(nd * E, lambda o: ops.embed_fwd(xd[:, :nd], tables, off, cards, E, o))
    ref = np.concatenate([tables.cpu().numpy()[off[d] + x[:, d].astype(int)] for d in range(nd)], 1)
    assert np.max(np.abs(em - ref)) <= (0 if dtype == torch.float32 else 2.0 ** -8 * np.abs(ref).max())
    for mx, tv in ((1.0, rng.random(rows, dtype=np.float32)), (100.0, rng.integers(0, 100, rows).astype(np.float32))):
        s_, sh = oops.posenc_tables(H, mx)
        w, b = rnd(rng, H // 2, 1), rnd(rng, H // 2)
        po = both(H, lambda o: ops.posenc_fwd(D(tv, dev), 1, rows, H, 1.0 / mx, D(w, dev), D(b, dev), D(s_, dev), D(sh, dev), o))
        pref = oops.posenc_fwd(tv[:, None], w, b, mx, s_, sh)
        assert np.max(np.abs(po - pref)) < (1e-5 if dtype == torch.float32 else 2.0 ** -8 * max(1.0, np.abs(pref).max()))


@pytest.mark.parametrize("M,N,K", [(24 * 41, 1600, 200), (700, 800, 128), (5, 64, 72), (4096, 600, 200)])
def test_linear_bwd_data_bf16_result(ops, dev, M, N, K):
    """tad_linear_bwd_data_b16o (bf16 result, bf16 mask operand, written by the product's epilogue) against the float32
    result of tad_linear_bwd_data_b16 rounded and masked by tad_cvt_bf16: the same accumulators, bit for bit."""
    rng = np.random.default_rng(M + K)
    dY16 = torch.from_numpy(rnd(rng, M, N)).bfloat16().to(dev)
    W16 = torch.from_numpy(rnd(rng, N, K)).bfloat16().to(dev)
    mask = torch.from_numpy(rnd(rng, M, K)).bfloat16().to(dev)
    mask[::3, ::5] = 0                                                  # zeros and negative zeros mask like negatives
    mask[1::3, 1::5] = -0.0
    for m in (None, mask):
        dX = torch.zeros(M, K, device=dev)
        ops.linear_bwd_data_b16(dY16, W16, dX)
        want = torch.full((M, K + 8), 3.0, dtype=torch.bfloat16, device=dev)
        ops.cvt_bf16(dX, want[:, :K], relu_of=m)
        got = torch.full((M, K + 8), 3.0, dtype=torch.bfloat16, device=dev)
        ops.linear_bwd_data_b16o(dY16, W16, got[:, :K], relu_of16=m)
        assert torch.equal(got, want)
        ref = dY16.float().cpu().numpy().astype(np.float64) @ W16.float().cpu().numpy().astype(np.float64)
        if m is not None:
            ref = ref * (m.float().cpu().numpy() > 0)
        g = got[:, :K].float().cpu().numpy()
        assert np.all(np.abs(g - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("M,N,L", [(24 * 37, 200, 100), (5000, 64, 32), (333, 512, 256), (3, 24, 12)])
def test_enc_grad_equals_column_sums_of_the_product(ops, dev, M, N, L):
    """tad_enc_grad_b16 against the form it replaces, in float64 numpy: the product g W[:, cols] of the bf16-rounded
    operands followed by its plain and row-weighted column sums (the model tests check the same gradients against the
    oracle, which forms that product)."""
    rng = np.random.default_rng(M + N)
    g = torch.from_numpy(rnd(rng, M, N)).to(dev).to(torch.bfloat16)
    W = rnd(rng, N, 4 * N)
    s1 = rng.random(M, dtype=np.float32)
    s2 = rng.integers(0, 100, M).astype(np.float32)
    col1, col2, mul2 = N, 2 * N, 1.0 / 100.0
    out = [torch.full((L,), 0.5, device=dev) for _ in range(4)]                 # gradients accumulate
    sums = torch.full((3 * N,), 7.0, device=dev)                                 # workspace: cleared by the call
    ops.enc_grad_b16(g, D(s1, dev), D(s2, dev), D(W, dev), col1, col2, L, 1.0, mul2, sums, *out)
    g64 = g.float().cpu().numpy().astype(np.float64)
    W16 = torch.from_numpy(W).to(torch.bfloat16).float().numpy().astype(np.float64)
    ref = []
    for col, s_, mul in ((col1, s1, 1.0), (col2, s2, mul2)):
        d = g64 @ W16[:, col:col + L]
        ref += [0.5 + d.sum(0), 0.5 + mul * (d * s_[:, None].astype(np.float64)).sum(0)]
    for o, r in zip(out, ref):
        assert fu.rel_err(o.cpu().numpy(), r) < 2e-5
    sums_ref = np.concatenate([(g64 * s1[:, None]).sum(0), (g64 * s2[:, None]).sum(0), g64.sum(0)])
    assert fu.rel_err(sums.cpu().numpy(), sums_ref) < 2e-5


def test_enc_grad_replayed_from_a_graph_starts_from_zeroed_sums(ops, dev):
    """The call clears its workspace of partial sums itself; captured in a graph and replayed between other work of the
    process (fills, small copies), every replay must start from zero again."""
    M, N, L = 24 * 37, 200, 100
    rng = np.random.default_rng(5)
    g = torch.from_numpy(rnd(rng, M, N)).to(dev).to(torch.bfloat16)
    W, s1, s2 = D(rnd(rng, N, 4 * N), dev), D(rng.random(M, dtype=np.float32), dev), D(rng.integers(0, 100, M).astype(np.float32), dev)
    sums = torch.full((3 * N,), 7.0, device=dev)

    def run(out):
        ops.enc_grad_b16(g, s1, s2, W, N, 2 * N, L, 1.0, 0.01, sums, *out)
    want = [torch.zeros(L, device=dev) for _ in range(4)]
    run(want)
    sums_want = sums.clone()
    held = [torch.zeros(L, device=dev) for _ in range(4)]
    gr = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        run([torch.zeros(L, device=dev) for _ in range(4)])
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=cap):
            run(held)
    for rep in range(10):
        assert int((g.float() > 0.1 * rep).nonzero().shape[0]) > 0
        for t in held:
            t.zero_()
        sums.fill_(3.0 + rep)
        gr.replay()
        torch.cuda.synchronize()
        assert fu.rel_err(sums.cpu().numpy(), sums_want.cpu().numpy()) < 1e-5, rep
        for a, b in zip(held, want):
            assert fu.rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, rep


@pytest.mark.parametrize("Dn", [48, 400, 1024])
def test_layernorm(ops, dev, Dn):
    rng = np.random.default_rng(Dn)
    rows = 77
    x, dy = rnd(rng, rows, Dn), rnd(rng, rows, Dn)
    g, b = rnd(rng, Dn), rnd(rng, Dn)
    y = torch.zeros(rows, Dn, device=dev)
    mr = torch.zeros(rows, 2, device=dev)
    ops.layernorm_fwd(D(x, dev), D(g, dev), D(b, dev), y, mr)
    yr, cache = oops.layernorm_fwd(x.astype(np.float64), g, b)
    assert fu.rel_err(y.cpu().numpy(), yr) < 1e-5
    dx = torch.zeros(rows, Dn, device=dev)
    dg = torch.zeros(Dn, device=dev)
    db = torch.zeros(Dn, device=dev)
    ops.layernorm_bwd(D(dy, dev), D(x, dev), D(g, dev), mr, dx, dg, db)
    dxr, dgr, dbr = oops.layernorm_bwd(dy.astype(np.float64), g.astype(np.float64), cache)
    assert fu.rel_err(dx.cpu().numpy(), dxr) < 1e-5
    assert fu.rel_err(dg.cpu().numpy(), dgr) < 1e-5
    assert fu.rel_err(db.cpu().numpy(), dbr) < 1e-5


def test_gather_and_adam(ops, dev):
    from oracle.optim import Adam
    rng = np.random.default_rng(6)
    src = rnd(rng, 20, 4, 3)
    idx = rng.permutation(20)[:7].astype(np.int64)
    out = torch.zeros(7, 4, 3, device=dev)
    ops.gather_rows(D(src, dev), D(idx, dev), out)
    assert np.array_equal(out.cpu().numpy(), src[idx])
    n = 1000
    p = rnd(rng, n)
    pd = D(p.copy(), dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    sd = {"p": p.copy()}
    opt = Adam(sd, lr=2e-4, weight_decay=1e-6)
    for step in range(1, 4):
        g = rnd(rng, n)
        ops.adam_step(pd, D(g, dev), m, v, 2e-4, 0.9, 0.999, 1e-8, 1e-6, step)
        opt.step({"p": g})
    assert fu.rel_err(pd.cpu().numpy() - p, sd["p"] - p) < 1e-4      # the UPDATE agrees, not just the value


# ------------------------------------------------------------------ VAE family
def test_vae_reparam(ops, dev):
    rng = np.random.default_rng(8)
    n = 500
    mu, lv, eps, dz = rnd(rng, n), rnd(rng, n, scale=0.5), rnd(rng, n), rnd(rng, n)
    z = torch.zeros(n, device=dev)
    ks = torch.zeros(1, device=dev)
    ops.vae_reparam_fwd(D(mu, dev), D(lv, dev), D(eps, dev), z, ks)
    assert fu.rel_err(z.cpu().numpy(), mu + eps * np.exp(0.5 * lv)) < 1e-6
    ref = float((1 + lv.astype(np.float64) - mu.astype(np.float64) ** 2 - np.exp(lv.astype(np.float64))).sum())
    assert abs(ks.item() - ref) < 1e-4 * abs(ref)
    dmu = torch.zeros(n, device=dev)
    dlv = torch.zeros(n, device=dev)
    ops.vae_reparam_bwd(D(dz, dev), D(mu, dev), D(lv, dev), D(eps, dev), 0.3, dmu, dlv)
    assert fu.rel_err(dmu.cpu().numpy(), dz + 0.3 * mu) < 1e-6
    assert fu.rel_err(dlv.cpu().numpy(), dz * eps * 0.5 * np.exp(0.5 * lv) - 0.15 * (1 - np.exp(lv))) < 1e-5


@pytest.mark.parametrize("schema,B,T", [("numeric", 6, 5), ("mixed", 6, 5), ("metro", 6, 5), ("mixed", 301, 17), ("metro", 257, 24), ("mixed", 2100, 24),
                                        ("manycats", 40, 7)])
def test_vae_heads_loss(ops, dev, schema, B, T):
    """(The launch has two forms: one thread per head column for up to eight categorical features and 256 head columns --
    several row groups per workgroup from 1024 x 7 rows on -- and one thread per row beyond: `manycats` takes the second.)"""
    from oracle import vae as ovae
    from oracle.denoiser import Schema
    s = fu.SCHEMAS[schema] if schema != "manycats" else dict(n_bins=1, n_cats=9, n_nums=2, cards=[3, 2, 4, 5, 2, 3, 6, 2, 4])
    sch = Schema(s["n_bins"], s["n_cats"], s["n_nums"], list(s["cards"]))
    rng = np.random.default_rng(12)
    rows = B * T
    data = fu.make_data(s, N=B, T=T, K=4, seed=5, mask_kind="bern")
    F = data["data"].shape[-1]
    NH = sch.n_bins + sum(sch.cards) + sch.n_nums
    raw = rnd(rng, rows, NH)
    mask = data["mask"].reshape(rows, F)
    target = (data["data"] * data["mask"]).reshape(rows, F)
    # oracle: rebuild the head outputs from the raw pre-activations
    out = {}
    o = 0
    if sch.n_bins:
        out["bins"] = (raw[:, o:o + sch.n_bins] + np.where(mask[:, :sch.n_bins] == 0, -100.0, 0.0)).reshape(B, T, -1)
        o += sch.n_bins
    if sch.n_cats:
        out["cats"] = []
        for j, c in enumerate(sch.cards):
            lg = raw[:, o:o + c].copy()
            lg[:, 0] += np.where(mask[:, sch.n_bins + j] == 0, 100.0, 0.0)
            out["cats"].append(lg.reshape(B, T, c))
            o += c
    sig = None
    if sch.n_nums:
        sig = oops.sigmoid(raw[:, o:o + sch.n_nums].astype(np.float64))
        mn = mask[:, sch.n_bins + sch.n_cats:]
        out["nums"] = (sig * mn).reshape(B, T, -1)
    mu = np.zeros((B, T, 2), np.float32)
    total, parts, d = ovae.loss_fwd(out, target.reshape(B, T, F), mu, mu, 0.0, sch)
    heads = D(raw, dev)
    dh = torch.zeros(rows, NH, device=dev)
    lp = torch.zeros(2 + max(sch.n_cats, 1), device=dev)
    s_bce = 1.0 / (rows * sch.n_bins) if sch.n_bins else 0.0
    s_ce = 1.0 / (rows * sch.n_cats) if sch.n_cats else 0.0
    s_mse = 2.0 / (rows * sch.n_nums) if sch.n_nums else 0.0
    ops.vae_heads_loss(heads, D(mask, dev), D(target, dev), sch.n_bins, sch.n_cats, sch.n_nums, sch.cards, s_bce, s_ce, s_mse, dh, lp)
    h = heads.cpu().numpy()
    g = dh.cpu().numpy()
    p = lp.cpu().numpy().astype(np.float64)
    o = 0
    if sch.n_bins:
        assert fu.rel_err(h[:, o:o + sch.n_bins], out["bins"].reshape(rows, -1)) < 1e-6
        assert fu.rel_err(g[:, o:o + sch.n_bins], d["bins"].reshape(rows, -1)) < 1e-5
        assert abs(p[0] / (rows * sch.n_bins) - parts["bce"]) < 1e-5 * max(parts["bce"], 1e-9)
        o += sch.n_bins
    for j, c in enumerate(sch.cards if sch.n_cats else []):
        assert fu.rel_err(h[:, o:o + c], out["cats"][j].reshape(rows, c)) < 1e-6
        assert np.max(np.abs(g[:, o:o + c] - d["cats"][j].reshape(rows, c))) < 1e-5 * max(np.max(np.abs(d["cats"][j])), 1e-3)
        o += c
    if sch.n_cats:
        assert abs(p[2:2 + sch.n_cats].sum() / (rows * sch.n_cats) - parts["ce"]) < 1e-5 * max(parts["ce"], 1e-9)
    if sch.n_nums:
        assert fu.rel_err(h[:, o:], out["nums"].reshape(rows, -1)) < 1e-6
        mn = mask[:, sch.n_bins + sch.n_cats:]
        assert fu.rel_err(g[:, o:], d["nums"].reshape(rows, -1) * mn * sig * (1 - sig)) < 1e-5
        assert abs(p[1] / (rows * sch.n_nums) - parts["mse"]) < 1e-5 * max(parts["mse"], 1e-9)


@pytest.mark.parametrize("name", ["convert_mixed", "convert_metro", "convert_numeric"])
def test_convert_to_tensor_vs_reference_golden(ops, dev, name):
    """tad_heads_to_tensor through the reference-signature wrapper: bit-exact against the reference's output
    (ties resolve to the first maximum, zeros and near-threshold logits included), and against the oracle on a
    larger seeded case."""
    from oracle import convert as oconv
    from timeautodiff_amd import process_edited as pce
    z, meta = fu.load(name)
    sch = meta["schema"]

    class P:
        def datatype_info(self):
            return dict(sch)
    gen = {}
    if sch["n_bins"]:
        gen["bins"] = D(z["in/bins"], dev)
    if sch["cards"]:
        gen["cats"] = [D(z[f"in/cats{j}"], dev) for j in range(len(sch["cards"]))]
    if sch["n_nums"]:
        gen["nums"] = D(z["in/nums"], dev)
    out = pce.convert_to_tensor(P(), gen, 1, meta["B"], meta["T"])
    assert out.shape == z["out"].shape and np.array_equal(out.cpu().numpy(), z["out"])
    out2 = pce.convert_to_tensor(P(), gen, 1, None, None)
    assert torch.equal(out, out2)
    # larger seeded case against the oracle
    rng = np.random.default_rng(5)
    B, T = 300, 24
    big = {}
    if sch["n_bins"]:
        big["bins"] = rnd(rng, B, T, sch["n_bins"], scale=3.0)
    if sch["cards"]:
        big["cats"] = [rnd(rng, B, T, c) for c in sch["cards"]]
    if sch["n_nums"]:
        big["nums"] = rng.random((B, T, sch["n_nums"]), dtype=np.float32)
    ref = oconv.convert_to_tensor(big, sch["n_bins"], sch["n_cats"], sch["n_nums"], sch["cards"])
    dv = {k: ([D(a, dev) for a in v] if isinstance(v, list) else D(v, dev)) for k, v in big.items()}
    got = pce.convert_to_tensor(P(), dv, 1, B, T).cpu().numpy()
    assert np.array_equal(got, ref)


def test_vae_prepare(ops, dev):
    rng = np.random.default_rng(13)
    a, b = rnd(rng, 100), rnd(rng, 100)
    m = (rng.random(100) < 0.5).astype(np.float32)
    inp = torch.zeros(100, device=dev)
    tgt = torch.zeros(100, device=dev)
    ops.vae_prepare(D(a, dev), D(b, dev), D(m, dev), inp, tgt)
    assert np.array_equal(inp.cpu().numpy(), a + b)
    assert np.array_equal(tgt.cpu().numpy(), (a + b) * m)


def test_bad_arguments_fail_loudly(ops, dev):
    from timeautodiff_amd._lib import TadError
    x = torch.zeros(4, 4, device=dev)
    with pytest.raises(TadError):
        ops.rnn_fwd(0, 0, 4, 1, 2000, [dict(xproj=x, w_packed=x, out=x, gates=None, reverse=False)])   # H too large
    xb = torch.zeros(8, 16, dtype=torch.bfloat16, device=dev)
    with pytest.raises(TadError):
        ops.rnn_fwd(0, 0, 2, 4, 4, [dict(xproj=xb, w_packed=x, out=x, gates=None, reverse=False)])     # bf16 storage in fp32 precision
    with pytest.raises(TadError):
        ops.linear_fwd_b16(x, torch.zeros(4, 4, dtype=torch.bfloat16, device=dev), None, x)             # K not a multiple of 8
    with pytest.raises(TadError):
        ops.linear_fwd(0, torch.zeros(4, 4), x, None, x)                                                  # host tensor


# ------------------------------------------------------------------ edge cases
def test_empty_batch_is_a_no_op(ops, dev):
    """M = 0 (an empty shard of a batch) returns without touching the outputs."""
    w = torch.ones(8, 4, device=dev)
    y = torch.full((0, 8), 7.0, device=dev)
    x = torch.zeros(0, 4, device=dev)
    lib = ops._lib.load()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tad_linear_fwd(0, 0, 8, 4, x.data_ptr() or 1, 4, w.data_ptr(), 4, None, None, None, 0, y.data_ptr() or 1, 8, 0, st) == 0
    dw = torch.full((8, 4), 0.5, device=dev)
    assert lib.tad_linear_bwd_weight(0, 0, 8, 4, y.data_ptr() or 1, 8, x.data_ptr() or 1, 4, dw.data_ptr(), 4, None, None, 0, 0, None, st) == 0
    assert lib.tad_linear_bwd_data(0, 0, 8, 4, y.data_ptr() or 1, 8, w.data_ptr(), 4, x.data_ptr() or 1, 4, None, 0, 0, st) == 0
    torch.cuda.synchronize()
    assert float(dw.min()) == 0.5 and float(dw.max()) == 0.5


@pytest.mark.parametrize("prec", [0, 1])
def test_linear_misaligned_operands_use_the_scalar_loaders(ops, dev, prec):
    """Views that start at an odd element and have odd leading dimensions (no 16-byte access possible)
    give the same result as aligned copies."""
    rng = np.random.default_rng(77)
    M, N, K = 150, 36, 52
    Xb, Wb, Yb = D(rnd(rng, M, K + 3), dev), D(rnd(rng, N, K + 1, scale=0.3), dev), torch.zeros(M, N + 5, device=dev)
    X, W, Y = Xb[:, 1:K + 1], Wb[:, 1:K + 1], Yb[:, 3:N + 3]
    Xr, Wr = X.cpu().numpy(), W.cpu().numpy()
    if prec:
        Xr, Wr = bf16_round(Xr), bf16_round(Wr)
    ref = Xr.astype(np.float64) @ Wr.astype(np.float64).T
    ops.linear_fwd(prec, X, W, None, Y)
    assert fu.rel_err(Y.cpu().numpy(), ref) < F32_TOL
    assert float(Yb[:, :3].abs().max()) == 0.0 and float(Yb[:, N + 3:].abs().max()) == 0.0
    # backward products on the same misaligned views
    dY = Y.clone()                      # contiguous copy is aligned; use the strided view instead
    dX = torch.zeros(M, K + 3, device=dev)
    ops.linear_bwd_data(prec, Y, W, dX[:, 1:K + 1])
    Yr = bf16_round(Y.cpu().numpy()) if prec else Y.cpu().numpy()
    assert fu.rel_err(dX[:, 1:K + 1].cpu().numpy(), Yr.astype(np.float64) @ Wr.astype(np.float64)) < F32_TOL
    dW = torch.zeros(N, K + 1, device=dev)
    db = torch.zeros(N, device=dev)
    ops.linear_bwd_weight(prec, Y, X, dW[:, 1:K + 1], db=db)
    assert fu.rel_err(dW[:, 1:K + 1].cpu().numpy(), Yr.astype(np.float64).T @ Xr.astype(np.float64)) < F32_TOL
    assert fu.rel_err(db.cpu().numpy(), Y.cpu().numpy().astype(np.float64).sum(0)) < F32_TOL
    del dY


@pytest.mark.parametrize("cell", [0, 1])
def test_rnn_single_sequence_and_maximum_hidden_size(ops, dev, cell):
    """B = 1 (one window) and H = 1024 (the largest hidden size the sequence kernels accept), fp32."""
    rng = np.random.default_rng(5 + cell)
    r = _run_rnn(ops, dev, cell, 0, 1, 3, 24, [False, True], rng)
    _check_rnn(r, cell, 1, 3, 24, F32_TOL)
    B, T, H = 3, 2, 1024
    r = _run_rnn(ops, dev, cell, 1, B, T, H, [False], np.random.default_rng(11), IN=16)
    assert r["s16"]
    _check_rnn(r, cell, B, T, H, BF16_EMU_TOL)


# ---------------------------------------------------------------------------- sliding windows on the device
@pytest.mark.parametrize("name", ["windows_stride1", "windows_stride12", "windows_exact"])
def test_sliding_window_vs_reference_golden(ops, dev, name):
    """tad_gather_windows through the reference-signature wrapper and the un-materialised WindowedTable:
    bit-exact against the reference's stacked tensor and its indexed minibatch."""
    from timeautodiff_amd import dataprovider_pypots as dpp
    z, meta = fu.load(name)
    T, stride = meta["seq_len"], meta["stride"]
    stacked = dpp.sliding_window(D(z["table"], dev), T, stride)
    assert stacked.shape == z["stacked"].shape and np.array_equal(stacked.cpu().numpy(), z["stacked"])
    wt = dpp.WindowedTable(torch.from_numpy(z["table"]), T, stride, device=dev)      # host table: copied once
    assert tuple(wt.shape) == z["stacked"].shape and len(wt) == z["stacked"].shape[0]
    batch = wt.gather(torch.from_numpy(z["idx"]))
    assert np.array_equal(batch.cpu().numpy(), z["batch"])
    with pytest.raises(IndexError):
        wt.gather(torch.tensor([len(wt)]))
    # 1-D series and a table whose width rules out 16-byte accesses
    s1 = dpp.sliding_window(D(z["table"][:, 0].copy(), dev), T, stride)
    assert np.array_equal(s1.cpu().numpy(), z["stacked"][:, :, 0])
    odd = dpp.sliding_window(D(z["table"][:, :3].copy(), dev), T, stride)
    assert np.array_equal(odd.cpu().numpy(), z["stacked"][:, :, :3])


def test_gather_windows_fullsize_vs_oracle(ops, dev):
    """Bench-sized table (stride 1, T = 24, 42 columns): every window of a random minibatch equals the oracle's,
    and the windows overlap as they must (row t+1 of window i is row t of window i+1)."""
    from oracle import windows as owin
    from timeautodiff_amd import dataprovider_pypots as dpp
    rng = np.random.default_rng(9)
    rows, F, T = 65536 + 23, 42, 24
    table = rng.standard_normal((rows, F)).astype(np.float32)
    wt = dpp.WindowedTable(D(table, dev), T, 1)
    assert len(wt) == 65536
    idx = rng.integers(0, len(wt), 4096)
    got = wt.gather(torch.from_numpy(idx)).cpu().numpy()
    assert np.array_equal(got, owin.gather_windows(table, idx, T, 1))
    two = wt.gather(torch.tensor([100, 101])).cpu().numpy()
    assert np.array_equal(two[0, 1:], two[1, :-1])
    with pytest.raises(RuntimeError):
        dpp.WindowedTable(D(table[:5], dev), T, 1)           # fewer rows than one window: the reference's stack([]) error


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("shape", [(1000, 400, 6, 200, False), (77, 1024, 4, 512, True), (130, 48, 8, 24, False)])
def test_sampler_tail_equals_the_four_launches(ops, dev, prec, shape):
    """tad_sampler_tail (LayerNorm -> fc -> sampler update -> next input_proj in one pass) against the four entry
    points it replaces, on the same inputs: 1e-5 relative (same roundings, different summation order in the two
    projections)."""
    rows, Dm, K, H, last = shape
    rng = np.random.default_rng(5)
    h = D(rng.standard_normal((rows, Dm)).astype(np.float32) * 3 + 1, dev)
    gamma, beta = D(rng.standard_normal(Dm).astype(np.float32), dev), D(rng.standard_normal(Dm).astype(np.float32), dev)
    fc_w, fc_b = D(rng.standard_normal((K, Dm)).astype(np.float32) * 0.1, dev), D(rng.standard_normal(K).astype(np.float32), dev)
    w_in, b_in = D(rng.standard_normal((H, K)).astype(np.float32), dev), D(rng.standard_normal(H).astype(np.float32), dev)
    x0 = D(rng.standard_normal((rows, K)).astype(np.float32), dev)
    z = D(rng.standard_normal((rows, K)).astype(np.float32), dev)
    c1, c2, c3 = 1.02, 0.31, 0.2
    ln = torch.empty(rows, Dm, device=dev)
    pred = torch.empty(rows, K, device=dev)
    xa, xha = x0.clone(), torch.empty(rows, H, device=dev)
    ops.layernorm_fwd(h, gamma, beta, ln, None)
    ops.linear_fwd(prec, ln, fc_w, fc_b, pred)
    ops.sampler_update(xa, pred, z, c1, c2, c3, last)
    ops.linear_fwd(prec, xa, w_in, b_in, xha)
    xb, xhb = x0.clone(), torch.full((rows, H + 3), 7.0, device=dev)[:, :H]        # strided result rows
    ops.sampler_tail(prec, h, gamma, beta, fc_w, fc_b, xb, z, c1, c2, c3, last, w_in, b_in, xhb)
    tol = 1e-5 if prec == 0 else 2e-3       # bf16: a last-bit difference of the updated x can flip its bf16 rounding
    assert float((xa - xb).abs().max()) <= 1e-5 * float(xa.abs().max())
    assert float((xha - xhb).abs().max()) <= tol * float(xha.abs().max())
    assert torch.all(xhb._base[:, H:] == 7.0)


@pytest.mark.parametrize("rows,Dm,K,H,last", [(300, 400, 6, 200, False), (70, 1024, 6, 512, True), (33, 256, 8, 128, False), (45, 128, 5, 64, True), (50, 128, 6, 200, False), (40, 512, 6, 384, True)])
def test_sampler_tail_with_folded_proj_share(ops, dev, rows, Dm, K, H, last):
    """tad_sampler_tail_proj_b16 -- the tail of a reverse step that also leaves relu(x_new (W_1 W_in)^T + W_1 b_in + V[s] +
    pinv), the LSTM's operand of the NEXT step -- against the two launches it replaces (tad_sampler_tail_b16, then the
    [rows, H] x [H, H] product with bias, residual and ReLU in its epilogue): the updated x bit for bit, the bf16 operand to
    two bf16 units in the last place (the unfolded path rounds x_h to bf16 between the two products; the folded one does not)."""
    rng = np.random.default_rng(8 + H)
    r = lambda *sh, sc=1.0: D((rng.standard_normal(sh) * sc).astype(np.float32), dev)
    h = r(rows, Dm, sc=3.0) + 1
    gamma, beta, fc_w, fc_b = r(Dm), r(Dm), r(K, Dm, sc=0.1), r(K)
    w_in, b_in, w1 = r(H, K), r(H), r(H, H, sc=H ** -0.5)
    vrow, pinv = r(H), r(rows, H)
    x0, z = r(rows, K), r(rows, K)
    c1, c2, c3 = 1.02, 0.31, 0.2
    # unfolded: tail -> x_h (bf16) -> product with the rounded W_1
    xa = x0.clone()
    xh16 = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    ops.sampler_tail(1, h, gamma, beta, fc_w, fc_b, xa, z, c1, c2, c3, last, w_in, b_in, xh16)
    ref = torch.relu(xh16.float() @ w1.bfloat16().float().t() + vrow + pinv)
    # folded: factors rounded as the two products round them, then multiplied in fp32
    w1r = w1.bfloat16().float()
    wc = (w1r @ w_in.bfloat16().float()).contiguous()
    bc = (w1r @ b_in).contiguous()
    xb = x0.clone()
    pa16 = torch.full((rows, H + 8), 7.0, dtype=torch.bfloat16, device=dev)[:, :H]
    ops.sampler_tail_proj(1, h, gamma, beta, fc_w, fc_b, xb, z, c1, c2, c3, last, wc, bc, vrow, pinv, pa16)
    assert torch.equal(xa, xb)
    got = pa16.float()
    assert float((got - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())
    assert float((got - ref).abs().mean()) <= 2.0 ** -9 * float(ref.abs().mean() + 1e-6) * 4
    assert torch.all(pa16._base[:, H:] == 7.0) and torch.all(got >= 0)
    # the same launch drawing the step's noise itself: bit for bit what it gives on the tensor tad_sampler_noise writes,
    # for rows that start anywhere in the global tensor (a chunk, a rank), odd K included
    seed, step, row0 = 0x1234_5678_9ABC_DEF0, 37, 1000 * rows + 3
    zc = torch.empty(rows, K, device=dev)
    ops.sampler_noise(seed, step, row0, rows, K, zc)
    xc, xd = x0.clone(), x0.clone()
    pc = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    pd = torch.empty_like(pc)
    ops.sampler_tail_proj(1, h, gamma, beta, fc_w, fc_b, xc, zc, c1, c2, c3, last, wc, bc, vrow, pinv, pc)
    ops.sampler_tail_proj_cn(1, h, gamma, beta, fc_w, fc_b, xd, seed, step, row0, c1, c2, c3, last, wc, bc, vrow, pinv, pd)
    assert torch.equal(xc, xd) and torch.equal(pc, pd)


# ---------------------------------------------------------------------------- TV-MCG metadata corruption
@pytest.mark.parametrize("name", ["tvmcg_mixed", "tvmcg_skip"])
def test_tvmcg_corruption_vs_reference_golden(ops, dev, name):
    """tad_metadata_noise / tad_flip_categories through the reference-signature wrappers, on the draws the reference
    took.  Flipped codes and untouched columns: bit-exact.  Noisy columns: 1e-6 relative (the column standard
    deviation is a float64 reduction rounded to float32; summation order may move its last bit)."""
    from timeautodiff_amd import dataprovider_pypots as dpp
    z, meta = fu.load(name)
    classes = {int(k): v for k, v in meta["classes"].items()}
    cond = D(z["cond"], dev)
    noisy = dpp.add_continuous_metadata_noise_subset(cond, meta["cont_idx"], meta["level"], noise=torch.from_numpy(z["noise"]))
    got = noisy.cpu().numpy()
    np.testing.assert_allclose(got, z["noisy"], rtol=1e-6, atol=1e-6)
    untouched = [c for c in range(meta["F"]) if c not in meta["cont_idx"]]
    assert np.array_equal(got[:, :, untouched], z["cond"][:, :, untouched])
    flipped = dpp.flip_categorical_metadata_subset(cond, meta["cat_idx"], meta["p"], classes, rand=list(torch.from_numpy(z["rand"])),
                                                   base=list(torch.from_numpy(z["base"])))
    assert np.array_equal(flipped.cpu().numpy(), z["flipped"])
    # the reference's early returns hand back the argument itself
    assert dpp.add_continuous_metadata_noise_subset(cond, meta["cont_idx"], 0.0) is cond
    assert dpp.add_continuous_metadata_noise_subset(cond, [], 0.3) is cond
    assert dpp.flip_categorical_metadata_subset(cond, meta["cat_idx"], 0.0, classes) is cond
    assert dpp.flip_categorical_metadata_subset(cond, [], 0.5, classes) is cond
    with pytest.raises(KeyError):
        dpp.flip_categorical_metadata_subset(cond, [0], 0.5, {})


def test_tvmcg_corruption_fullsize_vs_oracle(ops, dev):
    """Bench-sized cond (4096 x 24 x 28): oracle parity on injected draws, and the properties of device-drawn
    corruption: only the listed columns change, flipped codes are always another valid class, the flip rate is p,
    the added noise has level * std per column."""
    from oracle import tvmcg as otv
    from timeautodiff_amd import dataprovider_pypots as dpp
    rng = np.random.default_rng(17)
    B, T, F = 4096, 24, 28
    cond = rng.standard_normal((B, T, F)).astype(np.float32) * rng.uniform(0.1, 30, F).astype(np.float32) + 5
    cat_idx, classes = [0, 9], {0: 8, 9: 2}
    for c in cat_idx:
        cond[:, :, c] = rng.integers(0, classes[c], (B, T)).astype(np.float32)
    cont_idx = [3, 4, 27, 11]
    noise = rng.standard_normal((B, T, len(cont_idx))).astype(np.float32)
    cd = D(cond, dev)
    got = dpp.add_continuous_metadata_noise_subset(cd, cont_idx, 0.4, noise=torch.from_numpy(noise)).cpu().numpy()
    np.testing.assert_allclose(got, otv.add_continuous_metadata_noise_subset(cond, cont_idx, 0.4, noise), rtol=1e-6, atol=1e-6)
    rnd = rng.random((2, B, T)).astype(np.float32)
    base = np.stack([rng.integers(0, classes[c] - 1, (B, T)) for c in cat_idx]).astype(np.int64)
    gotf = dpp.flip_categorical_metadata_subset(cd, cat_idx, 0.2, classes, rand=list(torch.from_numpy(rnd)), base=list(torch.from_numpy(base)))
    assert np.array_equal(gotf.cpu().numpy(), otv.flip_categorical_metadata_subset(cond, cat_idx, 0.2, classes, rnd, base))
    # device draws
    g = torch.Generator(device=dev).manual_seed(3)
    nz = dpp.add_continuous_metadata_noise_subset(cd, cont_idx, 0.4, generator=g).cpu().numpy()
    other = [c for c in range(F) if c not in cont_idx]
    assert np.array_equal(nz[:, :, other], cond[:, :, other])
    d = (nz - cond)[:, :, cont_idx].reshape(-1, len(cont_idx))
    want = 0.4 * cond[:, :, cont_idx].reshape(-1, len(cont_idx)).std(0, ddof=1)
    np.testing.assert_allclose(d.std(0), want, rtol=2e-2)
    fl = dpp.flip_categorical_metadata_subset(cd, cat_idx, 0.2, classes, generator=g).cpu().numpy()
    other = [c for c in range(F) if c not in cat_idx]
    assert np.array_equal(fl[:, :, other], cond[:, :, other])
    for c in cat_idx:
        changed = fl[:, :, c] != cond[:, :, c]
        assert abs(changed.mean() - 0.2) < 0.01
        assert fl[:, :, c].min() >= 0 and fl[:, :, c].max() <= classes[c] - 1 and np.array_equal(fl[:, :, c], np.trunc(fl[:, :, c]))


@pytest.mark.parametrize("prec", [0, 1])
def test_linear_bwd_weight_skinny_strided_operands(ops, dev, prec):
    """Extents <= 8 on either side (input_proj, fc, time_encode) with operands that are column slices of wider
    buffers (leading dimension > extent), result accumulated into a slice of a wider dW."""
    rng = np.random.default_rng(31)
    M = 1000
    for N, K in ((6, 130), (130, 8)):
        dYb, Xb = rnd(rng, M, N + 11), rnd(rng, M, K + 5)
        dY, X = dYb[:, 3:3 + N], Xb[:, 2:2 + K]
        dYr, Xr = (bf16_round(dY), bf16_round(X)) if prec else (dY, X)
        ref = dYr.astype(np.float64).T @ Xr.astype(np.float64)
        dWb = torch.full((N, K + 7), 0.25, device=dev)
        db = torch.zeros(N, device=dev)
        ops.linear_bwd_weight(prec, D(dYb, dev)[:, 3:3 + N], D(Xb, dev)[:, 2:2 + K], dWb[:, 4:4 + K], db=db)
        got = dWb.cpu().numpy()
        assert fu.rel_err(got[:, 4:4 + K], ref + 0.25) < F32_TOL
        assert np.all(got[:, :4] == 0.25) and np.all(got[:, 4 + K:] == 0.25)
        assert fu.rel_err(db.cpu().numpy(), dY.astype(np.float64).sum(0)) < F32_TOL


@pytest.mark.parametrize("n", [1, 3, 4, 1027, 3476296])
def test_grad_norm_sq(ops, dev, n):
    """Sum of squares of a flat buffer against a float64 sum; accumulates into its output (fp32 partial sums: 1e-5)."""
    g = torch.randn(n + 4, device=dev)[4:]                 # 16-byte aligned, odd lengths exercise the tail
    out = torch.full((1,), 2.5, device=dev)
    ops.grad_norm_sq(g, out)
    torch.cuda.synchronize()
    ref = 2.5 + float((g.double() ** 2).sum())
    assert abs(float(out.item()) - ref) < 1e-5 * ref


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("rows,D,K", [(7, 48, 6), (1000, 400, 6), (333, 400, 8), (64, 130, 3), (700, 1024, 6), (90, 600, 8)])
def test_ln_fc_fused_equals_the_separate_launches(ops, dev, prec, rows, D, K):
    """LayerNorm + narrow projection in one launch, forward and backward, against layernorm_fwd/bwd + the linear family
    (the same arithmetic per element; the sums of the gradients take another order: 1e-5 fp32; bf16 mode rounds the
    projection's operands at the same points: 2e-3 on the weight gradient whose products it rounds)."""
    g = torch.Generator(device=dev).manual_seed(rows + D + K)
    x = torch.randn(rows, D, device=dev, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(D, device=dev, generator=g) + 0.5, torch.randn(D, device=dev, generator=g) * 0.1
    W, b = torch.randn(K, D, device=dev, generator=g) * 0.1, torch.randn(K, device=dev, generator=g)
    dpred = torch.randn(rows, K, device=dev, generator=g)
    # separate launches
    y, mr, pred = torch.empty(rows, D, device=dev), torch.empty(rows, 2, device=dev), torch.empty(rows, K, device=dev)
    ops.layernorm_fwd(x, gamma, beta, y, mr)
    ops.linear_fwd(prec, y, W, b, pred)
    dy, dx = torch.empty(rows, D, device=dev), torch.empty(rows, D, device=dev)
    dW, db, dga, dbe = (torch.zeros(K, D, device=dev), torch.zeros(K, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev))
    ops.linear_bwd_data(prec, dpred, W, dy)
    ops.linear_bwd_weight(prec, dpred, y, dW, db=db)
    ops.layernorm_bwd(dy, x, gamma, mr, dx, dga, dbe)
    # fused
    mr2, pred2, y2 = torch.empty(rows, 2, device=dev), torch.empty(rows, K, device=dev), torch.empty(rows, D, device=dev)
    ops.ln_fc_fwd(prec, x, gamma, beta, W, b, mr2, pred2, y=y2)
    dx2 = torch.empty(rows, D, device=dev)
    dW2, db2, dga2, dbe2 = (torch.full((K, D), 1.0, device=dev), torch.full((K,), 1.0, device=dev), torch.full((D,), 1.0, device=dev),
                            torch.full((D,), 1.0, device=dev))                # accumulated into
    ops.ln_fc_bwd(prec, dpred, x, gamma, beta, mr2, W, dx2, dga2, dbe2, dW2, db2)
    torch.cuda.synchronize()
    rel = lambda a, b_: float((a.double() - b_.double()).norm() / (b_.double().norm() + 1e-30))   # noqa: E731
    tol = 1e-5 if prec == 0 else 2e-3
    assert torch.equal(mr, mr2) and torch.equal(y, y2)
    assert rel(pred2, pred) < tol and rel(dx2, dx) < tol
    assert rel(dW2 - 1, dW) < tol and rel(db2 - 1, db) < 1e-5 and rel(dga2 - 1, dga) < tol and rel(dbe2 - 1, dbe) < tol


@pytest.mark.parametrize("N,B,T,K", [(1000, 64, 24, 6), (64, 64, 24, 6), (5, 3, 7, 1), (1, 1, 1, 1), (70001, 4096, 24, 6), (300, 100, 96, 3),
                                       (2000, 33, 200, 5)])
def test_draw_batch_vs_oracle(ops, dev, N, B, T, K):
    """tad_draw_batch against its numpy restatement: window indices, diffusion steps and the sorted uniforms bit-exact,
    normals to 2e-6 absolute (log / sin / cos of the device library against numpy's); a second call with the same (seed,
    counter) repeats the batch, another counter gives another."""
    from oracle import draws
    seed, counter = 0x1234567890ABCDEF, 41
    idx = torch.empty(B, dtype=torch.int64, device=dev)
    t = torch.empty(B, T, 1, device=dev)
    i = torch.empty(B, dtype=torch.int32, device=dev)
    noise = torch.empty(B, T, K, device=dev)
    ops.draw_batch(seed, counter, N, B, T, K, 100, idx=idx, t=t, i=i, noise=noise)
    ref = draws.draw_batch(seed, counter, N, B, T, K, 100)
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])
    assert np.array_equal(i.cpu().numpy(), ref["i"])
    assert np.array_equal(t.cpu().numpy()[:, :, 0], ref["t"])
    assert np.abs(noise.cpu().numpy() - ref["noise"]).max() < 2e-6 * max(1.0, np.abs(ref["noise"]).max())
    idx2, n2 = torch.empty_like(idx), torch.empty_like(noise)
    ops.draw_batch(seed, counter, N, B, T, K, 100, idx=idx2, noise=n2)           # outputs left out are not drawn
    assert torch.equal(idx, idx2) and torch.equal(noise, n2)
    ops.draw_batch(seed, counter + 1, N, B, T, K, 100, idx=idx2, noise=n2)
    assert not torch.equal(noise, n2)
    with pytest.raises(Exception):
        ops.draw_batch(seed, counter, B - 1, B, T, K, 100, idx=idx)             # fewer windows than the sample


@pytest.mark.parametrize("rows,E,cards", [(5000, 128, [8]), (1024, 16, [2, 7, 1]), (3001, 100, [16, 3]), (2000, 128, [17]), (4000, 160, [4])])
def test_embed_bwd_forms(ops, dev, rows, E, cards):
    """Embedding-table gradients at sizes that take the register-accumulating kernel (card <= 16, E <= 128, 1024+ rows)
    and at sizes that keep the LDS-atomic one, against a float64 scatter-add; codes outside [0, card) clamp like the
    forward's."""
    rng = np.random.default_rng(rows + E)
    nd = len(cards)
    x = np.stack([rng.integers(-1, c + 1, rows) for c in cards], 1).astype(np.float32)        # some codes out of range
    off = list(np.cumsum([0] + cards[:-1]))
    dout = rnd(rng, rows, nd * E)
    dt = torch.zeros(sum(cards), E, device=dev)
    ops.embed_bwd(D(x, dev), D(dout, dev), dt, off, cards, E)
    gref = np.zeros((sum(cards), E), dtype=np.float64)
    for d in range(nd):
        np.add.at(gref, off[d] + np.clip(x[:, d].astype(int), 0, cards[d] - 1), dout[:, d * E:(d + 1) * E])
    assert fu.rel_err(dt.cpu().numpy(), gref) < 1e-5


@pytest.mark.parametrize("rows,K,row0", [(1, 1, 0), (1000, 6, 0), (777, 5, 12345), (4096 * 24, 6, 2 ** 33)])
def test_sampler_noise_vs_oracle(ops, dev, rows, K, row0):
    """tad_sampler_noise against its numpy restatement (2e-6 absolute: log / sin / cos), and a slice of the rows equals
    the same rows drawn on their own (what makes sharded sampling independent of the sharding)."""
    from oracle import draws
    seed, step = 0xFEDCBA9876543210, 57
    out = torch.empty(rows, K, device=dev)
    ops.sampler_noise(seed, step, row0, rows, K, out)
    ref = draws.sampler_noise(seed, step, row0, rows, K)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
    h = rows // 3
    part = torch.empty(rows - h, K, device=dev)
    ops.sampler_noise(seed, step, row0 + h, rows - h, K, part)
    assert torch.equal(part, out[h:])
    other = torch.empty(rows, K, device=dev)
    ops.sampler_noise(seed, step + 1, row0, rows, K, other)
    assert rows * K < 4 or not torch.equal(other, out)


def test_zero_copy_add_helpers(ops, dev):
    """The fills / copies of a step (zero_grad, pad columns, the bias of a fused product) against torch."""
    for n in (1, 3, 4, 5, 1027, 3476297):
        g = torch.randn(n + 8, device=dev)
        ops.zero_(g[4:4 + n])                            # 16-byte aligned start
        assert float(g[4:4 + n].abs().sum()) == 0 and float(g[:4].abs().min()) > 0 and float(g[4 + n:].abs().min()) > 0
        g = torch.randn(n + 8, device=dev)
        ops.zero_(g[1:1 + n])                            # unaligned start
        assert float(g[1:1 + n].abs().sum()) == 0 and float(g[:1].abs().min()) > 0 and float(g[1 + n:].abs().min()) > 0
    m = torch.randn(37, 40, device=dev)
    keep = m.clone()
    ops.zero_(m[:, 8:20])
    keep[:, 8:20] = 0
    assert torch.equal(m, keep)
    mb = torch.randn(9, 16, device=dev).bfloat16()
    ops.zero_(mb[:, 4:8])
    assert float(mb[:, 4:8].float().abs().sum()) == 0 and float(mb[:, :4].float().abs().min()) > 0
    src = torch.randn(50, 28, device=dev)
    dst = torch.full((50, 32), 7.0, device=dev)
    ops.copy_2d(src, dst[:, :28])
    assert torch.equal(dst[:, :28], src) and float((dst[:, 28:] - 7).abs().sum()) == 0
    a, b, out = torch.randn(401, device=dev), torch.randn(401, device=dev), torch.empty(401, device=dev)
    ops.add_vec(a, b, out)
    assert torch.equal(out, a + b)
