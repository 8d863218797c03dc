"""The bf16-stored-operand matrix products (tad_mm16_nt: forward / input gradient with fused epilogue;
tad_mm16_tn: fused weight-gradient job lists) against float64 products of the same bf16-rounded operands.
Tolerances: bf16 output 4e-3 relative (one rounding of the result, 2^-9), fp32 output 1e-5, weight gradients 2e-5
(fp32 accumulation in a different order than the float64 reference)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from timeautodiff_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _nt(ops, dev, M, N, K, c16, act=0, res=None, relu=None, bias=True):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    Wp = ops.w16_image(N, K, dev)
    Wp[N:].fill_(float("nan"))          # image rows beyond N may hold anything
    Wp[:N, :K] = W
    b = torch.randn(N, device=dev, generator=g) if bias else None
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    R = torch.randn(M, N, device=dev, generator=g) if res else None
    if res == "bf16":
        R = R.bfloat16()
    U = torch.randn(M, N, device=dev, generator=g) if relu else None
    if relu == "bf16":
        U = U.bfloat16()
    Z = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if act == 2 else torch.float32) if act in (2, 4) else None
    ops.mm16_nt(A, Wp, b, C, act=act, residual=R, relu_of=U, z_out=Z)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    if bias:
        ref = ref + b.double()
    if R is not None:
        ref = ref + R.double()
    zref = ref.clone()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act in (2, 4):
        ref = ref * torch.sigmoid(ref)
    if U is not None and act == 3:
        sg = torch.sigmoid(U.double())
        ref = ref * sg * (1 + U.double() * (1 - sg))
    elif U is not None:
        ref = ref * (U.double() > 0)
    assert not torch.isnan(C.float()).any()
    assert _rel(C, ref) < (4e-3 if c16 else 1e-5)
    if Z is not None:
        assert _rel(Z, zref) < (4e-3 if act == 2 else 1e-5)


NT_SHAPES = [(5, 8, 8), (130, 72, 40), (257, 200, 208), (1000, 1600, 400), (300, 128, 448), (64, 432, 432), (77, 200, 800),
             (513, 560, 1024), (4100, 1600, 128)]


@pytest.mark.parametrize("c16", [True, False])
@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_mm16_nt_plain(ops, dev, M, N, K, c16):
    _nt(ops, dev, M, N, K, c16)
    _nt(ops, dev, M, N, K, c16, bias=False)


@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_mm16_nt_fused_epilogues(ops, dev, M, N, K):
    _nt(ops, dev, M, N, K, True, act=1, res="f32", relu="bf16")
    _nt(ops, dev, M, N, K, False, act=2, res="bf16", relu="f32")
    _nt(ops, dev, M, N, K, True, act=2)
    _nt(ops, dev, M, N, K, True, act=4)                               # SiLU with float32 pre-activations beside the bf16 result
    _nt(ops, dev, M, N, K, True, act=3, relu="f32", bias=False)      # the backward of a SiLU as the product's epilogue
    _nt(ops, dev, M, N, K, False, act=3, relu="f32", bias=False)


@pytest.mark.parametrize("M,N,K,tiled", [(300, 200, 800, True), (300, 128, 560, True), (300, 512, 512, True), (300, 1600, 400, False),
                                          (300, 200, 200, False), (300, 800, 512, False), (300, 1600, 800, True)])
def test_mm16_nt_plain_products_with_long_reductions_take_the_tiled_kernel(ops, dev, monkeypatch, M, N, K, tiled):
    """ops.mm16_nt hands products with a plain epilogue to the square-tiled kernel from K = 768 on, and from K = 384 on
    up to 512 result columns (a choice by N and K alone: the same kernel serves every row count); any fused operand
    keeps tad_mm16_nt's own kernel.  Both kernels against the float64 product, and against each other to bf16 rounding."""
    last = lambda: ops._lib.load().tad_last_kernel().decode()      # noqa: E731
    g = torch.Generator(device=dev).manual_seed(N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Wp = ops.w16_image(N, K, dev)
    Wp[:N, :K] = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    ref = (A.double() @ Wp[:N, :K].double().T + b.double()).clamp_min(0)
    C1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.mm16_nt(A, Wp, b, C1, act=1)
    assert last().startswith("gemm_kernel" if tiled else "mm16_nt_kernel"), last()
    monkeypatch.setattr(ops, "NT_TILED", False)
    C2 = torch.empty_like(C1)
    ops.mm16_nt(A, Wp, b, C2, act=1)
    assert last().startswith("mm16_nt_kernel"), last()
    monkeypatch.setattr(ops, "NT_TILED", True)
    R = torch.randn(M, N, device=dev, generator=g)
    C3 = torch.empty_like(C1)
    ops.mm16_nt(A, Wp, b, C3, act=1, residual=R)
    assert last().startswith("mm16_nt_kernel"), last()
    torch.cuda.synchronize()
    assert _rel(C1, ref) < 4e-3 and _rel(C2, ref) < 4e-3 and _rel(C1, C2) < 4e-3


def test_mm16_nt_dsilu_equals_product_then_silu_bwd(ops, dev):
    """act = DSILU against the two launches it replaces (product into float32, then tad_silu_bwd): the same values up to
    the sigmoid's last bits (hardware exp / reciprocal in the epilogue), i.e. at most one bf16 step apart."""
    M, N, K = 1000, 432, 432
    g = torch.Generator(device=dev).manual_seed(5)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Wp = ops.w16_image(N, K, dev)
    Wp[:N, :K] = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    z = torch.randn(M, N, device=dev, generator=g) * 3
    da, dz_two, dz_one = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.mm16_nt(A, Wp, None, da)
    ops.silu_bwd(z, da, dz_two)
    ops.mm16_nt(A, Wp, None, dz_one, act=3, relu_of=z)
    a, b = dz_one.float(), dz_two.float()
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()) and _rel(a, b) < 1e-3
    assert float((a != b).float().mean()) < 0.02
    with pytest.raises(Exception):
        ops.mm16_nt(A, Wp, None, dz_one, act=3)                        # no pre-activations given


def test_mm16_nt_rejects_bad_operands(ops, dev):
    A = torch.zeros(16, 40, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(16, 8, device=dev)
    with pytest.raises((RuntimeError, AssertionError)):    # weight not a padded image (row stride below 128)
        ops.mm16_nt(A, torch.zeros(64, 40, device=dev, dtype=torch.bfloat16), None, C)
    with pytest.raises((RuntimeError, AssertionError)):    # K not a multiple of 8
        ops.mm16_nt(A[:, :36], ops.w16_image(8, 36, dev), None, C)
    ops.mm16_nt(A[:0], ops.w16_image(8, 40, dev), None, C[:0])       # empty batch: no launch


def _tn(ops, dev, M, Na, spec, T=0):
    """spec: list of (a0, na, nb, shift, gate_map, with_db)."""
    g = torch.Generator(device=dev).manual_seed(M + Na)
    A = torch.randn(M, Na, device=dev, generator=g).bfloat16()
    jobs = []
    for (a0, na, nb, shift, gm, with_db) in spec:
        B = torch.randn(M, nb, device=dev, generator=g).bfloat16()
        n_out = na if gm is None else na // 4 * gm[1]
        jobs.append(dict(b=B, c=torch.zeros(n_out, nb, device=dev), a0=a0, na=na, shift=shift, gate_map=gm,
                         db=torch.zeros(n_out, device=dev) if with_db else None))
    ops.mm16_tn(A, jobs, T=T)
    torch.cuda.synchronize()
    for j in jobs:
        a = A[:, j["a0"]:j["a0"] + j["na"]].double()
        b = j["b"].double()
        if j["shift"]:
            tt = torch.arange(M, device=dev) % T
            ok = ((tt + j["shift"]) >= 0) & ((tt + j["shift"]) < T)
            src = (torch.arange(M, device=dev) + j["shift"]).clamp(0, M - 1)
            bs = torch.zeros_like(b)
            bs[ok] = b[src[ok]]
            b = bs
        full, dbf = a.T @ b, a.sum(0)
        gm = j["gate_map"]
        if gm is not None:
            H, G, slots = gm
            n = torch.arange(j["na"], device=dev)
            s_, q_ = n & 3, n >> 2
            sl = torch.tensor(slots, device=dev)[s_]
            dst = (q_ // H) * G * H + sl * H + q_ % H
            keep = sl >= 0
            ref = torch.zeros(j["c"].shape, device=dev, dtype=torch.float64).index_add_(0, dst[keep], full[keep])
            dref = torch.zeros(j["c"].shape[0], device=dev, dtype=torch.float64).index_add_(0, dst[keep], dbf[keep])
        else:
            ref, dref = full, dbf
        assert _rel(j["c"], ref) < 2e-5
        if j["db"] is not None:
            assert _rel(j["db"], dref) < 2e-5


@pytest.mark.parametrize("M,T", [(24, 8), (216, 24), (1000, 0), (4096, 8), (98304 // 16, 24)])
def test_mm16_tn_job_lists(ops, dev, M, T):
    H = 8
    lstm = (H, 4, (0, 1, 2, 3))
    gru_in, gru_hid = (H, 3, (0, 1, 2, -1)), (H, 3, (0, 1, -1, 2))
    _tn(ops, dev, M, 64, [(0, 64, 40, 0, None, True)], T=T)
    _tn(ops, dev, M, 200, [(0, 200, 136, 0, None, False), (8, 136, 200, 0, None, True)], T=0)
    _tn(ops, dev, M, 608, [(0, 608, 560, 0, None, True)], T=0)          # more than one tile either way
    if T and M % T == 0:
        _tn(ops, dev, M, 64, [(0, 64, 16, 0, lstm, True), (0, 32, 8, -1, lstm, False), (32, 32, 8, 1, lstm, False)], T=T)
        _tn(ops, dev, M, 64, [(0, 64, 24, 0, gru_in, True), (0, 32, 8, -1, gru_hid, True), (32, 32, 8, 1, gru_hid, True)], T=T)


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_mm16_tn_recurrent_layer_launch_at_model_width(ops, dev, cell):
    """The weight-gradient launch of a recurrent layer at the bench model's widths (H = 200: A has 1600 columns = 6 tiles
    and a quarter, the recurrent jobs 800 = 3 tiles and an eighth, B 400 / 200 columns), with enough rows that the reduction
    is split and the splits are sized tile by tile: gate maps, time shifts and bias sums through the edge tiles."""
    H, T, M = 200, 24, 24576
    if cell == "lstm":
        gi = gh = (H, 4, (0, 1, 2, 3))
    else:
        gi, gh = (H, 3, (0, 1, 2, -1)), (H, 3, (0, 1, -1, 2))
    _tn(ops, dev, M, 1600, [(0, 1600, 400, 0, gi, True), (0, 800, 200, -1, gh, True), (800, 800, 200, 1, gh, False)], T=T)
    _tn(ops, dev, M, 1600, [(0, 1600, 128, 0, gi, True), (0, 800, 200, -1, gh, False), (800, 800, 200, 1, gh, True)], T=T)


def test_mm16_tn_accumulates_and_rejects(ops, dev):
    A = torch.ones(64, 8, device=dev, dtype=torch.bfloat16)
    B = torch.ones(64, 8, device=dev, dtype=torch.bfloat16)
    C = torch.full((8, 8), 3.0, device=dev)
    ops.mm16_tn(A, [dict(b=B, c=C)])
    torch.cuda.synchronize()
    assert torch.equal(C, torch.full((8, 8), 67.0, device=dev))         # += , not =
    with pytest.raises((RuntimeError, AssertionError)):
        ops.mm16_tn(A, [dict(b=B[:, :4], c=C[:, :4].contiguous())])     # nb not a multiple of 8
    with pytest.raises((RuntimeError, AssertionError, IndexError)):
        ops.mm16_tn(A, [dict(b=B, c=C)] * 13)                           # more than twelve jobs


def test_mm16_tn_jobs_with_their_own_a_operands(ops, dev):
    """Eight small products of different layers (different A operands and widths, the same number of rows) in one
    launch, as the backward pass queues them; one job still uses the shared operand."""
    M = 3000
    g = torch.Generator(device=dev).manual_seed(77)
    shared = torch.randn(M, 48, device=dev, generator=g).bfloat16()
    jobs, refs = [], []
    for k, (na, nb) in enumerate([(200, 800), (200, 200), (128, 128), (128, 560), (432, 432), (96, 96), (8, 8), (48, 24)]):
        A = shared if k == 7 else torch.randn(M, na + 8, device=dev, generator=g).bfloat16()
        B = torch.randn(M, nb, device=dev, generator=g).bfloat16()
        c, db = torch.zeros(na, nb, device=dev), torch.zeros(na, device=dev)
        jobs.append(dict(b=B, c=c, db=db, a0=0, na=na, **({} if k == 7 else dict(a=A))))
        refs.append((A[:, :na].double().T @ B.double(), A[:, :na].double().sum(0)))
    ops.mm16_tn(shared, jobs)
    torch.cuda.synchronize()
    for j, (rc, rb) in zip(jobs, refs):
        assert _rel(j["c"], rc) < 2e-5 and _rel(j["db"], rb) < 2e-5
    with pytest.raises((RuntimeError, AssertionError)):
        ops.mm16_tn(None, [dict(b=jobs[0]["b"], c=jobs[0]["c"])])          # no A operand at all


def test_mm16_tn_weight_narrower_than_its_padded_operand(ops, dev):
    """Linear(6 -> 200): the input is stored padded to 8 columns (zeros), the weight gradient is [200, 6]; the product's
    padding columns must not be written (they would land in the next row of the gradient)."""
    M = 5000
    g = torch.Generator(device=dev).manual_seed(5)
    A = torch.randn(M, 200, device=dev, generator=g).bfloat16()
    B = torch.zeros(M, 16, device=dev, dtype=torch.bfloat16)
    B[:, :6] = torch.randn(M, 6, device=dev, generator=g).bfloat16()
    B[:, 8:14] = 1.0                                                       # neighbouring columns of a wider buffer: not part of b
    c = torch.full((200, 6), 0.5, device=dev)
    db = torch.zeros(200, device=dev)
    ops.mm16_tn(A, [dict(b=B[:, :8], c=c, db=db)])
    torch.cuda.synchronize()
    ref = 0.5 + A.double().T @ B[:, :6].double()
    assert _rel(c, ref) < 2e-5 and _rel(db, A.double().sum(0)) < 2e-5
    with pytest.raises((RuntimeError, AssertionError)):
        ops.mm16_tn(A, [dict(b=B, c=c)])                                   # 10 columns short: not padding any more
