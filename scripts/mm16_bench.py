#!/usr/bin/env python3
"""Correctness (against a float64 product of the bf16-rounded operands) and timing of the bf16-stored-operand
products (tad_mm16_nt / tad_mm16_tn) at the shapes of the training step, beside the fp32-stored-operand kernels.
usage: python scripts/mm16_bench.py [nt|tn|all] [M]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from timeautodiff_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nt_case(M, N, K, c16=True, bias=True, act=0, res=None, relu=None, check_rows=2048):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=dev, generator=g)).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.1).bfloat16()
    Wp = ops.w16_image(N, K, dev)
    Wp[N:].fill_(float("nan"))          # rows beyond N may hold anything
    Wp[:N, :K] = W
    b = torch.randn(N, device=dev, generator=g) if bias else None
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    R = torch.randn(M, N, device=dev, generator=g) if res else None
    if res == "bf16":
        R = R.bfloat16()
    U = torch.randn(M, N, device=dev, generator=g) if relu else None
    if relu == "bf16":
        U = U.bfloat16()
    Z = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) if act == 2 else None

    def run():
        ops.mm16_nt(A, Wp, b, C, act=act, residual=R, relu_of=U, z_out=Z)
    run()
    torch.cuda.synchronize()
    rows = min(M, check_rows)
    sel = torch.cat([torch.arange(rows // 2, device=dev), torch.arange(M - (rows - rows // 2), M, device=dev)])
    ref = A[sel].double() @ W.double().T
    if bias:
        ref = ref + b.double()
    if R is not None:
        ref = ref + R[sel].double()
    zref = ref.clone()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    if U is not None:
        ref = ref * (U[sel].double() > 0)
    err = rel(C[sel], ref)
    zerr = rel(Z[sel], zref) if Z is not None else 0.0
    tol = 4e-3 if c16 else 1e-5
    assert err < tol and zerr < 4e-3, (M, N, K, err, zerr)
    assert not torch.isnan(C.float()).any()
    us = timeit(run)
    fl = 2.0 * M * N * K
    by = 2.0 * M * K + 2.0 * N * K + C.element_size() * M * N
    return us, fl / us / 1e6, by / us / 1e3, err


def tn_case(M, Na, jobs_spec, T=0, check=True, reps=20):
    """jobs_spec: list of (a0, na, nb, shift, gate_map, with_db)."""
    g = torch.Generator(device=dev).manual_seed(M + Na)
    A = torch.randn(M, Na, device=dev, generator=g).bfloat16()
    jobs, refs = [], []
    for (a0, na, nb, shift, gm, with_db) in jobs_spec:
        B = torch.randn(M, nb, device=dev, generator=g).bfloat16()
        n_out = na if gm is None else na // 4 * gm[1]
        Cm = torch.zeros(n_out, nb, device=dev)
        db = torch.zeros(n_out, device=dev) if with_db else None
        jobs.append(dict(b=B, c=Cm, a0=a0, na=na, shift=shift, gate_map=gm, db=db))
    ops.mm16_tn(A, jobs, T=T)
    torch.cuda.synchronize()
    worst = 0.0
    if check:
        for j in jobs:
            a = A[:, j["a0"]:j["a0"] + j["na"]].double()
            b = j["b"].double()
            if j["shift"]:
                bs = torch.zeros_like(b)
                tt = torch.arange(M, device=dev) % T
                ok = ((tt + j["shift"]) >= 0) & ((tt + j["shift"]) < T)
                src = (torch.arange(M, device=dev) + j["shift"]).clamp(0, M - 1)
                bs[ok] = b[src[ok]]
                b = bs
            full = a.T @ b
            dbf = a.sum(0)
            gm = j["gate_map"]
            if gm is not None:
                H, G, slots = gm
                n = torch.arange(j["na"], device=dev)
                s_, q_ = n & 3, n >> 2
                blk, u = q_ // H, q_ % H
                sl = torch.tensor(slots, device=dev)[s_]
                dst = blk * G * H + sl * H + u
                keep = sl >= 0
                ref = torch.zeros(j["c"].shape, device=dev, dtype=torch.float64)
                ref.index_add_(0, dst[keep], full[keep])
                dref = torch.zeros(j["c"].shape[0], device=dev, dtype=torch.float64)
                dref.index_add_(0, dst[keep], dbf[keep])
            else:
                ref, dref = full, dbf
            worst = max(worst, rel(j["c"], ref))
            if j["db"] is not None:
                worst = max(worst, rel(j["db"], dref))
        assert worst < 2e-5, (M, Na, jobs_spec, T, worst)
    us = timeit(lambda: ops.mm16_tn(A, jobs, T=T), reps=reps)
    fl = sum(2.0 * M * j["na"] * j["b"].shape[1] for j in jobs)
    by = 2.0 * M * Na + sum(2.0 * M * j["b"].shape[1] for j in jobs)
    return us, fl / us / 1e6, by / us / 1e3, worst


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
    if what in ("nt", "all"):
        # small / ragged correctness cases
        for (m, n, k) in [(5, 8, 8), (130, 72, 40), (257, 200, 208), (1000, 1600, 400), (300, 128, 448), (64, 432, 432), (77, 200, 800), (513, 560, 1024)]:
            for c16 in (True, False):
                nt_case(m, n, k, c16=c16, check_rows=m)
            nt_case(m, n, k, c16=True, act=1, res="f32", relu="bf16", check_rows=m)
            nt_case(m, n, k, c16=False, act=2, res="bf16", relu="f32", check_rows=m)
        print("mm16_nt correctness ok")
        print(f"{'M':>7} {'N':>5} {'K':>5} out   {'us':>8} {'TFLOP/s':>8} {'GB/s':>8}  | old kernel us (fp32 A)")
        for (n, k) in [(1600, 400), (1600, 200), (1600, 128), (800, 200), (200, 800), (432, 432), (128, 560), (400, 800), (200, 400), (128, 128)]:
            for c16 in (True, False):
                us, tf, gb, err = nt_case(M, n, k, c16=c16)
                A32 = torch.randn(M, k, device=dev)
                W16 = (torch.randn(n, k, device=dev) * 0.1).bfloat16()
                Y = torch.empty(M, n, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
                old = timeit(lambda: ops.linear_fwd_b16(A32, W16, None, Y)) if (n % 4 == 0) else float("nan")
                print(f"{M:>7} {n:>5} {k:>5} {'bf16' if c16 else 'fp32'} {us:8.1f} {tf:8.1f} {gb:8.1f}  | {old:8.1f}")


def main_tn():
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
    H = 8
    lstm_in, lstm_hid = (H, 4, (0, 1, 2, 3)), (H, 4, (0, 1, 2, 3))
    gru_in, gru_hid = (H, 3, (0, 1, 2, -1)), (H, 3, (0, 1, -1, 2))
    for (m, T) in ([] if os.environ.get("TAD_MM16_DBG", "0") != "0" else [(24, 8), (72 * 3, 24), (1000, 0), (4096, 8)]):
        tn_case(m, 64, [(0, 64, 40, 0, None, True)], T=T)
        tn_case(m, 64, [(0, 64, 16, 0, lstm_in, True), (0, 32, 8, -1, lstm_hid, False), (32, 32, 8, 1, lstm_hid, False)], T=T or 8) if (T and m % T == 0) else None
        tn_case(m, 64, [(0, 64, 24, 0, gru_in, True), (0, 32, 8, -1, gru_hid, True), (32, 32, 8, 1, gru_hid, True)], T=T or 8) if (T and m % T == 0) else None
        tn_case(m, 200, [(0, 200, 136, 0, None, False), (8, 136, 200, 0, None, True)], T=0)
    print("mm16_tn correctness ok")
    Hh = 200
    li, lh = (Hh, 4, (0, 1, 2, 3)), (Hh, 4, (0, 1, 2, 3))
    cases = {
        "lstm L0 fused (dg 1600 | x 200, h 200+200)": (1600, [(0, 1600, 200, 0, li, True), (0, 800, 200, -1, lh, False), (800, 800, 200, 1, lh, False)], 24),
        "lstm L1 fused (dg 1600 | x 400, h 200+200)": (1600, [(0, 1600, 400, 0, li, True), (0, 800, 200, -1, lh, False), (800, 800, 200, 1, lh, False)], 24),
        "gru L0 fused (dg 1600 | x 128, h 200+200)": (1600, [(0, 1600, 128, 0, (Hh, 3, (0, 1, 2, -1)), True), (0, 800, 200, -1, (Hh, 3, (0, 1, -1, 2)), True), (800, 800, 200, 1, (Hh, 3, (0, 1, -1, 2)), True)], 24),
        "proj (dpz 200 x cat 800)": (200, [(0, 200, 800, 0, None, True)], 0),
        "mlp_nums (432 x 432)": (432, [(0, 432, 432, 0, None, True)], 0),
        "mlp_output.0 (128 x 560)": (128, [(0, 128, 560, 0, None, True)], 0),
        "cond_output (200 x 400)": (200, [(0, 200, 400, 0, None, True)], 0),
    }
    print(f"{'case':>46} {'us':>8} {'TFLOP/s':>8} {'GB/s':>8}")
    for name, (Na, spec, T) in cases.items():
        us, tf, gb, _ = tn_case(M, Na, spec, T=T, check=False)
        print(f"{name:>46} {us:8.1f} {tf:8.1f} {gb:8.1f}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("nt", "all"):
        main()
    if what in ("tn", "all"):
        main_tn()

