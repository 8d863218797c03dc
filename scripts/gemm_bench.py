#!/usr/bin/env python3
"""Device timing of the three linear-family entry points at the shapes of the bench workload.
usage: gemm_bench.py [bf16|fp32] [M]   (TAD_HIP_LIB selects an alternative build of the library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

SHAPES = [(1600, 400), (1600, 200), (1200, 400), (432, 432), (200, 800), (200, 200), (128, 560), (800, 200)]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    prec = ops.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else ops.F32
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
    dev = torch.device("cuda:0")
    print(f"lib={os.environ.get('TAD_HIP_LIB', 'default')} prec={prec} M={M}")
    print(f"{'N':>6} {'K':>6} | {'fwd us':>8} {'TF/s':>6} {'GB/s':>6} | {'bwd_d us':>8} {'TF/s':>6} | {'bwd_w us':>8} {'TF/s':>6} | {'bwd_w+T':>8}")
    shapes = SHAPES
    only = os.environ.get("GEMM_ONLY")          # e.g. "1600x400:fwd" -> one op at one shape (for counter runs)
    if only:
        shp, op = only.split(":")
        N, K = (int(v) for v in shp.split("x"))
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.1
        y = torch.randn(M, N, device=dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.zeros(N, K, device=dev)
        fn = {"fwd": lambda: ops.linear_fwd(prec, x, w, None, y), "bwd_d": lambda: ops.linear_bwd_data(prec, y, w, dx),
              "bwd_w": lambda: ops.linear_bwd_weight(prec, y, x, dw)}[op]
        print(only, f"{timeit(fn, 5):.1f} us")
        from timeautodiff_amd import _lib
        lib = _lib.load()
        if hasattr(lib, "tad_dbg_stamps"):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            lib.tad_dbg_stamps(None, 1)
            fn()
            torch.cuda.synchronize()
            lib.tad_dbg_stamps(buf, 0)
            n = max(buf[7], 1)
            names = ["prologue", "load issue", "mfma", "wait+lds store", "barrier", "epilogue", "total"]
            print("  per-workgroup cycles (wave 0): " + ", ".join(f"{nm} {buf[i] / n:.0f}" for i, nm in enumerate(names)) + f"  (n={buf[7]})")
        return
    if prec == ops.BF16 and os.environ.get("GEMM_B16", "1") == "1":
        print("bf16-stored operands (recurrent layers' input side): fwd X fp32, W bf16 -> Y bf16; bwd_d dY bf16, W bf16; bwd_w dY bf16, X fp32")
        for N, K in [(1600, 400), (1600, 200), (1600, 128), (800, 200)]:
            x = torch.randn(M, K, device=dev)
            w16 = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
            b = torch.randn(N, device=dev)
            y16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            dx = torch.empty(M, K, device=dev)
            dw = torch.zeros(N, K, device=dev)
            db = torch.zeros(N, device=dev)
            fl = 2.0 * M * N * K
            t_f = timeit(lambda: ops.linear_fwd_b16(x, w16, b, y16))
            t_d = timeit(lambda: ops.linear_bwd_data_b16(y16, w16, dx))
            t_w = timeit(lambda: ops.linear_bwd_weight(prec, y16, x, dw, db=db, gate_map=(N // 8, 4, (0, 1, 2, 3))))
            t_ws = timeit(lambda: ops.linear_bwd_weight(prec, y16, x, dw, T=24, tshift=-1)) if M % 24 == 0 else 0.0
            by = 4.0 * M * K + 2.0 * N * K + 2.0 * M * N
            print(f"{N:6d} {K:6d} | {t_f:8.1f} {fl / t_f / 1e6:6.1f} {by / t_f / 1e3:6.0f} | {t_d:8.1f} {fl / t_d / 1e6:6.1f} | "
                  f"{t_w:8.1f} {fl / t_w / 1e6:6.1f} | {t_ws:8.1f}")
    for N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        dx = torch.empty(M, K, device=dev)
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: ops.linear_fwd(prec, x, w, b, y))
        t_d = timeit(lambda: ops.linear_bwd_data(prec, y, w, dx))
        t_w = timeit(lambda: ops.linear_bwd_weight(prec, y, x, dw, db=db))
        t_ws = timeit(lambda: ops.linear_bwd_weight(prec, y, x, dw, T=24, tshift=-1)) if M % 24 == 0 else 0.0
        by = 4.0 * (M * K + N * K + M * N)
        print(f"{N:6d} {K:6d} | {t_f:8.1f} {fl / t_f / 1e6:6.1f} {by / t_f / 1e3:6.0f} | {t_d:8.1f} {fl / t_d / 1e6:6.1f} | "
              f"{t_w:8.1f} {fl / t_w / 1e6:6.1f} | {t_ws:8.1f}")


if __name__ == "__main__":
    main()
