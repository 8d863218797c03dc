#!/usr/bin/env python3
"""Timing of tad_mm16_nt with epilogues that read a second operand (residual, ReLU mask, SiLU pre-activations) beside
the plain product of the same shape.  usage: python scripts/nt_epilogue_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K, kind) in [(98304, 432, 432, "dsilu_f32"), (98304, 128, 128, "relu_b16"), (98304, 200, 200, "relu_f32"),
                        (24576, 200, 200, "res_f32_relu"), (98304, 200, 400, "relu_b16")]:
    A = torch.randn(M, K, device=dev).bfloat16()
    Wp = ops.w16_image(N, K, dev)
    Wp[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    U32, U16 = torch.randn(M, N, device=dev), torch.randn(M, N, device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    plain = timeit(lambda: ops.mm16_nt(A, Wp, None, C))
    if kind == "dsilu_f32":
        t = timeit(lambda: ops.mm16_nt(A, Wp, None, C, act=3, relu_of=U32))
    elif kind == "relu_b16":
        t = timeit(lambda: ops.mm16_nt(A, Wp, None, C, relu_of=U16))
    elif kind == "relu_f32":
        t = timeit(lambda: ops.mm16_nt(A, Wp, None, C, relu_of=U32))
    else:
        t = timeit(lambda: ops.mm16_nt(A, Wp, b, C, act=1, residual=U32))
    print(f"M={M} N={N} K={K} {kind:14s} plain {plain:7.1f} us   with epilogue operand {t:7.1f} us")
