"""tad_mm16_nt against the square-tiled kernel (tad_linear_fwd_b16x) on bf16 operands and results, at row counts of the
sampler (chunks of 256..1024 sequences x 24 steps) and of training.   usage: python scripts/nt_vs_tiled.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (6144, 12288, 24576, 98304):
    for N, K in ((1600, 200), (1600, 400), (1600, 128)):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = ops.w16_image(N, K, dev)
        W[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
        bias = torch.randn(N, device=dev)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        C2 = torch.empty_like(C)
        t1 = timed(lambda: ops.mm16_nt(A, W, bias, C))
        t2 = timed(lambda: ops.linear_fwd_b16x(A, W[:N, :K], bias, C2))
        err = (C.float() - C2.float()).abs().max().item()
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N} K={K}: mm16_nt {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF/s)  tiled {t2:7.1f} us ({fl / t2 / 1e6:6.1f} TF/s)  max diff {err:.3g}", flush=True)
