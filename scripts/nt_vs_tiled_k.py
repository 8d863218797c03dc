"""tad_mm16_nt against the square-tiled kernel (tad_linear_fwd_b16x) at 98 304 rows for results of few columns and long
reductions (the denoiser's `proj`: N = H, K = 4 H), bf16 operands and result, ReLU epilogue.
usage: python scripts/nt_vs_tiled_k.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402
from timeautodiff_amd._lib import ACT_RELU  # noqa: E402

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


shapes = [(N, K) for K in (256, 400, 512, 800, 1024) for N in (128, 200, 256, 400, 512, 800, 1600)]
if len(sys.argv) > 1 and sys.argv[1] == "step":
    shapes = [(200, 800), (200, 408), (200, 400), (200, 200), (128, 560), (400, 200), (512, 512), (432, 432)]
for M, (N, K) in [(m, nk) for m in (98304, 12288) for nk in shapes]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = ops.w16_image(N, K, dev)
    W[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty_like(C)
    t1 = timed(lambda: ops.mm16_nt(A, W, bias, C, act=ACT_RELU))
    t2 = timed(lambda: ops.linear_fwd_b16x(A, W[:N, :K], bias, C2, act=ACT_RELU))
    err = (C.float() - C2.float()).abs().max().item()
    print(f"M={M} N={N} K={K}: {'TILED' if t2 < 0.97 * t1 else ('nt   ' if t1 < 0.97 * t2 else 'same ')} mm16_nt {t1:7.1f} us  tiled {t2:7.1f} us  max diff {err:.3g}", flush=True)
