"""The square-tiled product kernel on bf16-stored operands at the shapes of the H = 512 and H = 200 training steps:
forward (tad_linear_fwd_b16x) and input gradient (tad_linear_bwd_data_b16), with a check of the first 256 rows.
usage: python scripts/tiled_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for M, N, K in ((196608, 4096, 1024), (196608, 4096, 512), (98304, 2048, 512), (98304, 1600, 400), (98304, 200, 800), (98304, 512, 512)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: ops.linear_fwd_b16x(x, w, b, y))
    ref = (x[:256].float() @ w.float().t() + b)
    err = (y[:256].float() - ref).abs().max().item() / ref.abs().max().item()
    out.append(f"fwd {N}x{K}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s err {err:.1e}")
    del x, y
for M, N, K in ((196608, 4096, 1024), (98304, 2048, 512), (98304, 1600, 400)):
    dy = torch.randn(M, N, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    dx = torch.empty(M, K, device=dev)
    t = timed(lambda: ops.linear_bwd_data_b16(dy, w, dx))
    ref = dy[:256].float() @ w.float()
    err = (dx[:256] - ref).abs().max().item() / ref.abs().max().item()
    out.append(f"dx {N}->{K}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s err {err:.1e}")
    del dy, dx
print(" | ".join(out), flush=True)
