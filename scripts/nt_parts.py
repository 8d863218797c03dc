"""tad_mm16_nt at the shapes of the training step with parts of the kernel switched off (TAD_MM16_DBG: 1 = no epilogue,
2 = no MFMAs, 4 = no ring loads, 8 = no stores; read once per process) and with the split count forced (TAD_MM16_NSPLIT).
usage: [TAD_MM16_DBG=..] python scripts/nt_parts.py"""
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


M = 98304
out = []
for N, K, cdt in ((432, 432, torch.bfloat16), (1600, 400, torch.bfloat16), (400, 200, torch.bfloat16), (128, 128, torch.bfloat16),
                  (2048, 512, torch.bfloat16)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = ops.w16_image(N, K, dev)
    W[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=cdt)
    t1 = timed(lambda: ops.mm16_nt(A, W, bias, C))
    out.append(f"N={N} K={K}: {t1:7.1f} us ({2.0 * M * N * K / t1 / 1e6:6.1f} TF/s, {(2.0 * M * K + C.element_size() * M * N) / t1 / 1e3:6.0f} GB/s)")
print(f"DBG={os.environ.get('TAD_MM16_DBG', '0')} NSPLIT={os.environ.get('TAD_MM16_NSPLIT', '-')}: " + " | ".join(out), flush=True)
