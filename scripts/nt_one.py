"""One product of tad_mm16_nt at the bench's row count, a few launches: the program scripts/pmc_mm16.sh profiles.
usage: python scripts/nt_one.py N K"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1600, 400)
M = 98304
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16()
W = ops.w16_image(N, K, dev)
W[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(8):
    ops.mm16_nt(A, W, None, C)
torch.cuda.synchronize()
