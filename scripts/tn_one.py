"""The weight-gradient launch of one recurrent layer (tad_mm16_tn, LSTM layer 0 at the bench shape by default), a few
launches: the program scripts/pmc_tn.sh profiles.   usage: python scripts/tn_one.py [lstm0|lstm1|gru0] [M]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "lstm0"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 98304
H, T = 200, 24
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
li, lh = (H, 4, (0, 1, 2, 3)), (H, 4, (0, 1, 2, 3))
gi, gh = (H, 3, (0, 1, 2, -1)), (H, 3, (0, 1, -1, 2))
spec = {"lstm0": [(0, 1600, 200, 0, li), (0, 800, 200, -1, lh), (800, 800, 200, 1, lh)],
        "lstm1": [(0, 1600, 400, 0, li), (0, 800, 200, -1, lh), (800, 800, 200, 1, lh)],
        "gru0": [(0, 1600, 128, 0, gi), (0, 800, 200, -1, gh), (800, 800, 200, 1, gh)]}[which]
A = torch.randn(M, 1600, device=dev, generator=g).bfloat16()
jobs = []
for a0, na, nb, shift, gm in spec:
    B = torch.randn(M, nb, device=dev, generator=g).bfloat16()
    jobs.append(dict(b=B, c=torch.zeros(na // 4 * gm[1], nb, device=dev), a0=a0, na=na, shift=shift, gate_map=gm))
for _ in range(6):
    ops.mm16_tn(A, jobs, T=T)
torch.cuda.synchronize()
