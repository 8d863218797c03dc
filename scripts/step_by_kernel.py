#!/usr/bin/env python3
"""Device time of one denoiser training step per kernel at a given batch (default: the reference's 100 windows).
usage: python scripts/step_by_kernel.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import ops  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.manual_seed(1)
latent, time_info, cond = bench.synth(max(4 * B, 2048), 1, dev)
model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype="bf16", draws="device", dev=dev, graph=False)
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
with ops.LaunchTrace() as t:
    for _ in range(5):
        tr.step()
torch.cuda.synchronize()
agg = {}
for key, kern, e0, e1, cost in t.records:
    a = agg.setdefault((key[0], kern), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values()) / 5
print(f"B={B}: {sum(v[0] for v in agg.values()) // 5} launches, {tot:.0f} us of kernel time per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "14"))]:
    print(f"  {v[0] // 5:3d} x {v[1] / v[0]:7.1f} us = {v[1] / 5:7.1f} us  {k[0]} {k[1]}")
