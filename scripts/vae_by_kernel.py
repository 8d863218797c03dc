#!/usr/bin/env python3
"""Device time of one VAE training step (DeapStack, hidden 512 by default) per kernel at a given batch.
usage: python scripts/vae_by_kernel.py [B [hidden]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import ops  # noqa: E402
from timeautodiff_amd.VAE import DeapStack, VAETrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else bench.VAE_HIDDEN
dev = torch.device("cuda:0")
schema = bench.SCHEMA
NV = max(4 * B, 1024)
_, _, resp = bench.synth(NV + 256, 77, dev)
mask = torch.ones(NV + 256, bench.T, bench.F, device=dev)
dd = dict(response_train=resp[:NV], target_mask_train=mask[:NV], cond_train=None,
          response_val=resp[NV:], target_mask_val=mask[NV:], cond_val=None)
ae = DeapStack(1, schema["n_bins"], schema["n_cats"], schema["n_nums"], schema["cards"], bench.F, hidden, bench.LAYERS, bench.E, 8, bench.K)
vt = VAETrainer(ae, dd, batch_size=B, lr=2e-4, weight_decay=1e-6, dtype="bf16", dev=dev, graph=False)
import random  # noqa: E402
random.seed(99)
for _ in range(5):
    vt.train_step(0.1)
torch.cuda.synchronize()
with ops.LaunchTrace() as t:
    for _ in range(5):
        vt.train_step(0.1)
torch.cuda.synchronize()
agg = {}
for key, kern, e0, e1, cost in t.records:
    a = agg.setdefault((key[0], kern), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values()) / 5
print(f"B={B} hidden={hidden}: {sum(v[0] for v in agg.values()) // 5} launches, {tot:.0f} us of kernel time per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "30"))]:
    print(f"  {v[0] // 5:3d} x {v[1] / v[0]:7.1f} us = {v[1] / 5:7.1f} us  {k[0]} {k[1]}")
