#!/usr/bin/env python3
"""Per-call device timing of one VAE training step at the bench workload (events around every C-ABI call)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from profile_step import table  # noqa: E402
from timeautodiff_amd import ops  # noqa: E402
from timeautodiff_amd.VAE import DeapStack, VAETrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
_, _, cond = bench.synth(bench.N_DATA, 1234, dev)
NV = 4 * bench.B_LOCAL
ones = torch.ones(NV, bench.T, bench.F, device=dev)
dd = dict(response_train=cond[:NV], target_mask_train=ones, cond_train=None,
          response_val=cond[NV:NV + 256], target_mask_val=ones[:256], cond_val=None)
HID = int(sys.argv[1]) if len(sys.argv) > 1 else bench.VAE_HIDDEN
ae = DeapStack(1, 0, 1, 27, [8], bench.F, HID, 2, bench.E, 8, bench.K)
vt = VAETrainer(ae, dd, batch_size=bench.B_LOCAL, lr=2e-4, weight_decay=1e-6, dtype="bf16", dev=dev)
random.seed(99)
for _ in range(3):
    vt.train_step(0.1)
torch.cuda.synchronize()
with ops.LaunchTrace() as t:
    for _ in range(3):
        vt.train_step(0.1)
table(t.summary(), 3, "VAE train step B=4096 bf16")
