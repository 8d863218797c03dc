#!/usr/bin/env python3
"""Host-side enqueue time of one denoiser training step versus its device time (bench workload)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
latent, time_info, cond = bench.synth(bench.N_DATA, 1234, dev)
model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=bench.B_LOCAL, dtype="bf16", draws="device", dev=dev)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
import gc
if os.environ.get("GC_FREEZE", "0") == "1":
    gc.collect()
    gc.freeze()
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"10 steps: host enqueue {1e2 * (t1 - t0):.2f} ms/step, wall {1e2 * (t2 - t0):.2f} ms/step, gc counts {gc.get_count()}")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
