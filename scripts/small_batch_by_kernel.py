#!/usr/bin/env python3
"""Device time per launch of one denoiser training step at a small batch (the reference's own B = 100), beside the wall
time per step kernel by kernel and as a graph.   usage: python scripts/small_batch_by_kernel.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import bench  # noqa: E402
from profile_step import table  # noqa: E402
from timeautodiff_amd import ops  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
latent, time_info, cond = bench.synth(2048, 1, dev)
for graph in (False, True):
    torch.manual_seed(1)
    model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
    tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype="bf16", draws="reference", dev=dev, graph=graph)
    for _ in range(20):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        tr.step()
    torch.cuda.synchronize()
    print(f"B={B} graph={graph}: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per step", flush=True)
    if not graph:
        with ops.LaunchTrace() as t:
            for _ in range(4):
                tr.step()
        s = t.summary()
        table(s, 4, f"denoiser train step B={B}")
        print("launches per step:", sum(v[0] for v in s.values()) / 4)
