#!/usr/bin/env python3
"""Wall time of a denoiser training step at small batches: kernel by kernel against one captured HIP graph per step,
with the draws of the reference's generators and with the counter-based device draws.
usage: python scripts/graph_vs_eager.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402

dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [100, 256, 512]:
    latent, time_info, cond = bench.synth(max(4 * B, 2048), 1, dev)
    for draws in ("reference", "device"):
        res = {}
        for graph in (False, True):
            torch.manual_seed(1)
            model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
            tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype="bf16", draws=draws, dev=dev, graph=graph)
            for _ in range(20):
                tr.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                tr.step()
            torch.cuda.synchronize()
            res[graph] = (time.perf_counter() - t0) / 300 * 1e3
        print(f"B={B:4d} draws={draws:9s}: kernel by kernel {res[False]:.3f} ms, graph {res[True]:.3f} ms", flush=True)
