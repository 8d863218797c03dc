"""Wall time of DIFF.sample (1024 sequences, energy shape, bf16, device noise) by number of concurrent chunks, launched
kernel by kernel and replayed as one captured graph.   usage: python scripts/sampler_chunks.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import DIFF  # noqa: E402

Bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(1)
latent, time_info, cond = bench.synth(2048, 1, dev)
model = DIFF.BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
model.set_compute_dtype("bf16")
tgrid = torch.linspace(0, 1, bench.T, device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
for graph in (False, True):
    for k in (1, 2, 3, 4):
        run = lambda: DIFF.sample(tgrid, Bs, bench.T, bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source="device", streams=k, graph=graph)  # noqa: E731
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        print(f"graph={graph} chunks={k}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call", flush=True)
