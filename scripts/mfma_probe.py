#!/usr/bin/env python3
"""The workload of one counter pass: steady-state denoiser training steps, or sampler calls, at the bench shape and
nothing else (run under rocprofv3 --pmc by scripts/pmc_mfma.sh).
usage: python scripts/mfma_probe.py train|sample"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import DIFF  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402


def main():
    what = sys.argv[1]
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    latent, time_info, cond = bench.synth(bench.N_DATA, 1234, dev)
    model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
    if what == "train":
        tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=bench.B_LOCAL, dtype="bf16", draws="device", dev=dev)
        for _ in range(6):
            tr.step()
    else:
        model.set_compute_dtype("bf16")
        Bs = bench.B_SAMPLE
        tgrid = torch.linspace(0, 1, bench.T, device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
        for _ in range(2):
            DIFF.sample(tgrid, Bs, bench.T, bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source="device")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
