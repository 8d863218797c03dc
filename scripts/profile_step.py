#!/usr/bin/env python3
"""Per-call device timing of one denoiser training step and one sampler run at the bench
workload (events around every C-ABI call).  Prints a table sorted by total time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import DIFF, ops  # noqa: E402
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer  # noqa: E402


def table(summ, reps, title):
    print(f"==== {title}: ms per iteration, sorted")
    rows = []
    for key, (cnt, ms, fl, by, _issued) in summ.items():
        avg = ms / cnt
        rows.append((ms / reps, cnt / reps, avg, fl / avg / 1e9 if avg > 0 else 0, by / avg / 1e6 if avg > 0 else 0, key))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    for r in rows[:60]:
        print(f"{r[0]:8.3f} ms  x{r[1]:5.1f}  avg {r[2]*1e3:8.1f} us  {r[3]:8.1f} TF/s  {r[4]:8.1f} GB/s  {r[5]}")
    print(f"total {tot:.3f} ms")


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.B_LOCAL
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    latent, time_info, cond = bench.synth(bench.N_DATA, 1234, dev)
    model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
    tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype=dtype, draws="device", dev=dev)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    with ops.LaunchTrace() as t:
        for _ in range(3):
            tr.step()
    table(t.summary(), 3, f"train step B={B} {dtype}")
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    print(f"wall per step: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")

    Bs = bench.B_SAMPLE
    tgrid = torch.linspace(0, 1, bench.T, device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
    DIFF.sample(tgrid, Bs, bench.T, bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source="device")
    with ops.LaunchTrace() as t:
        DIFF.sample(tgrid, Bs, bench.T, bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source="device")
    table(t.summary(), 1, f"sample B={Bs} {dtype}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    DIFF.sample(tgrid, Bs, bench.T, bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source="device")
    torch.cuda.synchronize()
    print(f"wall per sample call: {(time.perf_counter() - t0) * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
