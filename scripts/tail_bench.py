"""The folded tail of a reverse step (tad_sampler_tail_proj_cn_b16) alone, back to back, by row count: the constant and the
per-row share of a launch.   usage: python scripts/tail_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
Dm, K, H = 400, 6, 200
g = torch.Generator(device="cpu").manual_seed(1)
r = lambda *sh: torch.randn(*sh, generator=g).to(dev)  # noqa: E731
gamma, beta, fc_w, fc_b, wc, bc, vrow = r(Dm), r(Dm), r(K, Dm) * 0.1, r(K), r(H, K), r(H), r(H)
for rows in (256, 3072, 6144, 12288, 24576, 49152, 98304):
    h, pinv, x = r(rows, Dm), r(rows, H), r(rows, K)
    pa = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.sampler_tail_proj_cn(1, h, gamma, beta, fc_w, fc_b, x, 1234, 5, 0, 1.0, 0.1, 0.0, False, wc, bc, vrow, pinv, pa)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byt = rows * (Dm * 4 + H * 4 + H * 2 + K * 8)
    print(f"rows {rows:6d}: {us:7.1f} us per launch, {byt / us / 1e3:7.1f} GB/s", flush=True)
