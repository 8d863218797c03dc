import os, sys
sys.path.insert(0, '.')
import torch, bench
from timeautodiff_amd import ops
from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer
B = 4096
dev = torch.device("cuda:0")
torch.manual_seed(1)
latent, time_info, cond = bench.synth(4 * B, 1, dev)
model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype="bf16", draws="device", dev=dev, graph=False)
for _ in range(10): tr.step()
torch.cuda.synchronize()
with ops.LaunchTrace() as t:
    for _ in range(5): tr.step()
torch.cuda.synchronize()
agg = {}
for key, kern, e0, e1, cost in t.records:
    if key[0] not in ("tad_mm16_nt", "tad_linear_bwd_data_b16"): continue
    a = agg.setdefault(key[:4] + (kern,), [0, 0.0, cost]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = v[1] / v[0]
    fl, by = v[2]
    print(f"{v[0]//5} x {us:7.1f} us  M={k[1]} N={k[2]} K={k[3]}  {fl/us/1e6:7.1f} TFLOP/s {by/us/1e3:7.1f} GB/s  {k[4]}")
