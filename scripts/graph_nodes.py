"""What the HIP graphs of the trainers and the sampler launch when replayed (a memset node replayed from a graph was seen to
write other values than the captured one: the captured launch sequences must hold kernel nodes of this library only).
usage: cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python scripts/graph_nodes.py, then look for
__amd_rocclr_* rows in OUT/*_kernel_stats.csv with 40 x (graphs) or more calls"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from timeautodiff_amd import DIFF, VAE  # noqa: E402

dev = torch.device("cuda:0")
orig = torch.cuda.CUDAGraph
made = []


class Dbg(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k)
        made.append(g)
        return g


torch.cuda.CUDAGraph = Dbg
latent, time_info, cond = bench.synth(2048, 1, dev)
torch.manual_seed(1)
model = DIFF.BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
tr = DIFF.DiffusionTrainer(model, latent, time_info, cond, batch_size=100, dtype="bf16", draws="reference", dev=dev, graph=True)
for _ in range(3):
    tr.step()
tg = torch.linspace(0, 1, bench.T, device=dev).view(1, -1, 1).repeat(256, 1, 1)
DIFF.sample(tg, 256, bench.T, bench.K, 100, model, time_info[:256], cond[:256], None, noise_source="device", graph=True)
ones = torch.ones_like(cond)
dd = dict(response_train=cond[:1500], target_mask_train=ones[:1500], cond_train=None, response_val=cond[1500:1700], target_mask_val=ones[1500:1700], cond_val=None)
ae = VAE.DeapStack(1, 0, 1, 27, [8], bench.F, 512, 2, bench.E, 8, bench.K)
vt = VAE.VAETrainer(ae, dd, batch_size=100, lr=2e-4, weight_decay=1e-6, dtype="bf16", dev=dev)
for _ in range(3):
    vt.train_step(0.1)
    vt.eval_loss(0.1)
torch.cuda.synchronize()
print(len(made), "graphs captured")
# 40 replays of each: under `rocprofv3 --kernel-trace --stats` a memset or copy node shows as 40 more launches of the
# runtime's fill / copy kernels (__amd_rocclr_fillBuffer*, __amd_rocclr_copyBuffer*)
only = os.environ.get("GRAPH")
for i, g in enumerate(made):
    if only is not None and int(only) != i:
        continue
    for _ in range(40):
        g.replay()
torch.cuda.synchronize()
