"""Per-launch device timing (ops.LaunchTrace) of one denoiser training step and one VAE training step at a workload's
shape.   usage: python scripts/profile_workload.py [metro_T48_H512|airq_T48_H512|etth1_T96_H200|stock_T24_H200|energy_T24_H200] [den|vae|both]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import bench  # noqa: E402
from profile_step import table  # noqa: E402
from timeautodiff_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "metro_T48_H512"
what = sys.argv[2] if len(sys.argv) > 2 else "both"
wl = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cx = bench.Ctx(dev, 1, 0)
if what in ("den", "both"):
    out, roof, kernels, keep = bench.bench_denoiser(cx, wl["schema"], wl["T"], wl["H"], wl["big_B"], "bf16", 3, 1, prime=3, trace=False)
    print(name, "denoiser", out)
    tr = keep[1]
    with ops.LaunchTrace() as t:
        for _ in range(2):
            tr.step()
    table(t.summary(), 2, f"denoiser train step {name}")
    by = t.summary_by_kernel()
    print("---- by kernel")
    for k, calls in sorted(by.items(), key=lambda kv: -sum(c[1] for c in kv[1].values()))[:14]:
        ms = sum(c[1] for c in calls.values()) / 2
        n = sum(c[0] for c in calls.values()) / 2
        print(f"{ms:8.3f} ms  x{n:5.1f}  {k}")
    del keep, tr
    torch.cuda.empty_cache()
if what in ("vae", "both"):
    from timeautodiff_amd.VAE import DeapStack, VAETrainer
    sch = wl["schema"]
    F_ = sch["n_bins"] + sch["n_cats"] + sch["n_nums"]
    B_ = wl["big_B"]
    _, _, resp = bench.synth(4 * B_ + 256, 77, dev, sch, wl["T"])
    mask = torch.ones_like(resp)
    dd = dict(response_train=resp[:4 * B_], target_mask_train=mask[:4 * B_], cond_train=None, response_val=resp[4 * B_:],
              target_mask_val=mask[4 * B_:], cond_val=None)
    torch.manual_seed(1)
    ae = DeapStack(1, sch["n_bins"], sch["n_cats"], sch["n_nums"], sch["cards"], F_, 512, 2, 128, 8, 6)
    vt = VAETrainer(ae, dd, batch_size=B_, lr=2e-4, weight_decay=1e-6, dtype="bf16", dev=dev)
    import random
    random.seed(1)
    for _ in range(3):
        vt.train_step(0.1)
    with ops.LaunchTrace() as t:
        for _ in range(2):
            vt.train_step(0.1)
    table(t.summary(), 2, f"VAE train step {name} hidden 512")
    by = t.summary_by_kernel()
    print("---- by kernel")
    for k, calls in sorted(by.items(), key=lambda kv: -sum(c[1] for c in kv[1].values()))[:14]:
        ms = sum(c[1] for c in calls.values()) / 2
        n = sum(c[0] for c in calls.values()) / 2
        print(f"{ms:8.3f} ms  x{n:5.1f}  {k}")
