"""LayerNorm + fc (forward and backward launches) at the row shapes of the training steps: time, bytes per second and a
check of the forward against torch.  usage: python scripts/lnfc_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


K = 6
for rows, D in ((98304, 400), (196608, 1024), (2400, 400), (24576, 400)):
    x = torch.randn(rows, D, device=dev)
    gamma, beta = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
    fw, fb = torch.randn(K, D, device=dev) * 0.05, torch.randn(K, device=dev)
    mr, pred = torch.empty(rows, 2, device=dev), torch.empty(rows, K, device=dev)
    tf = timed(lambda: ops.ln_fc_fwd(0, x, gamma, beta, fw, fb, mr, pred))
    ref = torch.nn.functional.layer_norm(x[:512], (D,), gamma, beta) @ fw.t() + fb
    err = (pred[:512] - ref).abs().max().item()
    dpred, dx = torch.randn(rows, K, device=dev), torch.empty(rows, D, device=dev)
    dg, db, dfw, dfb = (torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(K, D, device=dev), torch.zeros(K, device=dev))
    tb = timed(lambda: ops.ln_fc_bwd(0, dpred, x, gamma, beta, mr, fw, dx, dg, db, dfw, dfb))
    print(f"{rows} x {D}: fwd {tf:6.1f} us ({4.0 * rows * D / tf / 1e3:5.0f} GB/s, err {err:.1e})  bwd pair {tb:6.1f} us ({12.0 * rows * D / tb / 1e3:5.0f} GB/s)", flush=True)
