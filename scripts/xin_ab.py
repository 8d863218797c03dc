#!/usr/bin/env python3
"""A/B of the VAE step with the first encoder layer's input product inside the sequence kernel (TAD_RNN_XIN) and as a
product of its own: latents, every workspace buffer of the encoder stack and every gradient, compared tensor by tensor.
Usage: xin_ab.py [shape name]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_models as m  # noqa: E402
from timeautodiff_amd import engine  # noqa: E402

fu = m.fu
name = sys.argv[1] if len(sys.argv) > 1 else "metro_T48_H512"
dev = torch.device("cuda:0")
sd_, _, T, B, Hv, mk = m.BASELINE_SHAPES[name]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
sch = m.Schema(sd_["n_bins"], sd_["n_cats"], sd_["n_nums"], list(sd_["cards"]))
L, E, beta = 6, 128, 0.05
data = fu.make_data(sd_, N=B, T=T, K=L, seed=96, mask_kind=mk)
eps = np.random.default_rng(97).standard_normal((B, T, L)).astype(np.float32)
rows = B * T


def run(flag):
    engine.RNN_XIN = flag
    ae = m._make_vae(dict(H=Hv, E=E, L=L, seed=95), sch, dev, "bf16")
    eng = ae.engine(dev)
    inp = m.T_(data["response"] + data["cond"], dev)
    mask, cond = m.T_(data["mask"], dev), m.T_(data["cond"], dev)
    parts = torch.zeros(3 + max(sch.n_cats, 1), device=dev)
    eng.fp.zero_grad()
    zz, mu, lv = eng.encoder(inp, m.T_(eps, dev), train=True, kld_sum=parts[2 + max(sch.n_cats, 1):])
    eng.decoder(zz, mask, cond, target=inp * mask, scales=eng.loss_scales(rows), loss_parts=parts, train=True)
    eng.backward(beta, rows)
    torch.cuda.synchronize()
    out = {"z": zz.float().cpu().numpy(), "mu": mu.float().cpu().numpy()}
    for k, v in eng.ws.bufs.items() if hasattr(eng, "ws") else []:
        if v.dtype in (torch.float32, torch.bfloat16):
            out["ws:" + k] = v.float().cpu().numpy()
    for k in eng.fp.offsets:
        if eng.fp.offsets[k] < eng.fp.n_active:
            out["g:" + k] = eng.fp.g(k).cpu().numpy().copy()
    return out


a, b = run(False), run(True)
for k in a:
    if k in b and a[k].shape == b[k].shape:
        d = np.linalg.norm((a[k] - b[k]).ravel().astype(np.float64)) / max(np.linalg.norm(a[k].ravel().astype(np.float64)), 1e-30)
        if d > float(os.environ.get("AB_MIN", "1e-3")):
            print(f"{d:10.3e}  {k}  {a[k].shape}")
print("only one side:", sorted(set(a) ^ set(b)))
