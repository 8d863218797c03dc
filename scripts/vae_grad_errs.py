#!/usr/bin/env python3
"""The VAE parity test at a baseline shape with its numbers printed: relative L2 error of the latents and of the largest
per-tensor gradient errors against the bf16-emulating oracle, with the first encoder layer's input product inside the
sequence kernel (TAD_RNN_XIN) and as a product of its own.  Usage: vae_grad_errs.py [shape name] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_models as m  # noqa: E402
from oracle import ops as oops, rnn as ornn, vae as ovae  # noqa: E402
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
l2 = lambda a, b: float(np.linalg.norm((np.asarray(a, np.float64) - b).ravel()) / max(np.linalg.norm(np.asarray(b, np.float64).ravel()), 1e-30))


def device(flag):
    engine.RNN_XIN = flag
    ae = m._make_vae(dict(H=Hv, E=E, L=L, seed=95), sch, dev, "bf16")
    sd = {k: v.detach().cpu().numpy().copy() for k, v in ae.state_dict().items()}
    eng = ae.engine(dev)
    inp = m.T_(data["response"] + data["cond"], dev)
    mask, cond = m.T_(data["mask"], dev), m.T_(data["cond"], dev)
    parts = torch.zeros(3 + max(sch.n_cats, 1), device=dev)
    eng.fp.zero_grad()
    zz, mu, lv = eng.encoder(inp, m.T_(eps, dev), train=True, kld_sum=parts[2 + max(sch.n_cats, 1):])
    eng.decoder(zz, mask, cond, target=inp * mask, scales=eng.loss_scales(rows), loss_parts=parts, train=True)
    eng.backward(beta, rows)
    torch.cuda.synchronize()
    grads = {k: eng.fp.g(k).cpu().numpy().copy() for k in eng.fp.offsets if eng.fp.offsets[k] < eng.fp.n_active}
    return sd, zz.cpu().numpy(), mu.cpu().numpy(), grads


def oracle(sd, on_chip):
    # on_chip False: the first layer's pre-activations go through bf16 storage whatever the shape
    saved = ornn.rnn_xproj_on_chip
    if not on_chip:
        import contextlib
        ornn.rnn_xproj_on_chip = contextlib.nullcontext
    try:
        with oops.bf16_operands(storage=True):
            return ovae.loss_and_grads(sd, data["response"] + data["cond"], data["mask"], data["cond"], eps, np.float32(beta), sch)
    finally:
        ornn.rnn_xproj_on_chip = saved


for flag in (False, True):
    sd, z, mu, grads = device(flag)
    for on_chip in (False, True):
        ototal, ores, ograds = oracle(sd, on_chip)
        errs = {k: l2(grads[k], v.reshape(grads[k].shape)) for k, v in ograds.items()}
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"device xin {int(flag)} vs oracle on-chip {int(on_chip)}: z {l2(z, ores['z']):.2e} mu {l2(mu, ores['mu']):.2e}  grads " +
              ", ".join(f"{k} {v:.2e}" for k, v in top))
