"""The three hot loops (denoiser training, reverse-diffusion sampling, VAE
training) restated in numpy with every random draw passed in explicitly.

Oracle code: test infrastructure only.
"""
import numpy as np

from . import denoiser, ops, vae
from .optim import Adam


# train_diffusion's iteration -- DIFF.py:305-326 (EMA is unobservable: the
# function returns ``model``, DIFF.py:330)
def train_diffusion_steps(sd, latent, time_info, cond, schema, draws, alphas, lr=1e-3,
                          diffusion_steps=100, num_layers=2):
    """draws: iterable of dicts with 'idx' [B], 't' [B,T,1], 'i' [B], 'noise' [B,T,K].
    Updates ``sd`` in place; returns the list of losses."""
    opt = Adam(sd, lr=lr)
    losses = []
    for d in draws:
        idx = np.asarray(d["idx"])
        loss, _, grads = denoiser.loss_and_grads(sd, latent[idx], d["t"], np.asarray(d["i"]), d["noise"],
                                                 time_info[idx], cond[idx], schema, alphas,
                                                 diffusion_steps=diffusion_steps, num_layers=num_layers)
        opt.step(grads)
        losses.append(loss)
    return losses


# sample -- DIFF.py:333-352
def sample(sd, t, time_info, cond, schema, x_T, zs, betas, alphas, diffusion_steps=100, num_layers=2,
           keep=()):
    """x_T [B,T,K]; zs[s] is the noise drawn at reverse step s (s = steps-1 ... 0).
    Returns (x, {step: x after that step} for steps in keep)."""
    x = x_T
    B, T, _ = x.shape
    kept = {}
    for s in range(diffusion_steps - 1, -1, -1):
        i = np.full((B, T, 1), s, dtype=x.dtype)
        pred, _ = denoiser.forward(sd, x, t, i, time_info, cond, schema, diffusion_steps, num_layers)
        x = ops.sampler_update(x, pred, zs[s], betas[s], alphas[s], last=(s == 0))
        if s in keep:
            kept[s] = x.copy()
    return x, kept


# train_autoencoder's iteration -- VAE.py:320-383 (selection by validation loss)
# and train_autoencoder_best_train's -- VAE.py:477-547 (selection by train loss)
def train_autoencoder_steps(sd, data, schema, draws, lr, weight_decay, min_beta, max_beta,
                            select="val", num_layers=2):
    """data: dict with response/cond/target_mask _train and _val arrays.
    draws: iterable of dicts with 'idx', 'eps_train' [B,T,L], 'eps_val' [Nv,T,L].
    Returns (history, best_sd, final_beta)."""
    dt = data["response_train"].dtype
    opt = Adam(sd, lr=lr, weight_decay=weight_decay)
    inp_tr = data["response_train"] + data["cond_train"]
    inp_va = data["response_val"] + data["cond_val"]
    beta = max_beta
    patience = 0
    best_val = float("inf")
    best_train = float("inf")
    best = {k: v.copy() for k, v in sd.items()}
    hist = {"train": [], "val": []}
    for d in draws:
        idx = np.asarray(d["idx"])
        tl, _, grads = vae.loss_and_grads(sd, inp_tr[idx], data["target_mask_train"][idx], data["cond_train"][idx],
                                          d["eps_train"], dt.type(beta), schema, num_layers=num_layers)
        opt.step(grads)
        if select == "train" and tl < best_train:
            best_train = tl
            best = {k: v.copy() for k, v in sd.items()}
        vl, _, _ = vae.loss_and_grads(sd, inp_va, data["target_mask_val"], data["cond_val"], d["eps_val"],
                                      dt.type(beta), schema, want_grads=False, num_layers=num_layers)
        if vl < best_val:
            best_val = vl
            if select == "val":
                best = {k: v.copy() for k, v in sd.items()}
            patience = 0
        else:
            patience += 1
            if patience == 10 and beta > min_beta:
                beta *= 0.7
                patience = 0
        hist["train"].append(tl)
        hist["val"].append(vl)
    return hist, best, beta
