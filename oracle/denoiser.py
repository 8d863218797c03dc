"""BiRNN_score (the diffusion denoiser), its loss and their gradients in numpy.

Oracle code: test infrastructure only.  Parameters are a dict keyed by the
reference's ``state_dict`` names.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import ops, rnn


@dataclass
class Schema:
    """Column typing of the tabular features (process_edited.py:245): columns are
    ordered binaries, categoricals, numericals."""
    n_bins: int
    n_cats: int
    n_nums: int
    cards: List[int] = field(default_factory=list)

    @property
    def n_disc(self):
        return self.n_bins + self.n_cats

    @property
    def table_sizes(self):
        return [2] * self.n_bins + list(self.cards)


# ----------------------------------------------------------------------------
# Embedding_data -- DIFF.py:35-85 and VAE.py:58-108 (same forward; the only
# difference is the width of mlp_output's last layer, which the weights carry)
def embedding_fwd(sd, prefix, x, schema: Schema):
    nd, nn_ = schema.n_disc, schema.n_nums
    parts = []
    cache = {}
    if nd:
        idx = x[..., :nd].astype(np.int64)             # .long(): truncation
        cache["idx"] = idx
        parts += [sd[f"{prefix}embeddings_list.{j}.weight"][idx[..., j]] for j in range(nd)]
    if nn_:
        feats = ops.sincos_features(x[..., nd:nd + nn_], 8)
        z1 = ops.linear_fwd(feats, sd[prefix + "mlp_nums.0.weight"], sd[prefix + "mlp_nums.0.bias"])
        a1 = ops.silu(z1)
        y1 = ops.linear_fwd(a1, sd[prefix + "mlp_nums.2.weight"], sd[prefix + "mlp_nums.2.bias"])
        cache.update(feats=feats, z1=z1, a1=a1)
        parts.append(y1)
    cat = np.concatenate(parts, -1)
    z2 = ops.linear_fwd(cat, sd[prefix + "mlp_output.0.weight"], sd[prefix + "mlp_output.0.bias"])
    a2 = np.maximum(z2, 0)
    out = ops.linear_fwd(a2, sd[prefix + "mlp_output.2.weight"], sd[prefix + "mlp_output.2.bias"])
    cache.update(cat=cat, a2=a2)
    return out, cache


def embedding_bwd(sd, prefix, cache, dout, schema: Schema, grads):
    nd, nn_ = schema.n_disc, schema.n_nums
    da2, grads[prefix + "mlp_output.2.weight"], grads[prefix + "mlp_output.2.bias"] = \
        ops.linear_bwd(cache["a2"], sd[prefix + "mlp_output.2.weight"], dout)
    dz2 = da2 * (cache["a2"] > 0)
    dcat, grads[prefix + "mlp_output.0.weight"], grads[prefix + "mlp_output.0.bias"] = \
        ops.linear_bwd(cache["cat"], sd[prefix + "mlp_output.0.weight"], dz2)
    E = sd[f"{prefix}embeddings_list.0.weight"].shape[1] if nd else 0
    for j in range(nd):
        tab = sd[f"{prefix}embeddings_list.{j}.weight"]
        g = np.zeros_like(tab)
        np.add.at(g, cache["idx"][..., j].reshape(-1), dcat[..., j * E:(j + 1) * E].reshape(-1, E))
        grads[f"{prefix}embeddings_list.{j}.weight"] = g
    if nn_:
        dy1 = dcat[..., nd * E:]
        da1, grads[prefix + "mlp_nums.2.weight"], grads[prefix + "mlp_nums.2.bias"] = \
            ops.linear_bwd(cache["a1"], sd[prefix + "mlp_nums.2.weight"], dy1)
        dz1 = da1 * ops.silu_grad(cache["z1"])
        _, grads[prefix + "mlp_nums.0.weight"], grads[prefix + "mlp_nums.0.bias"] = \
            ops.linear_bwd(cache["feats"], sd[prefix + "mlp_nums.0.weight"], dz1, need_dx=False)


# ----------------------------------------------------------------------------
# BiRNN_score.forward -- DIFF.py:196-224.  target_mask is accepted and ignored
# exactly like the reference; Emb_hidden (DIFF.py:188) is never used.
def forward(sd, x, t, i, time_info, cond, schema: Schema, diffusion_steps=100, num_layers=2):
    dt = x.dtype
    H = sd["input_proj.net.0.weight"].shape[0]
    cache = {}
    xh = ops.linear_fwd(x, sd["input_proj.net.0.weight"], sd["input_proj.net.0.bias"])
    sc_t, sh_t = ops.posenc_tables(H, 1.0, dt)
    sc_i, sh_i = ops.posenc_tables(H, float(diffusion_steps), dt)
    te = ops.posenc_fwd(t, sd["t_enc.linear_proj.weight"], sd["t_enc.linear_proj.bias"], 1.0, sc_t, sh_t)
    ie = ops.posenc_fwd(i, sd["i_enc.linear_proj.weight"], sd["i_enc.linear_proj.bias"],
                        float(diffusion_steps), sc_i, sh_i)
    tz = ops.linear_fwd(time_info, sd["time_encode.0.weight"], sd["time_encode.0.bias"])
    ta = np.maximum(tz, 0)
    tenc = ops.linear_fwd(ta, sd["time_encode.2.weight"], sd["time_encode.2.bias"])
    emb, ecache = embedding_fwd(sd, "Emb.", cond, schema)
    assert emb.shape[1] == x.shape[1], "cond shorter than x in T is not supported"
    gout, gcache = rnn.rnn_fwd("gru", sd, "cond_GRU.", emb, num_layers, True)
    cout = ops.linear_fwd(gout, sd["cond_output.weight"], sd["cond_output.bias"])
    cat = np.concatenate([xh + cout, te, ie, tenc], -1)
    pz = ops.linear_fwd(cat, sd["proj.net.0.weight"], sd["proj.net.0.bias"])
    pa = np.maximum(pz, 0)
    lout, lcache = rnn.rnn_fwd("lstm", sd, "lstm.", pa, num_layers, True)
    ln, lncache = ops.layernorm_fwd(lout, sd["layer_norm.weight"], sd["layer_norm.bias"])
    out = ops.linear_fwd(ln, sd["fc.weight"], sd["fc.bias"])
    cache.update(x=x, t=t, i=i, time_info=time_info, ta=ta, ecache=ecache, emb=emb, gcache=gcache,
                 gout=gout, cat=cat, pa=pa, lcache=lcache, ln=ln, lncache=lncache,
                 num_layers=num_layers, diffusion_steps=diffusion_steps)
    return out, cache


def backward(sd, cache, dout, schema: Schema):
    """Gradients of all used parameters given d(loss)/d(output)."""
    g = {}
    H = sd["input_proj.net.0.weight"].shape[0]
    L = cache["num_layers"]
    dln, g["fc.weight"], g["fc.bias"] = ops.linear_bwd(cache["ln"], sd["fc.weight"], dout)
    dlout, g["layer_norm.weight"], g["layer_norm.bias"] = ops.layernorm_bwd(dln, sd["layer_norm.weight"],
                                                                         cache["lncache"])
    dpa = rnn.rnn_bwd("lstm", sd, "lstm.", cache["lcache"], dlout, L, True, g)
    dpz = dpa * (cache["pa"] > 0)
    dcat, g["proj.net.0.weight"], g["proj.net.0.bias"] = ops.linear_bwd(cache["cat"], sd["proj.net.0.weight"], dpz)
    dxh = dcat[..., :H]
    dte = dcat[..., H:2 * H]
    die = dcat[..., 2 * H:3 * H]
    dtenc = dcat[..., 3 * H:]
    _, g["input_proj.net.0.weight"], g["input_proj.net.0.bias"] = \
        ops.linear_bwd(cache["x"], sd["input_proj.net.0.weight"], dxh, need_dx=False)
    g["t_enc.linear_proj.weight"], g["t_enc.linear_proj.bias"] = \
        ops.posenc_bwd(cache["t"], sd["t_enc.linear_proj.weight"], dte, 1.0)
    g["i_enc.linear_proj.weight"], g["i_enc.linear_proj.bias"] = \
        ops.posenc_bwd(cache["i"], sd["i_enc.linear_proj.weight"], die, float(cache["diffusion_steps"]))
    dta, g["time_encode.2.weight"], g["time_encode.2.bias"] = \
        ops.linear_bwd(cache["ta"], sd["time_encode.2.weight"], dtenc)
    dtz = dta * (cache["ta"] > 0)
    _, g["time_encode.0.weight"], g["time_encode.0.bias"] = \
        ops.linear_bwd(cache["time_info"], sd["time_encode.0.weight"], dtz, need_dx=False)
    dgout, g["cond_output.weight"], g["cond_output.bias"] = ops.linear_bwd(cache["gout"], sd["cond_output.weight"], dxh)
    demb = rnn.rnn_bwd("gru", sd, "cond_GRU.", cache["gcache"], dgout, L, True, g)
    embedding_bwd(sd, "Emb.", cache["ecache"], demb, schema, g)
    return g


# ----------------------------------------------------------------------------
# get_loss -- DIFF.py:252-260 with the random draws (i, noise) passed in
def loss_and_grads(sd, x0, t, i_idx, noise, time_info, cond, schema, alphas, want_grads=True,
                   diffusion_steps=100, num_layers=2, global_numel=None):
    dt = x0.dtype
    B, T, K = x0.shape
    x_noisy = ops.add_noise(x0, i_idx, noise, alphas)
    i = np.broadcast_to(i_idx.astype(dt)[:, None, None], (B, T, 1))
    pred, cache = forward(sd, x_noisy, t, i, time_info, cond, schema, diffusion_steps, num_layers)
    diff = pred - noise
    loss = float(np.mean(diff.astype(np.float64) ** 2))
    if not want_grads:
        return loss, pred, None
    # under data parallelism each rank scales by the GLOBAL batch's element count
    dout = (2.0 / (global_numel if global_numel is not None else diff.size)) * diff
    grads = backward(sd, cache, dout.astype(dt), schema)
    return loss, pred, grads
