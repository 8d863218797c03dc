"""DeapStack (the mixed-type sequence VAE), its loss and their gradients in numpy.

Oracle code: test infrastructure only.  Parameters are a dict keyed by the
reference's ``state_dict`` names.
"""
import numpy as np

from . import ops, rnn
from .denoiser import Schema, embedding_fwd, embedding_bwd


def _head_slices(schema: Schema):
    """Column ranges of bins / each categorical / nums in the fused logits row."""
    o = 0
    sl = {}
    if schema.n_bins:
        sl["bins"] = (o, o + schema.n_bins)
        o += schema.n_bins
    sl["cats"] = []
    for c in (schema.cards if schema.n_cats else []):
        sl["cats"].append((o, o + c))
        o += c
    if schema.n_nums:
        sl["nums"] = (o, o + schema.n_nums)
        o += schema.n_nums
    sl["total"] = o
    return sl


# DeapStack.encoder -- VAE.py:152-160 (reparametrize VAE.py:147-150 with eps passed in)
def encoder_fwd(sd, x, eps, schema: Schema, num_layers=2):
    e, ecache = embedding_fwd(sd, "Emb.", x, schema)
    mu_h, mcache = rnn.rnn_fwd("gru", sd, "encoder_mu.", e, num_layers, False)
    lv_h, vcache = rnn.rnn_fwd("gru", sd, "encoder_logvar.", e, num_layers, False)
    mu = ops.linear_fwd(mu_h, sd["fc_mu.weight"], sd["fc_mu.bias"])
    lv = ops.linear_fwd(lv_h, sd["fc_logvar.weight"], sd["fc_logvar.bias"])
    std = np.exp(0.5 * lv)
    z = mu + eps * std
    cache = dict(ecache=ecache, mcache=mcache, vcache=vcache, mu_h=mu_h, lv_h=lv_h, std=std, eps=eps,
                 num_layers=num_layers)
    return z, mu, lv, cache


# DeapStack.decoder -- VAE.py:193-215 with decompose_target_mask VAE.py:162-191
def decoder_fwd(sd, z, mask, cond, schema: Schema):
    dt = z.dtype
    nb, nc, nn_ = schema.n_bins, schema.n_cats, schema.n_nums
    d1z = ops.linear_fwd(z, sd["decoder_mlp.0.weight"], sd["decoder_mlp.0.bias"])
    d1 = np.maximum(d1z, 0)
    lat = ops.linear_fwd(d1, sd["decoder_mlp.2.weight"], sd["decoder_mlp.2.bias"])
    cache = dict(z=z, d1=d1)
    if cond is not None:
        ce, cecache = embedding_fwd(sd, "Emb_decoder.", cond, schema)
        lat = lat + ops.linear_fwd(ce, sd["Emb_hidden_decoder.weight"], sd["Emb_hidden_decoder.bias"])
        cache.update(ce=ce, cecache=cecache)
    cache["lat"] = lat
    out = {}
    maskf = mask.astype(dt)
    if nb:
        mb = np.where(mask[..., :nb] == 0, dt.type(-100.0), dt.type(0.0))
        out["bins"] = ops.linear_fwd(lat, sd["bins_linear.weight"], sd["bins_linear.bias"]) + mb
    if nc:
        out["cats"] = []
        for j in range(nc):
            lg = ops.linear_fwd(lat, sd[f"cats_linears.{j}.weight"], sd[f"cats_linears.{j}.bias"]).copy()
            lg[..., 0] += np.where(mask[..., nb + j] == 0, dt.type(100.0), dt.type(0.0))
            out["cats"].append(lg)
    if nn_:
        a = ops.linear_fwd(lat, sd["nums_linear.weight"], sd["nums_linear.bias"])
        s = ops.sigmoid(a)
        mn = maskf[..., nb + nc:nb + nc + nn_]
        out["nums"] = s * mn
        cache.update(s=s, mn=mn)
    return out, cache


# auto_loss -- VAE.py:222-253, KLD VAE.py:350-352
def loss_fwd(out, target, mu, lv, beta, schema: Schema):
    """Returns (total, parts dict, dlogits dict) where dlogits are d(total)/d(head
    outputs) ('bins', 'cats'[j], 'nums'), d(total)/dmu and d(total)/dlv excluding
    the reparametrisation path."""
    dt = mu.dtype
    nb, nc, nn_ = schema.n_bins, schema.n_cats, schema.n_nums
    B, T, _ = target.shape
    parts = dict(bce=0.0, ce=0.0, mse=0.0)
    d = {}
    if nb:
        x = out["bins"].astype(np.float64)
        y = target[..., :nb].astype(np.float64)
        l = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
        parts["bce"] = float(l.mean())
        d["bins"] = ((ops.sigmoid(x) - y) / x.size).astype(dt)
    if nc:
        ces = []
        d["cats"] = []
        for j in range(nc):
            lg = out["cats"][j].astype(np.float64).reshape(B * T, -1)
            tgt = target[..., nb + j].astype(np.int64).reshape(B * T)
            m = lg.max(-1, keepdims=True)
            lse = m[:, 0] + np.log(np.exp(lg - m).sum(-1))
            ces.append(float((lse - lg[np.arange(B * T), tgt]).mean()))
            p = np.exp(lg - lse[:, None])
            p[np.arange(B * T), tgt] -= 1.0
            d["cats"].append((p / (B * T * nc)).reshape(out["cats"][j].shape).astype(dt))
        parts["ce"] = float(np.mean(ces))
    if nn_:
        diff = out["nums"].astype(np.float64) - target[..., nb + nc:nb + nc + nn_].astype(np.float64)
        parts["mse"] = float((diff ** 2).mean())
        d["nums"] = (2.0 * diff / diff.size).astype(dt)
    mu64 = mu.astype(np.float64)
    lv64 = lv.astype(np.float64)
    parts["kld"] = float(-0.5 * np.mean(1 + lv64 - mu64 ** 2 - np.exp(lv64)))
    n = mu.size
    d["mu"] = (beta * mu64 / n).astype(dt)
    d["lv"] = (beta * (-0.5) * (1 - np.exp(lv64)) / n).astype(dt)
    total = parts["mse"] + parts["bce"] + parts["ce"] + beta * parts["kld"]
    return total, parts, d


def decoder_bwd(sd, cache, dheads, schema: Schema, grads):
    nb, nc, nn_ = schema.n_bins, schema.n_cats, schema.n_nums
    lat = cache["lat"]
    dlat = np.zeros_like(lat)
    if nb:
        dx, grads["bins_linear.weight"], grads["bins_linear.bias"] = \
            ops.linear_bwd(lat, sd["bins_linear.weight"], dheads["bins"])
        dlat += dx
    for j in range(nc):
        dx, grads[f"cats_linears.{j}.weight"], grads[f"cats_linears.{j}.bias"] = \
            ops.linear_bwd(lat, sd[f"cats_linears.{j}.weight"], dheads["cats"][j])
        dlat += dx
    if nn_:
        s, mn = cache["s"], cache["mn"]
        da = dheads["nums"] * mn * s * (1 - s)
        dx, grads["nums_linear.weight"], grads["nums_linear.bias"] = ops.linear_bwd(lat, sd["nums_linear.weight"], da)
        dlat += dx
    if "ce" in cache:
        dce, grads["Emb_hidden_decoder.weight"], grads["Emb_hidden_decoder.bias"] = \
            ops.linear_bwd(cache["ce"], sd["Emb_hidden_decoder.weight"], dlat)
        embedding_bwd(sd, "Emb_decoder.", cache["cecache"], dce, schema, grads)
    dd1, grads["decoder_mlp.2.weight"], grads["decoder_mlp.2.bias"] = \
        ops.linear_bwd(cache["d1"], sd["decoder_mlp.2.weight"], dlat)
    dd1z = dd1 * (cache["d1"] > 0)
    dz, grads["decoder_mlp.0.weight"], grads["decoder_mlp.0.bias"] = \
        ops.linear_bwd(cache["z"], sd["decoder_mlp.0.weight"], dd1z)
    return dz


def encoder_bwd(sd, cache, dz, dmu_extra, dlv_extra, schema: Schema, grads):
    L = cache["num_layers"]
    dmu = dz + dmu_extra
    dlv = dz * cache["eps"] * 0.5 * cache["std"] + dlv_extra
    dmu_h, grads["fc_mu.weight"], grads["fc_mu.bias"] = ops.linear_bwd(cache["mu_h"], sd["fc_mu.weight"], dmu)
    dlv_h, grads["fc_logvar.weight"], grads["fc_logvar.bias"] = ops.linear_bwd(cache["lv_h"], sd["fc_logvar.weight"], dlv)
    de = rnn.rnn_bwd("gru", sd, "encoder_mu.", cache["mcache"], dmu_h, L, False, grads)
    de = de + rnn.rnn_bwd("gru", sd, "encoder_logvar.", cache["vcache"], dlv_h, L, False, grads)
    embedding_bwd(sd, "Emb.", cache["ecache"], de, schema, grads)


# DeapStack.forward + the loss of train_autoencoder's iteration -- VAE.py:217-220, 336-352
def loss_and_grads(sd, inp, mask, cond, eps, beta, schema: Schema, want_grads=True, num_layers=2):
    """inp = response + cond (VAE.py:331-334); target = inp * mask (VAE.py:336)."""
    dt = inp.dtype
    z, mu, lv, ecache = encoder_fwd(sd, inp, eps, schema, num_layers)
    out, dcache = decoder_fwd(sd, z, mask, cond, schema)
    target = inp * mask.astype(dt)
    total, parts, d = loss_fwd(out, target, mu, lv, beta, schema)
    res = dict(out=out, z=z, mu=mu, lv=lv, parts=parts)
    if not want_grads:
        return total, res, None
    grads = {}
    dz = decoder_bwd(sd, dcache, d, schema, grads)
    encoder_bwd(sd, ecache, dz, d["mu"], d["lv"], schema, grads)
    return total, res, grads
