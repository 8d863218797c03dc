#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference.

Run in the build container only (the reference checkout is not available on
the GPU box):  python tests/golden/make_fixtures.py [--ref /root/reference]

The reference's hot-path modules import ``dataprovider`` which imports
``pygrinder`` (not installed, unused on the path); an empty stand-in module is
placed in ``sys.modules`` for the import.  Nothing of the reference is copied:
the outputs are data only (inputs, random draws or their seeds, outputs,
gradients, loss curves).

Conventions shared with tests/fixture_util.py:
  * synthetic inputs come from ``fixture_util.make_data`` (numpy, seeded);
  * model weights come from ``torch.manual_seed(seed)`` followed by the model
    constructor; small cases store the weights, the medium case stores per
    tensor checksums so a re-created model can be verified;
  * random draws inside the loops follow the reference's consumption order of
    python ``random`` and torch's CPU generator and are replayed from seeds.
"""
import argparse
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fixture_util as fu  # noqa: E402


def import_reference(ref):
    sys.dont_write_bytecode = True
    stub = types.ModuleType("pygrinder")
    for a in ("mcar", "seq_missing", "block_missing"):
        setattr(stub, a, lambda *a, **k: None)
    sys.modules["pygrinder"] = stub
    sys.path.insert(0, ref)
    import torch  # noqa
    import DIFF
    import VAE
    return DIFF, VAE


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def grads_to_np(model):
    return {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}


def stats(a):
    a64 = a.astype(np.float64).ravel()
    return np.array([a64.sum(), np.sqrt((a64 ** 2).sum()), a64[:8].sum() if a64.size else 0.0])


# ----------------------------------------------------------------------------
def denoiser_case(DIFF, name, schema, H, E, T, B, K, seed, store_weights=True):
    import torch
    F = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    data = fu.make_data(schema, N=B, T=T, K=K, seed=seed, mask_kind="bern")
    torch.manual_seed(seed)
    model = DIFF.BiRNN_score(K, F, H, 2, 100, 8, E, schema["n_bins"], schema["n_cats"], schema["n_nums"],
                             list(schema["cards"]))
    sd0 = sd_to_np(model.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    t = torch.rand(B, T, 1, generator=g).sort(1)[0]
    i_idx = torch.randint(0, 100, (B,), generator=g)
    noise = torch.randn(B, T, K, generator=g)
    x0 = torch.from_numpy(data["latent"])
    cond = torch.from_numpy(data["cond"])
    tinfo = torch.from_numpy(data["time_info"])
    i = i_idx.view(-1, 1, 1).expand_as(x0[..., :1]).to(x0)
    alpha = DIFF.alphas[i.long()].to(x0)
    x_noisy = torch.sqrt(alpha) * x0 + torch.sqrt(1 - alpha) * noise
    # check add_noise's arithmetic agrees (its own randn is replaced by ours)
    orig = torch.randn_like
    torch.randn_like = lambda x: noise
    try:
        xn2, n2 = DIFF.add_noise(x0, t, i)
    finally:
        torch.randn_like = orig
    assert torch.equal(xn2, x_noisy) and torch.equal(n2, noise)
    pred = model(x_noisy, t, i, tinfo, cond, None)
    loss = torch.mean((pred - noise) ** 2)
    loss.backward()
    grads = grads_to_np(model)
    out = dict(meta=fu.pack_meta(dict(kind="denoiser", schema=schema, H=H, E=E, T=T, B=B, K=K, seed=seed,
                                      store_weights=store_weights)),
               t=t.numpy(), i_idx=i_idx.numpy(), noise=noise.numpy(), x_noisy=x_noisy.numpy(),
               pred=pred.detach().numpy(), loss=np.float64(loss.item()))
    if store_weights:
        for k, v in sd0.items():
            out["w/" + k] = v
        for k, v in grads.items():
            out["g/" + k] = v
    else:
        for k, v in sd0.items():
            out["ws/" + k] = stats(v)
        for k, v in grads.items():
            out["gs/" + k] = stats(v)
            out["gh/" + k] = v.ravel()[:64].copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss.item(), "params", sum(v.size for v in sd0.values()))


# ----------------------------------------------------------------------------
def vae_case(VAE, name, schema, H, E, L, T, B, seed, mask_kind):
    import torch
    F = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    data = fu.make_data(schema, N=B, T=T, K=L, seed=seed, mask_kind=mask_kind)
    torch.manual_seed(seed)
    ae = VAE.DeapStack(1, schema["n_bins"], schema["n_cats"], schema["n_nums"], list(schema["cards"]),
                       F, H, 2, E, 8, L)
    sd0 = sd_to_np(ae.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    eps = torch.randn(B, T, L, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda x: eps
    try:
        resp = torch.from_numpy(data["response"])
        cond = torch.from_numpy(data["cond"])
        mask = torch.from_numpy(data["mask"])
        inp = resp + cond
        target = inp * mask
        out, z, mu, lv = ae(inp, mask, cond)
    finally:
        torch.randn_like = orig
    disc, num = VAE.auto_loss(target, out, schema["n_bins"], schema["n_nums"], schema["n_cats"], list(schema["cards"]))
    kld = -0.5 * torch.mean((1 + lv - mu.pow(2) - lv.exp()).mean(-1).mean())
    beta = 0.1
    loss = num + disc + beta * kld
    loss.backward()
    grads = grads_to_np(ae)
    o = dict(meta=fu.pack_meta(dict(kind="vae", schema=schema, H=H, E=E, L=L, T=T, B=B, seed=seed,
                                    mask_kind=mask_kind, beta=beta)),
             eps=eps.numpy(), z=z.detach().numpy(), mu=mu.detach().numpy(), lv=lv.detach().numpy(),
             loss=np.float64(loss.item()), disc=np.float64(float(disc)), num=np.float64(float(num)),
             kld=np.float64(kld.item()))
    if "bins" in out:
        o["out/bins"] = out["bins"].detach().numpy()
    if "cats" in out:
        for j, c in enumerate(out["cats"]):
            o[f"out/cats{j}"] = c.detach().numpy()
    if "nums" in out:
        o["out/nums"] = out["nums"].detach().numpy()
    for k, v in sd0.items():
        o["w/" + k] = v
    for k, v in grads.items():
        o["g/" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **o)
    print(name, "loss", loss.item(), "params", sum(v.size for v in sd0.values()))


# ----------------------------------------------------------------------------
class _Parser:
    def __init__(self, schema):
        self.s = schema

    def datatype_info(self):
        return dict(n_bins=self.s["n_bins"], n_cats=self.s["n_cats"], n_nums=self.s["n_nums"],
                    cards=list(self.s["cards"]))


def diffusion_loop_case(DIFF, name, schema, H, T, K, N, steps, seed):
    """train_diffusion for ``steps`` iterations then sample; emb_dim is the
    reference's hard-coded 128 and the batch its hard-coded 100."""
    import torch
    data = fu.make_data(schema, N=N, T=T, K=K, seed=seed, mask_kind="ones")
    dd = dict(parser=_Parser(schema), cond_train=torch.from_numpy(data["cond"]),
              time_info_train=torch.from_numpy(data["time_info"]),
              target_mask_train=torch.from_numpy(data["mask"]))
    losses = []
    orig = DIFF.get_loss

    def rec(*a, **k):
        l = orig(*a, **k)
        losses.append(float(l.item()))
        return l
    DIFF.get_loss = rec
    try:
        random.seed(seed)
        torch.manual_seed(seed)
        model = DIFF.train_diffusion(torch.from_numpy(data["latent"]), None, dd, H, 2, 100, steps)
    finally:
        DIFF.get_loss = orig
    sd1 = sd_to_np(model.state_dict())
    # sampling with the trained model
    Bs = 3
    torch.manual_seed(seed + 7)
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(Bs, 1, 1)
    x = DIFF.sample(tgrid, Bs, T, K, 100, model, torch.from_numpy(data["time_info"][:Bs]),
                    torch.from_numpy(data["cond"][:Bs]), None)
    out = dict(meta=fu.pack_meta(dict(kind="diffusion_loop", schema=schema, H=H, T=T, K=K, N=N, steps=steps,
                                      seed=seed, Bs=Bs)),
               losses=np.array(losses), sample=x.numpy())
    for k, v in sd1.items():
        out["ws/" + k] = stats(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "losses", losses[:3], "...", losses[-1])


def sampler_case(DIFF, name, schema, H, E, T, K, B, seed):
    """sample() on a freshly initialised small model; keeps x after steps 99, 50, 1, 0."""
    import torch
    F = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    data = fu.make_data(schema, N=B, T=T, K=K, seed=seed, mask_kind="ones")
    torch.manual_seed(seed)
    model = DIFF.BiRNN_score(K, F, H, 2, 100, 8, E, schema["n_bins"], schema["n_cats"], schema["n_nums"],
                             list(schema["cards"]))
    sd0 = sd_to_np(model.state_dict())
    kept = {}
    calls = {"n": 0}
    orig_fwd = model.forward

    def fwd(x, *a, **k):
        # x entering call number n is the state after reverse step 100-n
        s_done = 100 - calls["n"]
        if s_done in (99, 50, 1):
            kept[s_done] = x.detach().numpy().copy()
        calls["n"] += 1
        return orig_fwd(x, *a, **k)
    model.forward = fwd
    torch.manual_seed(seed + 7)
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(B, 1, 1)
    x = DIFF.sample(tgrid, B, T, K, 100, model, torch.from_numpy(data["time_info"]),
                    torch.from_numpy(data["cond"]), None)
    out = dict(meta=fu.pack_meta(dict(kind="sampler", schema=schema, H=H, E=E, T=T, K=K, B=B, seed=seed)),
               x_final=x.numpy(), x_after_99=kept[99], x_after_50=kept[50], x_after_1=kept[1])
    for k, v in sd0.items():
        out["w/" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "x_final norm", float(np.linalg.norm(x.numpy())))


def vae_loop_case(VAE, name, schema, H, T, L, N, Nv, B, steps, seed, fn_name):
    import torch
    F = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    tr = fu.make_data(schema, N=N, T=T, K=L, seed=seed, mask_kind="bern")
    va = fu.make_data(schema, N=Nv, T=T, K=L, seed=seed + 100, mask_kind="bern")
    dd = dict(parser=_Parser(schema),
              response_train=torch.from_numpy(tr["response"]), cond_train=torch.from_numpy(tr["cond"]),
              target_mask_train=torch.from_numpy(tr["mask"]),
              response_val=torch.from_numpy(va["response"]), cond_val=torch.from_numpy(va["cond"]),
              target_mask_val=torch.from_numpy(va["mask"]))
    random.seed(seed)
    torch.manual_seed(seed)
    fn = getattr(VAE, fn_name)
    ae, latent, output, hist = fn(None, 1, H, 2, 2e-4, 1e-6, steps, B, 1, 1e-5, 0.1, 16, 8, L, "cpu", dd)
    o = dict(meta=fu.pack_meta(dict(kind="vae_loop", schema=schema, H=H, T=T, L=L, N=N, Nv=Nv, B=B, steps=steps,
                                    seed=seed, fn=fn_name, lr=2e-4, wd=1e-6, min_beta=1e-5, max_beta=0.1, E=16)),
             train=np.array(hist["train"]), val=np.array(hist["val"]), latent=latent.numpy())
    for k, v in sd_to_np(ae.state_dict()).items():
        o["ws/" + k] = stats(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **o)
    print(name, "train", hist["train"][:2], hist["train"][-1], "val", hist["val"][-1])


def convert_case(ref, name, schema, B, T, seed):
    """process_edited.convert_to_tensor (process_edited.py:252-301) on seeded decoder outputs that include exact
    ties, zeros and near-threshold logits.  The function reads the module global ``device`` (hard-coded 'cuda',
    process_edited.py:250) at call time; it is pointed at 'cpu' for this run, the source is untouched."""
    import torch
    sys.path.insert(0, ref)
    import process_edited as pce
    pce.device = "cpu"
    rng = np.random.default_rng(seed)
    nb, nc, nn_, cards = schema["n_bins"], schema["n_cats"], schema["n_nums"], list(schema["cards"])
    gen, save = {}, {}
    if nb:
        x = (rng.standard_normal((B, T, nb)) * 2).astype(np.float32)
        x.flat[::7] = 0.0
        x.flat[3::11] = np.float32(1e-3) * rng.standard_normal(x.flat[3::11].shape).astype(np.float32)
        x.flat[5::13] -= 100.0                                    # masked-out entries carry the -100 offset
        gen["bins"] = torch.from_numpy(x)
        save["in/bins"] = x
    if cards:
        gen["cats"] = []
        for j, c in enumerate(cards):
            x = rng.standard_normal((B, T, c)).astype(np.float32)
            if c > 1:
                x[::3, :, c - 1] = x[::3].max(-1)                # exact ties between the last class and the maximum
                x[1::4, ::2, 0] += 100.0                          # masked-out entries: +100 on class 0
            gen["cats"].append(torch.from_numpy(x))
            save[f"in/cats{j}"] = x
    if nn_:
        x = rng.random((B, T, nn_), dtype=np.float32)
        gen["nums"] = torch.from_numpy(x)
        save["in/nums"] = x

    class P:
        def datatype_info(self):
            return dict(schema)
    out = pce.convert_to_tensor(P(), gen, 1, B, T)
    out2 = pce.convert_to_tensor(P(), gen, 1, None, None)
    assert torch.equal(out, out2)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), out=out.numpy().astype(np.float32),
                        meta=fu.pack_meta(dict(schema=schema, B=B, T=T, seed=seed)), **save)


def windows_case(ref, name, rows, F, seq_len, stride, B, seed):
    """dataprovider_pypots.sliding_window (dataprovider_pypots.py:75-87) on a seeded table, plus the minibatch the
    trainers take from its result by list indexing (DIFF.py:320, VAE.py:340-348)."""
    import random
    import torch
    sys.path.insert(0, ref)
    import dataprovider_pypots as dpp          # needs the pygrinder stand-in import_reference() installs
    rng = np.random.default_rng(seed)
    table = rng.standard_normal((rows, F)).astype(np.float32)
    stacked = dpp.sliding_window(torch.from_numpy(table), seq_len, stride)
    random.seed(seed)
    idx = random.sample(range(stacked.shape[0]), B)
    batch = stacked[idx]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), table=table, stacked=stacked.numpy(), idx=np.asarray(idx, np.int64),
                        batch=batch.numpy(), meta=fu.pack_meta(dict(rows=rows, F=F, seq_len=seq_len, stride=stride, B=B, seed=seed)))
    print(name, tuple(stacked.shape))


def tvmcg_case(ref, name, B, T, F, cont_idx, cat_idx, classes, level, p, seed):
    """The TV-MCG corruption helpers (dataprovider_pypots.py:149-229) run by the reference on a seeded cond tensor,
    with the draws they took from torch's CPU generator (replayed from the same seed in the same order)."""
    import torch
    sys.path.insert(0, ref)
    import dataprovider_pypots as dpp
    rng = np.random.default_rng(seed)
    cond = rng.standard_normal((B, T, F)).astype(np.float32) * rng.uniform(0.5, 20, F).astype(np.float32) + 3
    for c in cat_idx:
        cond[:, :, c] = rng.integers(0, max(classes[c], 1), (B, T)).astype(np.float32)
    ct = torch.from_numpy(cond)
    torch.manual_seed(seed)
    noisy = dpp.add_continuous_metadata_noise_subset(ct, list(cont_idx), level)
    torch.manual_seed(seed)
    noise = torch.randn(B, T, len(cont_idx))
    torch.manual_seed(seed + 1)
    flipped = dpp.flip_categorical_metadata_subset(ct, list(cat_idx), p, dict(classes))
    torch.manual_seed(seed + 1)
    rand, base = [], []
    for c in cat_idx:
        if classes[c] <= 1:
            continue
        rand.append(torch.rand(B, T))
        base.append(torch.randint(0, classes[c] - 1, (B, T)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cond=cond, noisy=noisy.numpy(), noise=noise.numpy(),
                        flipped=flipped.numpy(), rand=np.stack([r.numpy() for r in rand]) if rand else np.zeros((0, B, T), np.float32),
                        base=np.stack([b.numpy() for b in base]) if base else np.zeros((0, B, T), np.int64),
                        meta=fu.pack_meta(dict(B=B, T=T, F=F, cont_idx=list(cont_idx), cat_idx=list(cat_idx),
                                               classes={str(k): v for k, v in classes.items()}, level=level, p=p, seed=seed)))
    print(name, float((noisy - ct).abs().max()), int((flipped != ct).sum()))


def posthoc_cases(ref):
    """The post-hoc evaluation networks (reference Evaluation/Metrics.py:54-65 Discriminator,
    Evaluation/predictive_metrics.py:8-17 PosthocRNN): forward, loss, every gradient, and a 20-step Adam loss curve
    on recorded minibatch indices (the loop bodies of Metrics.py:91-109 / predictive_metrics.py:30-41)."""
    import importlib.util
    import torch
    import torch.nn as nn

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    M = load(os.path.join(ref, "Evaluation", "Metrics.py"), "ref_metrics")
    P = load(os.path.join(ref, "Evaluation", "predictive_metrics.py"), "ref_predictive")
    rng = np.random.default_rng(301)
    N, T, dim, nb, steps = 48, 8, 5, 16, 20
    real = rng.random((N, T, dim), dtype=np.float32)
    fake = (rng.random((N, T, dim), dtype=np.float32) * 0.8 + 0.1).astype(np.float32)
    # ---- discriminator
    torch.manual_seed(302)
    disc = M.Discriminator(dim, 50, 2)
    out = dict(real=real, fake=fake)
    for k, v in sd_to_np(disc.state_dict()).items():
        out["w/" + k] = v
    xr, xf = torch.from_numpy(real[:nb]), torch.from_numpy(fake[:nb])
    lr_, _ = disc(xr)
    lf_, _ = disc(xf)
    loss = nn.BCEWithLogitsLoss()(lr_, torch.ones_like(lr_)) + nn.BCEWithLogitsLoss()(lf_, torch.zeros_like(lf_))
    loss.backward()
    out["logit_real"], out["logit_fake"], out["loss"] = lr_.detach().numpy(), lf_.detach().numpy(), np.float64(loss.item())
    for k, v in grads_to_np(disc).items():
        out["g/" + k] = v
    disc.zero_grad()
    opt = torch.optim.Adam(disc.parameters())
    ir = np.stack([rng.permutation(N)[:nb] for _ in range(steps)])
    ih = np.stack([rng.permutation(N)[:nb] for _ in range(steps)])
    curve = []
    for s_ in range(steps):
        opt.zero_grad()
        a_, _ = disc(torch.from_numpy(real[ir[s_]]))
        b_, _ = disc(torch.from_numpy(fake[ih[s_]]))
        l_ = nn.BCEWithLogitsLoss()(a_, torch.ones_like(a_)) + nn.BCEWithLogitsLoss()(b_, torch.zeros_like(b_))
        l_.backward()
        opt.step()
        curve.append(l_.item())
    out["idx_real"], out["idx_fake"], out["curve"] = ir.astype(np.int64), ih.astype(np.int64), np.array(curve)
    for k, v in sd_to_np(disc.state_dict()).items():
        out["w_end/" + k] = v
    out["meta"] = fu.pack_meta(dict(kind="posthoc_disc", N=N, T=T, dim=dim, nb=nb, steps=steps, hidden=50, layers=2))
    np.savez_compressed(os.path.join(HERE, "posthoc_disc.npz"), **out)
    # ---- predictor: column 2 one step ahead from the others
    col = 2
    torch.manual_seed(303)
    pr = P.PosthocRNN(dim, int(dim / 2))
    out = dict(fake=fake)
    for k, v in sd_to_np(pr.state_dict()).items():
        out["w/" + k] = v
    g = torch.from_numpy(fake)

    def xy(idx):
        X = torch.cat((g[idx, :-1, :col], g[idx, :-1, col + 1:]), dim=2)
        return X, g[idx, 1:, col].unsqueeze(-1)
    X, Y = xy(np.arange(nb))
    p_ = pr(X)
    loss = nn.L1Loss()(p_, Y)
    loss.backward()
    out["pred"], out["loss"] = p_.detach().numpy(), np.float64(loss.item())
    for k, v in grads_to_np(pr).items():
        out["g/" + k] = v
    pr.zero_grad()
    opt = torch.optim.Adam(pr.parameters())
    idx = np.stack([rng.permutation(N)[:nb] for _ in range(steps)])
    curve = []
    for s_ in range(steps):
        X, Y = xy(idx[s_])
        opt.zero_grad()
        l_ = nn.L1Loss()(pr(X), Y)
        l_.backward()
        opt.step()
        curve.append(l_.item())
    out["idx"], out["curve"] = idx.astype(np.int64), np.array(curve)
    for k, v in sd_to_np(pr.state_dict()).items():
        out["w_end/" + k] = v
    out["meta"] = fu.pack_meta(dict(kind="posthoc_pred", N=N, T=T, dim=dim, nb=nb, steps=steps, col=col))
    np.savez_compressed(os.path.join(HERE, "posthoc_pred.npz"), **out)
    print("posthoc_disc / posthoc_pred written")


def schedule_case(DIFF):
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), betas=DIFF.betas.numpy(), alphas=DIFF.alphas.numpy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="'posthoc' / 'convert' / 'windows' / 'tvmcg': write only those fixtures (and the ones listed before them)")
    a = ap.parse_args()
    DIFF, VAE = import_reference(a.ref)
    import torch
    torch.set_num_threads(4)
    posthoc_cases(a.ref)
    if a.only == "posthoc":
        return
    S = fu.SCHEMAS
    windows_case(a.ref, "windows_stride1", rows=60, F=8, seq_len=24, stride=1, B=16, seed=71)
    windows_case(a.ref, "windows_stride12", rows=131, F=13, seq_len=48, stride=12, B=5, seed=72)
    windows_case(a.ref, "windows_exact", rows=24, F=3, seq_len=24, stride=5, B=1, seed=73)
    if a.only == "windows":
        return
    tvmcg_case(a.ref, "tvmcg_mixed", B=7, T=24, F=9, cont_idx=[3, 5, 6], cat_idx=[2, 4], classes={2: 5, 4: 2}, level=0.2, p=0.4, seed=81)
    tvmcg_case(a.ref, "tvmcg_skip", B=4, T=12, F=6, cont_idx=[0], cat_idx=[1, 5, 3], classes={1: 1, 5: 12, 3: 3}, level=0.4, p=0.1, seed=82)
    if a.only == "tvmcg":
        return
    convert_case(a.ref, "convert_mixed", S["mixed"], B=6, T=8, seed=61)
    convert_case(a.ref, "convert_metro", S["metro"], B=5, T=6, seed=62)
    convert_case(a.ref, "convert_numeric", S["numeric"], B=4, T=5, seed=63)
    if a.only == "convert":
        return
    schedule_case(DIFF)
    denoiser_case(DIFF, "den_numeric_small", S["numeric"], H=24, E=16, T=8, B=5, K=6, seed=11)
    denoiser_case(DIFF, "den_mixed_small", S["mixed"], H=24, E=16, T=8, B=5, K=6, seed=12)
    denoiser_case(DIFF, "den_metro_small", S["metro"], H=40, E=16, T=6, B=3, K=4, seed=13)
    denoiser_case(DIFF, "den_numeric_h200", S["numeric"], H=200, E=128, T=24, B=4, K=6, seed=14,
                  store_weights=False)
    vae_case(VAE, "vae_numeric_small", S["numeric"], H=24, E=16, L=6, T=8, B=5, seed=21, mask_kind="ones")
    vae_case(VAE, "vae_mixed_small", S["mixed"], H=24, E=16, L=6, T=8, B=5, seed=22, mask_kind="bern")
    vae_case(VAE, "vae_metro_small", S["metro"], H=40, E=16, L=4, T=6, B=3, seed=23, mask_kind="forecast")
    sampler_case(DIFF, "sampler_mixed_small", S["mixed"], H=24, E=16, T=8, K=6, B=3, seed=31)
    diffusion_loop_case(DIFF, "diffloop_numeric", S["numeric"], H=16, T=6, K=4, N=128, steps=20, seed=41)
    vae_loop_case(VAE, "vaeloop_mixed_val", S["mixed"], H=24, T=6, L=4, N=64, Nv=16, B=16, steps=30, seed=51,
                  fn_name="train_autoencoder")
    vae_loop_case(VAE, "vaeloop_mixed_train", S["mixed"], H=24, T=6, L=4, N=64, Nv=16, B=16, steps=30, seed=52,
                  fn_name="train_autoencoder_best_train")


if __name__ == "__main__":
    main()
