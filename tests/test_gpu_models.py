"""Model-level parity on the GPU: the HIP path (through the drop-in Python interface and
the C ABI) against golden vectors produced by the reference, against the numpy oracle on
fresh seeded inputs, and through size-independent properties at full benchmark sizes.

Tolerances: fp32 path 1e-4 relative (north star); bf16 path stated per test."""
import random

import numpy as np
import pytest
import torch

import fixture_util as fu
from oracle import denoiser as oden
from oracle import ops as oops
from oracle import vae as ovae
from oracle.denoiser import Schema

pytestmark = pytest.mark.gpu

TOL = 1e-4
BF16_OUT_TOL = 3e-2      # bf16 path vs the fp32 reference: outputs / loss, max-norm relative
BF16_GRAD_TOL = 1e-1     # bf16 path vs the fp32 reference fixture: gradient norms (informative bound; the binding check
                         # of the bf16 path is the emulating oracle below, at BF16_EMU_GRAD_TOL)
BF16_EMU_TOL = 5e-3      # bf16 path vs the oracle with operands rounded to bf16 at the same points: outputs
BF16_EMU_GRAD_TOL = 2e-2  # ... gradients, relative L2 per tensor


def grad_err(a, b, dtype):
    """fp32: max-norm relative error; bf16: relative L2 error of the tensor."""
    if dtype == "fp32":
        return fu.rel_err(a, b)
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def dev(lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _schema(meta):
    s = meta["schema"]
    return Schema(s["n_bins"], s["n_cats"], s["n_nums"], list(s["cards"]))


def T_(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _make_denoiser(meta, sch, dev, dtype, sd=None):
    from timeautodiff_amd import DIFF
    F = sch.n_bins + sch.n_cats + sch.n_nums
    torch.manual_seed(meta["seed"])
    m = DIFF.BiRNN_score(meta["K"], F, meta["H"], 2, 100, 8, meta["E"], sch.n_bins, sch.n_cats, sch.n_nums, sch.cards)
    if sd is not None:
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.set_compute_dtype(dtype)
    return m


def _denoiser_step(m, dev, data, t, i_idx, noise):
    from timeautodiff_amd import DIFF
    eng = m.engine(dev)
    eng.fp.zero_grad()
    ls = torch.zeros(1, device=dev)
    pred = eng.loss_and_grads(T_(data["latent"], dev), T_(t, dev), T_(i_idx.astype(np.int32), dev), T_(noise, dev),
                              T_(data["time_info"], dev), T_(data["cond"], dev), DIFF.alphas.to(dev), ls)
    torch.cuda.synchronize()
    grads = {k: eng.fp.g(k).cpu().numpy() for k in eng.fp.offsets if eng.fp.offsets[k] < eng.fp.n_active}
    return pred.cpu().numpy(), ls.item() / pred.numel(), grads


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["den_numeric_small", "den_mixed_small", "den_metro_small"])
def test_denoiser_vs_reference_golden(dev, name, dtype):
    z, meta = fu.load(name)
    sch = _schema(meta)
    sd = fu.sub(z, "w")
    m = _make_denoiser(meta, sch, dev, dtype, sd)
    assert list(m.state_dict().keys()) == list(sd.keys())           # same keys, same order as the reference
    data = fu.make_data(meta["schema"], N=meta["B"], T=meta["T"], K=meta["K"], seed=meta["seed"], mask_kind="bern")
    pred, loss, grads = _denoiser_step(m, dev, data, z["t"], z["i_idx"], z["noise"])
    gref = fu.sub(z, "g")
    assert set(gref) == set(grads)
    if dtype == "fp32":
        assert fu.rel_err(pred, z["pred"]) < TOL
        assert abs(loss - float(z["loss"])) < TOL * float(z["loss"])
        errs = {k: grad_err(grads[k], gref[k], dtype) for k in gref}
        print("worst grad errors:", sorted(errs.items(), key=lambda kv: -kv[1])[:4])
        assert max(errs.values()) < TOL, errs
    else:
        # (a) against the oracle with the SAME rounding points (operands of every matrix product
        # rounded to bf16, fp32 accumulation): only accumulation order and rare rounding flips differ
        _, alphas = oops.schedule(100)
        with oops.bf16_operands(storage=True):
            oloss, opred, ograds = oden.loss_and_grads(sd, data["latent"], z["t"], z["i_idx"], z["noise"], data["time_info"],
                                                       data["cond"], sch, alphas)
        assert fu.rel_err(pred, opred) < BF16_EMU_TOL
        assert abs(loss - oloss) < BF16_EMU_TOL * oloss
        errs = {k: grad_err(grads[k], ograds[k].reshape(grads[k].shape), dtype) for k in gref}
        print("worst grad errors vs bf16-emulating oracle:", sorted(errs.items(), key=lambda kv: -kv[1])[:4])
        assert max(errs.values()) < BF16_EMU_GRAD_TOL, errs
        # (b) against the fp32 reference: the stated looser bf16 tolerance on outputs and loss
        assert fu.rel_err(pred, z["pred"]) < BF16_OUT_TOL
        assert abs(loss - float(z["loss"])) < BF16_OUT_TOL * float(z["loss"])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_denoiser_h200_vs_reference_golden(dev, dtype):
    """The stock_data shape (H=200, T=24, emb 128): weights re-created from the seed."""
    z, meta = fu.load("den_numeric_h200")
    sch = _schema(meta)
    m = _make_denoiser(meta, sch, dev, dtype)
    ws = fu.sub(z, "ws")
    for k, v in m.state_dict().items():
        assert np.allclose(fu.stats(v.cpu().numpy()), ws[k], rtol=1e-12, atol=1e-12), k
    data = fu.make_data(meta["schema"], N=meta["B"], T=meta["T"], K=meta["K"], seed=meta["seed"], mask_kind="bern")
    pred, loss, grads = _denoiser_step(m, dev, data, z["t"], z["i_idx"], z["noise"])
    ot, gt = (TOL, TOL) if dtype == "fp32" else (BF16_OUT_TOL, BF16_GRAD_TOL)
    assert fu.rel_err(pred, z["pred"]) < ot
    if dtype == "bf16":
        # the binding check at this hidden size: the oracle with operands rounded to bf16 at the kernels' rounding
        # points, 5e-3 on the output and 2e-2 (relative L2) on every gradient tensor
        sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
        _, alphas = oops.schedule(100)
        with oops.bf16_operands(storage=True):
            oloss, opred, ograds = oden.loss_and_grads(sd, data["latent"], z["t"], z["i_idx"], z["noise"], data["time_info"],
                                                       data["cond"], sch, alphas)
        assert fu.rel_err(pred, opred) < BF16_EMU_TOL and abs(loss - oloss) < BF16_EMU_TOL * oloss
        errs = {k: grad_err(grads[k], v.reshape(grads[k].shape), dtype) for k, v in ograds.items()}
        assert max(errs.values()) < BF16_EMU_GRAD_TOL, sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    gs, gh = fu.sub(z, "gs"), fu.sub(z, "gh")
    assert set(gs) == set(grads)
    for k in gs:
        n = gs[k][1]
        assert abs(np.linalg.norm(grads[k].astype(np.float64)) - n) < gt * max(n, 1e-12), k
        assert np.max(np.abs(grads[k].ravel()[:64] - gh[k])) < gt * max(np.max(np.abs(grads[k])), 1e-12), k


def test_denoiser_module_call_and_state_dict(dev):
    """Drop-in surface: model(x, t, i, time_info, cond, mask), state_dict round trip, in-place edits."""
    z, meta = fu.load("den_mixed_small")
    sch = _schema(meta)
    sd = fu.sub(z, "w")
    m = _make_denoiser(meta, sch, dev, "fp32", sd)
    data = fu.make_data(meta["schema"], N=meta["B"], T=meta["T"], K=meta["K"], seed=meta["seed"], mask_kind="bern")
    B, T = meta["B"], meta["T"]
    i = torch.from_numpy(z["i_idx"]).view(-1, 1, 1).expand(B, T, 1).float()
    out = m(torch.from_numpy(z["x_noisy"]), torch.from_numpy(z["t"]), i, torch.from_numpy(data["time_info"]),
            torch.from_numpy(data["cond"]), None)
    assert out.is_cuda and fu.rel_err(out.cpu().numpy(), z["pred"]) < TOL
    sd2 = m.state_dict()
    for k in sd:
        assert np.array_equal(sd2[k].cpu().numpy(), sd[k]), k
    # a second model loaded from the first gives the same output; zeroed fc gives the bias
    m2 = _make_denoiser(dict(meta, seed=999), sch, dev, "fp32")
    m2.load_state_dict(sd2)
    out2 = m2(torch.from_numpy(z["x_noisy"]), torch.from_numpy(z["t"]), i, torch.from_numpy(data["time_info"]),
              torch.from_numpy(data["cond"]), None)
    assert torch.equal(out, out2)
    with torch.no_grad():
        m2.fc.weight.zero_()
    out3 = m2(torch.from_numpy(z["x_noisy"]), torch.from_numpy(z["t"]), i, torch.from_numpy(data["time_info"]),
              torch.from_numpy(data["cond"]), None)
    assert fu.rel_err(out3.cpu().numpy(), np.broadcast_to(sd["fc.bias"], out3.shape)) < 1e-6


# ------------------------------------------------------------------ VAE
def _make_vae(meta, sch, dev, dtype, sd=None):
    from timeautodiff_amd import VAE
    F = sch.n_bins + sch.n_cats + sch.n_nums
    torch.manual_seed(meta["seed"])
    ae = VAE.DeapStack(1, sch.n_bins, sch.n_cats, sch.n_nums, sch.cards, F, meta["H"], 2, meta["E"], 8, meta["L"])
    if sd is not None:
        ae.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ae.set_compute_dtype(dtype)
    return ae


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["vae_numeric_small", "vae_mixed_small", "vae_metro_small"])
def test_vae_vs_reference_golden(dev, name, dtype):
    z, meta = fu.load(name)
    sch = _schema(meta)
    sd = fu.sub(z, "w")
    ae = _make_vae(meta, sch, dev, dtype, sd)
    assert list(ae.state_dict().keys()) == list(sd.keys())
    data = fu.make_data(meta["schema"], N=meta["B"], T=meta["T"], K=meta["L"], seed=meta["seed"], mask_kind=meta["mask_kind"])
    B, T = meta["B"], meta["T"]
    rows = B * T
    eng = ae.engine(dev)
    inp = T_(data["response"] + data["cond"], dev)
    mask = T_(data["mask"], dev)
    cond = T_(data["cond"], dev)
    target = inp * mask
    beta = meta["beta"]
    parts = torch.zeros(3 + max(sch.n_cats, 1), device=dev)
    eng.fp.zero_grad()
    zz, mu, lv = eng.encoder(inp, T_(z["eps"], dev), train=True, kld_sum=parts[2 + max(sch.n_cats, 1):])
    heads = eng.decoder(zz, mask, cond, target=target, scales=eng.loss_scales(rows), loss_parts=parts, train=True)
    eng.backward(beta, rows)
    torch.cuda.synchronize()
    gref = fu.sub(z, "g")
    grads = {k: eng.fp.g(k).cpu().numpy() for k in eng.fp.offsets if eng.fp.offsets[k] < eng.fp.n_active}
    assert set(gref) == set(grads)
    out = eng.split_heads(heads, B, T)
    p = parts.tolist()
    total, pr = eng.total_loss(p[:2 + sch.n_cats], p[2 + max(sch.n_cats, 1)], beta, rows)
    if dtype == "fp32":
        ref = dict(mu=z["mu"], lv=z["lv"], z=z["z"], loss=float(z["loss"]), kld=float(z["kld"]))
        if sch.n_bins:
            ref["bins"] = z["out/bins"]
        ref["cats"] = [z[f"out/cats{j}"] for j in range(sch.n_cats)]
        if sch.n_nums:
            ref["nums"] = z["out/nums"]
        ot, gt, gr = TOL, TOL, gref
    else:
        # the oracle with operands rounded to bf16 at the same points as the kernels
        with oops.bf16_operands(storage=True):
            ototal, ores, ograds = ovae.loss_and_grads(sd, data["response"] + data["cond"], data["mask"], data["cond"], z["eps"],
                                                       np.float32(beta), sch)
        ref = dict(mu=ores["mu"], lv=ores["lv"], z=ores["z"], loss=ototal, kld=ores["parts"]["kld"],
                   cats=ores["out"].get("cats", []))
        for k in ("bins", "nums"):
            if k in ores["out"]:
                ref[k] = ores["out"][k]
        ot, gt, gr = BF16_EMU_TOL, BF16_EMU_GRAD_TOL, {k: v.reshape(grads[k].shape) for k, v in ograds.items()}
        # and the stated looser tolerance against the fp32 reference
        assert fu.rel_err(zz.cpu().numpy(), z["z"]) < BF16_OUT_TOL
        assert abs(total - float(z["loss"])) < BF16_OUT_TOL * abs(float(z["loss"]))
    assert fu.rel_err(mu.cpu().numpy(), ref["mu"]) < ot
    assert fu.rel_err(lv.cpu().numpy(), ref["lv"]) < ot
    assert fu.rel_err(zz.cpu().numpy(), ref["z"]) < ot
    if sch.n_bins:
        assert fu.rel_err(out["bins"].cpu().numpy(), ref["bins"]) < ot
    for j in range(sch.n_cats):
        assert fu.rel_err(out["cats"][j].cpu().numpy(), ref["cats"][j]) < ot
    if sch.n_nums:
        assert fu.rel_err(out["nums"].cpu().numpy(), ref["nums"]) < ot
    assert abs(total - ref["loss"]) < ot * abs(ref["loss"])
    assert abs(pr["kld"] - ref["kld"]) < ot * abs(ref["kld"])
    errs = {k: grad_err(grads[k], gr[k], dtype) for k in gref}
    print("worst grad errors:", sorted(errs.items(), key=lambda kv: -kv[1])[:4])
    assert max(errs.values()) < gt, errs


def test_vae_module_surface(dev):
    """ae.encoder / ae.decoder / ae(x, mask, cond) as the reference's drivers call them."""
    z, meta = fu.load("vae_mixed_small")
    sch = _schema(meta)
    ae = _make_vae(meta, sch, dev, "fp32", fu.sub(z, "w"))
    data = fu.make_data(meta["schema"], N=meta["B"], T=meta["T"], K=meta["L"], seed=meta["seed"], mask_kind=meta["mask_kind"])
    inp = torch.from_numpy(data["response"] + data["cond"])
    out, emb, mu, lv = ae(inp, torch.from_numpy(data["mask"]), torch.from_numpy(data["cond"]), eps=torch.from_numpy(z["eps"]))
    assert set(out.keys()) == {"bins", "cats", "nums"} and len(out["cats"]) == 2
    assert fu.rel_err(emb.cpu().numpy(), z["z"]) < TOL and fu.rel_err(out["nums"].cpu().numpy(), z["out/nums"]) < TOL
    dec = ae.decoder(torch.from_numpy(z["z"]), torch.from_numpy(data["mask"]), torch.from_numpy(data["cond"]))
    assert fu.rel_err(dec["bins"].cpu().numpy(), z["out/bins"]) < TOL
    # decoder -> convert_to_tensor, as every driver does after sampling (Run_Imputation.ipynb:171-172): the fused
    # path equals the two-step path, and both equal the oracle's conversion of the reference's decoder outputs
    from oracle import convert as oconv
    from timeautodiff_amd import process_edited as pce

    class P:
        def datatype_info(self):
            return dict(meta["schema"])
    args = (torch.from_numpy(z["z"]), torch.from_numpy(data["mask"]), torch.from_numpy(data["cond"]))
    fused = ae.decode_to_tensor(*args)
    two_step = pce.convert_to_tensor(P(), ae.decoder(*args), 1, meta["B"], meta["T"])
    assert torch.equal(fused, two_step)
    ref_gen = dict(bins=z["out/bins"], cats=[z[f"out/cats{j}"] for j in range(sch.n_cats)], nums=z["out/nums"])
    ref_tab = oconv.convert_to_tensor(ref_gen, sch.n_bins, sch.n_cats, sch.n_nums, sch.cards)
    got = fused.cpu().numpy()
    nd = sch.n_bins + sch.n_cats
    assert np.array_equal(got[..., :nd], ref_tab[..., :nd])                   # codes: exact
    assert fu.rel_err(got[..., nd:], ref_tab[..., nd:]) < TOL                 # numericals: decoder tolerance
    # auto_loss on decoded outputs reproduces the reference's (disc, num)
    from timeautodiff_amd import VAE
    target = (inp * torch.from_numpy(data["mask"])).to(dev)
    disc, num = VAE.auto_loss(target, out, sch.n_bins, sch.n_nums, sch.n_cats, sch.cards)
    assert abs(float(disc) - float(z["disc"])) < TOL * float(z["disc"])
    assert abs(float(num) - float(z["num"])) < TOL * max(float(z["num"]), 1e-9)


# ------------------------------------------------------------------ loops
def test_sampler_trajectory_vs_reference_golden(dev):
    from timeautodiff_amd import DIFF
    z, meta = fu.load("sampler_mixed_small")
    sch = _schema(meta)
    m = _make_denoiser(meta, sch, dev, "fp32", fu.sub(z, "w"))
    B, T, K = meta["B"], meta["T"], meta["K"]
    data = fu.make_data(meta["schema"], N=B, T=T, K=K, seed=meta["seed"], mask_kind="ones")
    torch.manual_seed(meta["seed"] + 7)
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(B, 1, 1)
    keep = {99: None, 50: None, 1: None}
    x = DIFF.sample(tgrid, B, T, K, 100, m, torch.from_numpy(data["time_info"]), torch.from_numpy(data["cond"]), None, keep=keep)
    assert fu.rel_err(keep[99].cpu().numpy(), z["x_after_99"]) < TOL
    assert fu.rel_err(keep[50].cpu().numpy(), z["x_after_50"]) < TOL
    assert fu.rel_err(keep[1].cpu().numpy(), z["x_after_1"]) < TOL
    assert fu.rel_err(x.cpu().numpy(), z["x_final"]) < TOL


def test_train_diffusion_loss_curve_vs_reference_golden(dev):
    from timeautodiff_amd import DIFF
    from test_host_logic import StubParser
    z, meta = fu.load("diffloop_numeric")
    N, T, K, H, steps = meta["N"], meta["T"], meta["K"], meta["H"], meta["steps"]
    data = fu.make_data(meta["schema"], N=N, T=T, K=K, seed=meta["seed"], mask_kind="ones")
    dd = dict(parser=StubParser(meta["schema"]), cond_train=torch.from_numpy(data["cond"]),
              time_info_train=torch.from_numpy(data["time_info"]), target_mask_train=torch.from_numpy(data["mask"]))
    random.seed(meta["seed"])
    torch.manual_seed(meta["seed"])
    losses = []
    model = DIFF.train_diffusion(torch.from_numpy(data["latent"]), None, dd, H, 2, 100, steps, losses=losses)
    ref = z["losses"]
    assert len(losses) == steps
    assert np.max(np.abs(np.array(losses) - ref) / np.abs(ref)) < 1e-3          # fixed-seed loss-curve window
    ws = fu.sub(z, "ws")
    sd = model.state_dict()
    for k in ws:
        if not k.startswith("Emb_hidden"):
            assert abs(fu.stats(sd[k].cpu().numpy())[1] - ws[k][1]) < 1e-3 * max(ws[k][1], 1e-6), k
    # and the sampler on the trained model reproduces the reference's sample
    Bs = meta["Bs"]
    torch.manual_seed(meta["seed"] + 7)
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(Bs, 1, 1)
    x = DIFF.sample(tgrid, Bs, T, K, 100, model, torch.from_numpy(data["time_info"][:Bs]), torch.from_numpy(data["cond"][:Bs]), None)
    assert fu.rel_err(x.cpu().numpy(), z["sample"]) < 5e-3


@pytest.mark.parametrize("name,fn", [("vaeloop_mixed_val", "train_autoencoder"), ("vaeloop_mixed_train", "train_autoencoder_best_train")])
def test_train_autoencoder_loss_curve_vs_reference_golden(dev, name, fn):
    from timeautodiff_amd import VAE
    from test_host_logic import StubParser
    z, meta = fu.load(name)
    N, Nv, T, L, B, steps = meta["N"], meta["Nv"], meta["T"], meta["L"], meta["B"], meta["steps"]
    tr = fu.make_data(meta["schema"], N=N, T=T, K=L, seed=meta["seed"], mask_kind="bern")
    va = fu.make_data(meta["schema"], N=Nv, T=T, K=L, seed=meta["seed"] + 100, mask_kind="bern")
    dd = dict(parser=StubParser(meta["schema"]),
              response_train=torch.from_numpy(tr["response"]), cond_train=torch.from_numpy(tr["cond"]),
              target_mask_train=torch.from_numpy(tr["mask"]).int(),             # int32 masks as dataprovider_pypots emits
              response_val=torch.from_numpy(va["response"]), cond_val=torch.from_numpy(va["cond"]),
              target_mask_val=torch.from_numpy(va["mask"]).int())
    random.seed(meta["seed"])
    torch.manual_seed(meta["seed"])
    ae, latent, output, hist = getattr(VAE, fn)(None, 1, meta["H"], 2, meta["lr"], meta["wd"], steps, B, 1, meta["min_beta"],
                                                meta["max_beta"], meta["E"], 8, L, "cuda", dd, noise_source="reference")
    assert np.max(np.abs(np.array(hist["train"]) - z["train"]) / np.abs(z["train"])) < 1e-3
    assert np.max(np.abs(np.array(hist["val"]) - z["val"]) / np.abs(z["val"])) < 1e-3
    assert fu.rel_err(latent.cpu().numpy(), z["latent"]) < 2e-3
    assert set(output.keys()) == {"bins", "cats", "nums"} and not latent.requires_grad
    ws = fu.sub(z, "ws")
    sd = ae.state_dict()
    for k in ws:
        assert abs(fu.stats(sd[k].cpu().numpy())[1] - ws[k][1]) < 1e-3 * max(ws[k][1], 1e-6), k


# ------------------------------------------------------------------ properties at benchmark size
FULLSIZE = {
    # name: (schema, denoiser H, T, windows per launch, mask of the VAE run)
    "energy_T24_H200": (dict(n_bins=0, n_cats=1, n_nums=27, cards=[8]), 200, 24, 4096, "ones"),
    "metro_T48_H512": (dict(n_bins=0, n_cats=3, n_nums=5, cards=[12, 1, 11]), 512, 48, 4096, "bern"),
    "airq_T48_H512": (dict(n_bins=0, n_cats=0, n_nums=13, cards=[]), 512, 48, 4096, "bern"),
}


@pytest.mark.parametrize("name", list(FULLSIZE))
def test_denoiser_fullsize_properties(dev, name):
    """The benchmark batch (4096 windows, bf16) at the energy shape (T=24, H=200) and at the Metro / AirQuality shapes
    (T=48, H=512: 32-row workgroups with two chunks of hidden units per wave): rows are
    independent (a sub-batch reproduces its rows of the big batch) and the gradient is
    additive over batch halves under the same global denominator."""
    from timeautodiff_amd import DIFF
    s, H, T, B, _ = FULLSIZE[name]
    K = 6
    F = s["n_bins"] + s["n_cats"] + s["n_nums"]
    data = fu.make_data(s, N=B, T=T, K=K, seed=1, mask_kind="ones")
    data["cond"] = data["data"]
    torch.manual_seed(5)
    m = DIFF.BiRNN_score(K, F, H, 2, 100, 8, 128, s["n_bins"], s["n_cats"], s["n_nums"], list(s["cards"]))
    m.set_compute_dtype("bf16")
    eng = m.engine(dev)
    g = torch.Generator().manual_seed(6)
    t = torch.rand(B, T, 1, generator=g).sort(1)[0].to(dev)
    i_idx = torch.randint(0, 100, (B,), generator=g).to(dev, torch.int32)
    noise = torch.randn(B, T, K, generator=g).to(dev)
    x0, ti, cond = T_(data["latent"], dev), T_(data["time_info"], dev), T_(data["cond"], dev)
    al = DIFF.alphas.to(dev)
    n_glob = B * T * K

    def run(lo, hi):
        eng.fp.zero_grad()
        ls = torch.zeros(1, device=dev)
        pred = eng.loss_and_grads(x0[lo:hi].contiguous(), t[lo:hi].contiguous(), i_idx[lo:hi].contiguous(),
                                  noise[lo:hi].contiguous(), ti[lo:hi].contiguous(), cond[lo:hi].contiguous(), al, ls,
                                  global_numel=n_glob)
        return pred.clone(), ls.item(), eng.fp.gflat[:eng.fp.n_active].clone()
    p_all, l_all, g_all = run(0, B)
    p_a, l_a, g_a = run(0, B // 2)
    p_b, l_b, g_b = run(B // 2, B)
    assert torch.isfinite(p_all).all() and torch.isfinite(g_all).all()
    assert torch.equal(p_all[:B // 2], p_a) and torch.equal(p_all[B // 2:], p_b)      # row independence, bit exact
    assert abs(l_all - (l_a + l_b)) < 1e-5 * l_all
    gs = g_a + g_b
    assert float((g_all - gs).abs().max()) < 1e-3 * float(g_all.abs().max())           # atomics reorder sums only


@pytest.mark.parametrize("name", list(FULLSIZE))
def test_vae_fullsize_properties(dev, name):
    """DeapStack at the reference's hidden size 512 (VAE.py:119-120) on 4096 windows, bf16 -- the launch shapes the bench
    times (32-row workgroups, two chunks of hidden units per wave, the GRU's backward gate by gate), with the masked mixed
    loss (Metro / AirQuality: Bernoulli(0.1) point mask and a cond): the latents and head outputs of a window do not
    depend on the batch it is computed in (bit exact), the loss parts add up over batch halves, and so do the gradients
    under the same global denominators."""
    s, _, T, B, mk = FULLSIZE[name]
    sch = Schema(s["n_bins"], s["n_cats"], s["n_nums"], list(s["cards"]))
    L, beta = 6, 0.05
    ae = _make_vae(dict(H=512, E=128, L=L, seed=7), sch, dev, "bf16")
    eng = ae.engine(dev)
    data = fu.make_data(s, N=B, T=T, K=L, seed=8, mask_kind=mk)
    inp = T_(data["response"] + data["cond"], dev)
    mask, cond = T_(data["mask"], dev), T_(data["cond"], dev)
    target = inp * mask
    eps = torch.randn(B, T, L, generator=torch.Generator().manual_seed(9)).to(dev)
    nparts = 3 + max(sch.n_cats, 1)
    rows_glob = B * T

    def run(lo, hi):
        eng.fp.zero_grad()
        parts = torch.zeros(nparts, device=dev)
        c = lambda a: a[lo:hi].contiguous()     # noqa: E731
        zz, mu, lv = eng.encoder(c(inp), c(eps), train=True, kld_sum=parts[2 + max(sch.n_cats, 1):])
        heads = eng.decoder(zz, c(mask), c(cond), target=c(target), scales=eng.loss_scales(rows_glob), loss_parts=parts, train=True)
        keep = (zz.clone(), mu.clone(), heads.clone())
        eng.backward(beta, rows_glob)
        torch.cuda.synchronize()
        return keep, parts.double().cpu(), eng.fp.gflat[:eng.fp.n_active].clone()
    (z_all, mu_all, h_all), p_all, g_all = run(0, B)
    (z_a, mu_a, h_a), p_a, g_a = run(0, B // 2)
    (z_b, mu_b, h_b), p_b, g_b = run(B // 2, B)
    assert torch.isfinite(h_all).all() and torch.isfinite(g_all).all() and float(g_all.abs().max()) > 0
    hr = (B // 2) * T
    assert torch.equal(z_all[:B // 2], z_a) and torch.equal(z_all[B // 2:], z_b)       # row independence, bit exact
    assert torch.equal(mu_all[:B // 2], mu_a) and torch.equal(mu_all[B // 2:], mu_b)
    assert torch.equal(h_all[:hr], h_a) and torch.equal(h_all[hr:], h_b)
    assert float((p_all - (p_a + p_b)).abs().max()) < 1e-4 * float(p_all.abs().max())
    assert float((g_all - (g_a + g_b)).abs().max()) < 1e-3 * float(g_all.abs().max())   # atomics reorder sums only


def test_vae_encoder_beyond_the_cluster_forms_address_range(dev):
    """2700 windows of 96 steps at hidden 512: the stash of the encoder GRUs exceeds the 2 GiB that the eight-workgroup
    sequence form addresses, so the first layer may not take the launch that multiplies its input itself (it has no
    other form to fall back to inside the library): the engine forms the pre-activations and the plain kernels run.
    The latents of the first windows equal those of a small batch of the same windows to the bf16 tolerance."""
    s = dict(n_bins=0, n_cats=0, n_nums=7, cards=[])
    sch = Schema(0, 0, 7, [])
    B, T, L = 2700, 96, 6
    ae = _make_vae(dict(H=512, E=128, L=L, seed=3), sch, dev, "bf16")
    eng = ae.engine(dev)
    data = fu.make_data(s, N=64, T=T, K=L, seed=4, mask_kind="ones")
    x64 = T_(data["response"] + data["cond"], dev)
    inp = x64.repeat(B // 64 + 1, 1, 1)[:B].contiguous()
    eps = torch.zeros(B, T, L, device=dev)
    z_big, mu_big, _ = eng.encoder(inp, eps, train=True)
    from timeautodiff_amd import ops
    assert not ops._lib.load().tad_last_kernel().decode().startswith("rnn_cluster"), "checked below through the values"
    mu_big = mu_big[:64].clone()
    z_small, mu_small, _ = eng.encoder(x64, eps[:64].contiguous(), train=True)
    torch.cuda.synchronize()
    assert torch.isfinite(mu_big).all()
    assert fu.rel_err(mu_big.cpu().numpy(), mu_small.cpu().numpy()) < BF16_EMU_TOL


# ------------------------------------------------------------------ bf16 loss-curve windows
BF16_CURVE_TOL = 2e-2    # bf16 training loop vs the bf16-emulating oracle's loop, per-iteration loss, relative
BF16_CURVE_REF_TOL = 5e-2   # bf16 training loop vs the fp32 reference's recorded curve, per-iteration loss, relative


def test_train_diffusion_bf16_loss_curve_window(dev):
    """The mode the bench times (bf16 operands, bf16-stored products) through 20 optimizer steps on the reference's
    recorded draws (DIFF.py:305-328): the loss of every iteration against the oracle's loop that rounds at the kernels'
    rounding points (2e-2: one step agrees to 5e-3, Adam's first steps move every weight by lr whatever the size of
    its gradient, so rounding-level gradient differences grow for a few steps before the curve contracts them), and
    against the fp32 reference's recorded curve at the stated looser bf16 tolerance of 5e-2."""
    from oracle import loops as oloops
    from timeautodiff_amd import DIFF
    from test_oracle_golden import replay_diffusion_draws
    z, meta = fu.load("diffloop_numeric")
    sch = _schema(meta)
    N, T, K, H, steps = meta["N"], meta["T"], meta["K"], meta["H"], meta["steps"]
    data = fu.make_data(meta["schema"], N=N, T=T, K=K, seed=meta["seed"], mask_kind="ones")
    random.seed(meta["seed"])
    torch.manual_seed(meta["seed"])
    m = DIFF.BiRNN_score(K, sch.n_bins + sch.n_cats + sch.n_nums, H, 2, 100, 8, 128, sch.n_bins, sch.n_cats, sch.n_nums, sch.cards)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    draws = list(replay_diffusion_draws(N, T, K, steps))
    tr = DIFF.DiffusionTrainer(m, torch.from_numpy(data["latent"]), torch.from_numpy(data["time_info"]), torch.from_numpy(data["cond"]),
                               dtype="bf16", dev=dev)
    assert tr.eng.mm16                                   # the bf16-stored-operand path, as in the bench
    hist = torch.zeros(steps, device=dev)
    for k, d in enumerate(draws):
        tr.step(draw=dict(idx=torch.tensor(d["idx"]), t=torch.from_numpy(d["t"]), i=torch.from_numpy(d["i"]),
                          noise=torch.from_numpy(d["noise"])), loss_out=hist[k:k + 1])
    got = hist.cpu().numpy().astype(np.float64)
    _, alphas = oops.schedule(100)
    with oops.bf16_operands(storage=True):
        ref16 = np.array(oloops.train_diffusion_steps(sd, data["latent"], data["time_info"], data["cond"], sch, draws, alphas))
    e16 = np.max(np.abs(got - ref16) / np.abs(ref16))
    e32 = np.max(np.abs(got - z["losses"]) / np.abs(z["losses"]))
    print("bf16 loss curve: vs emulating oracle %.3g, vs fp32 reference %.3g" % (e16, e32))
    assert e16 < BF16_CURVE_TOL and e32 < BF16_CURVE_REF_TOL
    # the weights after the window: norms against the emulating oracle's
    sd_dev = m.state_dict()
    for k, v in sd.items():
        if not k.startswith("Emb_hidden"):
            a, b = fu.stats(sd_dev[k].cpu().numpy())[1], fu.stats(v)[1]
            assert abs(a - b) < 1e-2 * max(b, 1e-6), k


@pytest.mark.parametrize("name,select", [("vaeloop_mixed_val", "val"), ("vaeloop_mixed_train", "train")])
def test_train_autoencoder_bf16_loss_curve_window(dev, name, select):
    """VAE.train_autoencoder / _best_train in bf16 through the reference's 30 recorded iterations (VAE.py:320-388,
    477-547; train loss, validation loss, beta annealing, best-state rule): both curves against the bf16-emulating
    oracle's loop at 2e-2 and against the fp32 reference's recorded curves at 5e-2."""
    from oracle import loops as oloops
    from timeautodiff_amd import VAE
    from test_host_logic import StubParser
    z, meta = fu.load(name)
    sch = _schema(meta)
    N, Nv, T, L, B, steps = meta["N"], meta["Nv"], meta["T"], meta["L"], meta["B"], meta["steps"]
    tr = fu.make_data(meta["schema"], N=N, T=T, K=L, seed=meta["seed"], mask_kind="bern")
    va = fu.make_data(meta["schema"], N=Nv, T=T, K=L, seed=meta["seed"] + 100, mask_kind="bern")
    dd = dict(parser=StubParser(meta["schema"]),
              response_train=torch.from_numpy(tr["response"]), cond_train=torch.from_numpy(tr["cond"]),
              target_mask_train=torch.from_numpy(tr["mask"]).int(),
              response_val=torch.from_numpy(va["response"]), cond_val=torch.from_numpy(va["cond"]),
              target_mask_val=torch.from_numpy(va["mask"]).int())
    fn = "train_autoencoder" if select == "val" else "train_autoencoder_best_train"
    random.seed(meta["seed"])
    torch.manual_seed(meta["seed"])
    ae, latent, output, hist = getattr(VAE, fn)(None, 1, meta["H"], 2, meta["lr"], meta["wd"], steps, B, 1, meta["min_beta"],
                                                meta["max_beta"], meta["E"], 8, L, "cuda", dd, noise_source="reference", dtype="bf16")
    # the oracle's loop on the same draws (same seeds, same order: parameter init, then per iteration indices, eps, eps_val)
    random.seed(meta["seed"])
    torch.manual_seed(meta["seed"])
    F = sch.n_bins + sch.n_cats + sch.n_nums
    sd = {k: v.numpy().copy() for k, v in
          VAE.init_vae_state(sch.n_bins, sch.n_cats, sch.n_nums, sch.cards, F, meta["H"], 2, meta["E"], 8, L).items()}

    def draws():
        for _ in range(steps):
            idx = random.sample(list(range(N)), B)
            yield dict(idx=idx, eps_train=torch.randn(B, T, L).numpy(), eps_val=torch.randn(Nv, T, L).numpy())
    data = dict(response_train=tr["response"], cond_train=tr["cond"], target_mask_train=tr["mask"],
                response_val=va["response"], cond_val=va["cond"], target_mask_val=va["mask"])
    with oops.bf16_operands(storage=True):
        ohist, _, _ = oloops.train_autoencoder_steps(sd, data, sch, draws(), meta["lr"], meta["wd"], meta["min_beta"], meta["max_beta"],
                                                     select=select)
    for k in ("train", "val"):
        got = np.array(hist[k], np.float64)
        e16 = np.max(np.abs(got - np.array(ohist[k])) / np.abs(np.array(ohist[k])))
        e32 = np.max(np.abs(got - z[k]) / np.abs(z[k]))
        print("bf16 VAE %s curve: vs emulating oracle %.3g, vs fp32 reference %.3g" % (k, e16, e32))
        assert e16 < BF16_CURVE_TOL and e32 < BF16_CURVE_REF_TOL, k
    assert not latent.requires_grad and torch.isfinite(latent).all()


def test_sampler_fullsize_sharding_invariance(dev):
    """1k-sequence generation: sampling a batch in two halves equals sampling it whole."""
    from timeautodiff_amd import DIFF
    s = dict(n_bins=0, n_cats=1, n_nums=27, cards=[8])
    B, T, K, H = 1024, 24, 6, 200
    data = fu.make_data(s, N=B, T=T, K=K, seed=2, mask_kind="ones")
    torch.manual_seed(5)
    m = DIFF.BiRNN_score(K, 28, H, 2, 100, 8, 128, 0, 1, 27, [8])
    m.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(9)
    noise = dict(x_T=torch.randn(B, T, K, generator=g), z=torch.randn(100, B, T, K, generator=g))
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(B, 1, 1)
    ti, cond = torch.from_numpy(data["time_info"]), torch.from_numpy(data["data"])
    x = DIFF.sample(tgrid, B, T, K, 100, m, ti, cond, None, noise=noise)
    h = B // 2
    xa = DIFF.sample(tgrid[:h], h, T, K, 100, m, ti[:h], cond[:h], None, noise=dict(x_T=noise["x_T"][:h], z=noise["z"][:, :h]))
    assert torch.isfinite(x).all()
    assert torch.equal(x[:h], xa)
    # the batch walks the loop in concurrent chunks on streams of their own: no value depends on how many
    x1 = DIFF.sample(tgrid, B, T, K, 100, m, ti, cond, None, noise=noise, streams=1)
    x3 = DIFF.sample(tgrid, B, T, K, 100, m, ti, cond, None, noise=noise, streams=3)
    assert torch.equal(x, x1) and torch.equal(x, x3)
    # ... nor on replaying the loop as one captured HIP graph (second call: replay of the cached graph on new noise)
    xg = DIFF.sample(tgrid, B, T, K, 100, m, ti, cond, None, noise=noise, streams=2, graph=True)
    assert torch.equal(x, xg)
    noise2 = dict(x_T=noise["x_T"].flip(0), z=noise["z"].flip(1))
    xg2 = DIFF.sample(tgrid, B, T, K, 100, m, ti.flip(0), cond.flip(0), None, noise=noise2, streams=2, graph=True)
    assert torch.equal(xg2, x.flip(0))


def test_sampler_device_noise_is_counter_based(dev):
    """noise_source="device": every step's noise is drawn on the device when the step runs, as a function of (seed, step,
    global row) -- the result equals the loop on the explicitly staged tensors of the same values, whatever the chunking,
    with or without graph replay, and for a slice of the batch (the rows a second rank would own)."""
    from timeautodiff_amd import DIFF, ops
    s = dict(n_bins=0, n_cats=1, n_nums=27, cards=[8])
    B, T, K, H, S = 600, 24, 6, 200, 100
    data = fu.make_data(s, N=B, T=T, K=K, seed=3, mask_kind="ones")
    torch.manual_seed(11)
    m = DIFF.BiRNN_score(K, 28, H, 2, S, 8, 128, 0, 1, 27, [8])
    m.set_compute_dtype("bf16")
    tgrid = torch.linspace(0, 1, T).view(1, -1, 1).repeat(B, 1, 1)
    ti, cond = torch.from_numpy(data["time_info"]), torch.from_numpy(data["data"])
    seed = torch.initial_seed()
    x_T = torch.empty(B, T, K, device=dev)
    ops.sampler_noise(seed, S, 0, B * T, K, x_T)
    z = torch.empty(S, B, T, K, device=dev)
    for st in range(S):
        ops.sampler_noise(seed, st, 0, B * T, K, z[st])
    ref = DIFF.sample(tgrid, B, T, K, S, m, ti, cond, None, noise=dict(x_T=x_T, z=z), streams=1)
    assert torch.isfinite(ref).all()
    for kw in (dict(streams=1), dict(streams=2), dict(streams=2, graph=True), dict(streams=2, graph=True)):
        x = DIFF.sample(tgrid, B, T, K, S, m, ti, cond, None, noise_source="device", **kw)
        assert torch.equal(x, ref), kw
    # the second half of the batch alone, as the rows of a second rank: same values as in the whole batch
    lo = 300
    xT2 = torch.empty(B - lo, T, K, device=dev)
    ops.sampler_noise(seed, S, lo * T, (B - lo) * T, K, xT2)
    assert torch.equal(xT2, x_T[lo:])
    # fresh noise under the same seed: another call number; another seed, another sample
    x1 = DIFF.sample(tgrid, B, T, K, S, m, ti, cond, None, noise_source="device", noise_call=1)
    assert torch.isfinite(x1).all() and not torch.equal(x1, ref)
    assert torch.equal(DIFF.sample(tgrid, B, T, K, S, m, ti, cond, None, noise_source="device", noise_call=1, streams=1), x1)
    torch.manual_seed(12)
    assert not torch.equal(DIFF.sample(tgrid, B, T, K, S, m, ti, cond, None, noise_source="device"), ref)


def test_train_diffusion_on_windowed_tables_equals_stacked_tensors(dev):
    """time_info / cond handed over as WindowedTable views of their [rows, F] tables (windows cut per minibatch
    on the device) train like the reference's stacked [N, T, F] tensors: same loss curve and weights."""
    from oracle import windows as owin
    from timeautodiff_amd import DIFF
    from timeautodiff_amd.dataprovider_pypots import WindowedTable
    from test_host_logic import StubParser
    sch = fu.SCHEMAS["mixed"]
    T, K, H, N, steps = 8, 4, 16, 130, 6
    rows = N + T - 1
    flat = fu.make_data(sch, N=rows, T=1, K=K, seed=77, mask_kind="ones")
    cond_tab, time_tab = flat["cond"][:, 0, :], flat["time_info"][:, 0, :]
    latent = np.random.default_rng(78).standard_normal((N, T, K)).astype(np.float32)
    runs = []
    for windowed in (False, True):
        if windowed:
            cond, ti = WindowedTable(torch.from_numpy(cond_tab), T, 1, device=dev), WindowedTable(torch.from_numpy(time_tab), T, 1, device=dev)
        else:
            cond, ti = torch.from_numpy(owin.sliding_window(cond_tab, T, 1)), torch.from_numpy(owin.sliding_window(time_tab, T, 1))
        assert tuple(cond.shape) == (N, T, cond_tab.shape[1])
        dd = dict(parser=StubParser(sch), cond_train=cond, time_info_train=ti)
        random.seed(5)
        torch.manual_seed(5)
        losses = []
        model = DIFF.train_diffusion(torch.from_numpy(latent), None, dd, H, 2, 100, steps, losses=losses)
        runs.append((losses, {k: v.cpu().numpy().copy() for k, v in model.state_dict().items()}))
    # same windows, same draws: the runs differ only by the summation order of float atomics (the loss sum,
    # the split reductions of the weight gradients), i.e. by rounding: 1e-6 on the first loss, 1e-5 on the curve
    assert abs(runs[0][0][0] - runs[1][0][0]) < 1e-6 * abs(runs[0][0][0])
    assert np.max(np.abs(np.array(runs[0][0]) - np.array(runs[1][0])) / np.abs(np.array(runs[0][0]))) < 1e-5
    for k in runs[0][1]:
        assert fu.rel_err(runs[1][1][k], runs[0][1][k]) < 1e-4, k


# ---------------------------------------------------------------------------- model dimensions of the other workloads
BASELINE_SHAPES = {
    # name: (schema, denoiser H, T, batch, VAE hidden, mask)
    "metro_T48_H512": (dict(n_bins=0, n_cats=3, n_nums=5, cards=[12, 1, 11]), 512, 48, 3, 512, "bern"),       # Metro_Traffic
    "etth1_T96_H200": (dict(n_bins=0, n_cats=0, n_nums=7, cards=[]), 200, 96, 3, 512, "ones"),                # ETTh1 long windows
    "airq_T48_H512": (dict(n_bins=0, n_cats=0, n_nums=13, cards=[]), 512, 48, 2, 512, "bern"),                # AirQuality, 10 % point mask
    "energy_T24_H200": (dict(n_bins=0, n_cats=1, n_nums=27, cards=[8]), 200, 24, 4, 512, "ones"),             # energy_data (the bench shape)
}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(BASELINE_SHAPES))
def test_denoiser_at_baseline_model_dimensions_vs_oracle(dev, name, dtype):
    """BiRNN_score at the hidden size, window length and column schema of the Metro / ETTh1 / AirQuality / energy
    workloads (H = 512 takes the two-chunks-per-wave and gate-by-gate forms of the sequence kernels), small batch:
    output, loss and every parameter gradient against the oracle on the same seeded inputs -- fp32 1e-4; bf16 against
    the oracle that rounds operands and stored tensors to bf16 at the kernels' rounding points, 5e-3 / 2e-2."""
    sd_, H, T, B, _, _ = BASELINE_SHAPES[name]
    sch = Schema(sd_["n_bins"], sd_["n_cats"], sd_["n_nums"], list(sd_["cards"]))
    K, E = 6, 128
    meta = dict(K=K, H=H, E=E, seed=91)
    m = _make_denoiser(meta, sch, dev, dtype)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    data = fu.make_data(sd_, N=B, T=T, K=K, seed=92, mask_kind="ones")
    rng = np.random.default_rng(93)
    t = np.sort(rng.random((B, T, 1), dtype=np.float32), 1)
    i_idx = rng.integers(0, 100, B)
    noise = rng.standard_normal((B, T, K)).astype(np.float32)
    pred, loss, grads = _denoiser_step(m, dev, data, t, i_idx, noise)
    _, alphas = oops.schedule(100)
    if dtype == "fp32":
        oloss, opred, ograds = oden.loss_and_grads(sd, data["latent"], t, i_idx, noise, data["time_info"], data["cond"], sch, alphas)
        ot, gt = TOL, TOL
    else:
        with oops.bf16_operands(storage=True):
            oloss, opred, ograds = oden.loss_and_grads(sd, data["latent"], t, i_idx, noise, data["time_info"], data["cond"], sch, alphas)
        ot, gt = BF16_EMU_TOL, BF16_EMU_GRAD_TOL
    assert fu.rel_err(pred, opred) < ot
    assert abs(loss - oloss) < ot * abs(oloss)
    errs = {k: grad_err(grads[k], v.reshape(grads[k].shape), dtype) for k, v in ograds.items()}
    assert max(errs.values()) < gt, sorted(errs.items(), key=lambda kv: -kv[1])[:4]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["metro_T48_H512", "airq_T48_H512", "energy_T24_H200"])
def test_vae_at_baseline_model_dimensions_vs_oracle(dev, name, dtype):
    """DeapStack at hidden 512 with the Metro (cards [12, 1, 11], T = 48), AirQuality (13 numericals, T = 48; both with a
    Bernoulli(0.1) int32 point mask) and energy (1 categorical of 8 + 27 numericals, T = 24) schemas: latents, heads,
    the masked mixed loss and every gradient against the oracle -- fp32 1e-4; bf16 against the bf16-emulating oracle."""
    sd_, _, T, B, Hv, mk = BASELINE_SHAPES[name]
    # (Metro: 4 windows.  With 3 and these seeds one pre-activation of the decoder's first ReLU lies within a bf16 step of
    # zero: the fp32 summation order decides its mask, and with it 2 % of the embedding gradients, between device and oracle.)
    B = 4 if name == "metro_T48_H512" else B
    sch = Schema(sd_["n_bins"], sd_["n_cats"], sd_["n_nums"], list(sd_["cards"]))
    L, E, beta = 6, 128, 0.05
    meta = dict(H=Hv, E=E, L=L, seed=95)
    ae = _make_vae(meta, sch, dev, dtype)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in ae.state_dict().items()}
    data = fu.make_data(sd_, N=B, T=T, K=L, seed=96, mask_kind=mk)
    eps = np.random.default_rng(97).standard_normal((B, T, L)).astype(np.float32)
    rows = B * T
    eng = ae.engine(dev)
    inp = T_(data["response"] + data["cond"], dev)
    mask = T_(data["mask"], dev)
    cond = T_(data["cond"], dev)
    parts = torch.zeros(3 + max(sch.n_cats, 1), device=dev)
    eng.fp.zero_grad()
    zz, mu, lv = eng.encoder(inp, T_(eps, dev), train=True, kld_sum=parts[2 + max(sch.n_cats, 1):])
    eng.decoder(zz, mask, cond, target=inp * mask, scales=eng.loss_scales(rows), loss_parts=parts, train=True)
    eng.backward(beta, rows)
    torch.cuda.synchronize()
    p = parts.tolist()
    total, _ = eng.total_loss(p[:2 + sch.n_cats], p[2 + max(sch.n_cats, 1)], beta, rows)
    if dtype == "fp32":
        ototal, ores, ograds = ovae.loss_and_grads(sd, data["response"] + data["cond"], data["mask"], data["cond"], eps, np.float32(beta), sch)
        ot, gt = TOL, TOL
    else:
        with oops.bf16_operands(storage=True):
            ototal, ores, ograds = ovae.loss_and_grads(sd, data["response"] + data["cond"], data["mask"], data["cond"], eps,
                                                       np.float32(beta), sch)
        ot, gt = BF16_EMU_TOL, BF16_EMU_GRAD_TOL
    assert fu.rel_err(zz.cpu().numpy(), ores["z"]) < ot and fu.rel_err(mu.cpu().numpy(), ores["mu"]) < ot
    assert abs(total - ototal) < ot * abs(ototal)
    grads = {k: eng.fp.g(k).cpu().numpy() for k in eng.fp.offsets if eng.fp.offsets[k] < eng.fp.n_active}
    errs = {k: grad_err(grads[k], v.reshape(grads[k].shape), dtype) for k, v in ograds.items()}
    assert max(errs.values()) < gt, sorted(errs.items(), key=lambda kv: -kv[1])[:4]


def test_sampler_long_windows_vs_oracle(dev):
    """Reverse diffusion on 96-step windows (the ETTh1 workload's length), H = 200: the full 100-step trajectory with
    injected noise against the oracle's loop, incl. the double update of the last step; fp32, 1e-4."""
    from oracle import loops as oloops
    from timeautodiff_amd import DIFF
    sd_ = BASELINE_SHAPES["etth1_T96_H200"][0]
    sch = Schema(0, 0, 7, [])
    B, T, K, H = 2, 96, 6, 200
    m = _make_denoiser(dict(K=K, H=H, E=128, seed=98), sch, dev, "fp32")
    sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    data = fu.make_data(sd_, N=B, T=T, K=K, seed=99, mask_kind="ones")
    rng = np.random.default_rng(100)
    xT = rng.standard_normal((B, T, K)).astype(np.float32)
    zs = rng.standard_normal((100, B, T, K)).astype(np.float32)
    tgrid = np.tile(np.linspace(0, 1, T, dtype=np.float32).reshape(1, T, 1), (B, 1, 1))
    x = DIFF.sample(torch.from_numpy(tgrid), B, T, K, 100, m, torch.from_numpy(data["time_info"]), torch.from_numpy(data["cond"]), None,
                    noise=dict(x_T=torch.from_numpy(xT), z=torch.from_numpy(zs)))
    betas, alphas = oops.schedule(100)
    ref, _ = oloops.sample(sd, tgrid, data["time_info"], data["cond"], sch, xT, zs, betas, alphas)
    assert fu.rel_err(x.cpu().numpy(), ref) < TOL
