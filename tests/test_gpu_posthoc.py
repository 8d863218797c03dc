"""The post-hoc evaluation networks on the HIP kernels against the fixtures generated from the reference's
Evaluation/Metrics.py (Discriminator) and Evaluation/predictive_metrics.py (PosthocRNN): forward, loss, every
gradient (fp32, 1e-4 like the other model tests; 2e-4 on the 20-step loss curve, 2e-3 on the weights after it), the
graph-replayed training loop, and the two metric functions end to end."""
import numpy as np
import pytest
import torch

import fixture_util as fu

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _load_into(module, sd):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def test_discriminator_forward_loss_grads_and_curve():
    from timeautodiff_amd import Metrics
    z, meta = fu.load("posthoc_disc")
    dev = torch.device("cuda:0")
    nb, dim = meta["nb"], meta["dim"]
    disc = Metrics.Discriminator(dim, meta["hidden"], meta["layers"])
    sd = fu.sub(z, "w")
    assert list(disc.state_dict().keys()) == list(sd.keys())            # the reference's parameter names and order
    _load_into(disc, sd)
    real, fake = torch.from_numpy(z["real"]).to(dev), torch.from_numpy(z["fake"]).to(dev)
    logit, prob = disc(real[:nb])
    assert fu.rel_err(logit.cpu().numpy(), z["logit_real"]) < TOL
    assert fu.rel_err(prob.cpu().numpy(), 1 / (1 + np.exp(-z["logit_real"]))) < TOL
    eng = disc.engine(dev)
    xb = torch.cat([real[:nb], fake[:nb]], 0).contiguous()
    yb = torch.cat([torch.ones(nb, 1, device=dev), torch.zeros(nb, 1, device=dev)], 0)
    ones = torch.ones(2 * nb, 1, device=dev)
    # one step with lr = 0: loss and gradients without moving the weights
    eng.disc_step_body(xb, yb, ones, nb, False, lr=0.0)
    torch.cuda.synchronize()
    assert abs(float(eng.loss.item()) / nb - float(z["loss"])) < TOL * float(z["loss"])
    for k, g in fu.sub(z, "g").items():
        assert fu.rel_err(eng.fp.g(k).cpu().numpy(), g) < TOL, k
    # the 20-step loop on the recorded minibatches, replayed from the captured graph
    disc2 = Metrics.Discriminator(dim, meta["hidden"], meta["layers"])
    _load_into(disc2, sd)
    e2 = disc2.engine(dev)
    curve = []
    for s in range(meta["steps"]):
        xb[:nb].copy_(real[torch.from_numpy(z["idx_real"][s]).to(dev)])
        xb[nb:].copy_(fake[torch.from_numpy(z["idx_fake"][s]).to(dev)])
        e2.replay("disc", lambda cap, coef: e2.disc_step_body(xb, yb, ones, nb, cap, coef))
        curve.append(float(e2.loss.item()) / nb)
    assert np.max(np.abs(np.array(curve) - z["curve"]) / z["curve"]) < 2e-4
    for k, w in fu.sub(z, "w_end").items():
        assert fu.rel_err(e2.fp.p(k).cpu().numpy(), w) < 2e-3, k


def test_predictor_forward_loss_grads_and_curve():
    from timeautodiff_amd import Metrics
    z, meta = fu.load("posthoc_pred")
    dev = torch.device("cuda:0")
    nb, dim, col, T = meta["nb"], meta["dim"], meta["col"], meta["T"]
    others = [c for c in range(dim) if c != col]
    model = Metrics.PosthocRNN(dim, int(dim / 2))
    sd = fu.sub(z, "w")
    assert list(model.state_dict().keys()) == list(sd.keys())
    _load_into(model, sd)
    fake = torch.from_numpy(z["fake"]).to(dev)
    gx, gy = fake[:, :-1, others].contiguous(), fake[:, 1:, col].contiguous()
    p = model(gx[:nb])
    assert fu.rel_err(p.cpu().numpy(), z["pred"]) < TOL
    eng = model.engine(dev)
    eng.pred_step_body(gx[:nb].contiguous(), gy[:nb].reshape(-1, 1).contiguous(), False, lr=0.0)
    torch.cuda.synchronize()
    assert abs(float(eng.loss.item()) / (nb * (T - 1)) - float(z["loss"])) < TOL * float(z["loss"])
    for k, g in fu.sub(z, "g").items():
        assert fu.rel_err(eng.fp.g(k).cpu().numpy(), g) < TOL, k
    model2 = Metrics.PosthocRNN(dim, int(dim / 2))
    _load_into(model2, sd)
    e2 = model2.engine(dev)
    xb, yb = torch.zeros(nb, T - 1, dim - 1, device=dev), torch.zeros(nb, T - 1, device=dev)
    curve = []
    for s in range(meta["steps"]):
        idx = torch.from_numpy(z["idx"][s]).to(dev)
        xb.copy_(gx[idx])
        yb.copy_(gy[idx])
        e2.replay("pred", lambda cap, coef: e2.pred_step_body(xb, yb.view(-1, 1), cap, coef))
        curve.append(float(e2.loss.item()) / (nb * (T - 1)))
    assert np.max(np.abs(np.array(curve) - z["curve"]) / z["curve"]) < 2e-4


def test_metric_functions_end_to_end():
    """discriminative_score_metrics / predictive_score_metrics with the reference's signatures: a generator that
    copies the data is indistinguishable (score near 0, well below a shifted one) and predictable."""
    from timeautodiff_amd import Metrics
    rng = np.random.default_rng(5)
    t = np.linspace(0, 3, 12, dtype=np.float32)
    real = np.stack([np.sin(t[None, :] * f[:, None] + p[:, None]) for f, p in zip(rng.uniform(1, 2, (200, 4)).T, rng.uniform(0, 3, (200, 4)).T)], -1)
    real = (0.5 + 0.4 * real).astype(np.float32)                     # [200, 12, 4]
    same = real[rng.permutation(200)].copy()
    shifted = np.clip(real * 0.3, 0, 1).astype(np.float32)
    np.random.seed(0)
    torch.manual_seed(0)
    losses = []
    s_same = Metrics.discriminative_score_metrics(torch.from_numpy(real), torch.from_numpy(same), 150, losses=losses)
    np.random.seed(0)
    torch.manual_seed(0)
    s_shift = Metrics.discriminative_score_metrics(torch.from_numpy(real), torch.from_numpy(shifted), 150)
    assert len(losses) == 150 and np.isfinite(losses).all()
    assert 0.0 <= s_same <= 0.5 and s_shift > 0.4 and s_same < s_shift
    np.random.seed(1)
    torch.manual_seed(1)
    pl = []
    mae = Metrics.predictive_score_metrics(real, same, 1, iterations=200, batch_size=64, losses=pl)
    assert np.isfinite(mae) and mae < 0.5 and pl[-1] < pl[0]


def test_discriminator_step_with_halves_of_different_length_vs_oracle():
    """Fewer generated training windows than the batch size: the reference's batch_generator then returns a shorter
    generated half, and each BCE mean runs over its own count (Evaluation/Metrics.py:96-106).  Loss and every gradient of
    such a step against the oracle (the equal-length goldens do not cover it: parity here is pinned by the oracle, which
    those goldens pin)."""
    from oracle import posthoc as opost
    from timeautodiff_amd import Metrics, ops
    z, meta = fu.load("posthoc_disc")
    dev = torch.device("cuda:0")
    dim = meta["dim"]
    disc = Metrics.Discriminator(dim, meta["hidden"], meta["layers"])
    sd = fu.sub(z, "w")
    _load_into(disc, sd)
    nr, nf = 7, 4
    real, fake = z["real"][:nr], z["fake"][:nf]
    oloss, _, _, ograds = opost.disc_loss_and_grads({k: v.copy() for k, v in sd.items()}, real, fake, meta["layers"])
    eng = disc.engine(dev)
    xb = torch.from_numpy(np.concatenate([real, fake], 0)).to(dev).contiguous()
    yb = torch.empty(nr + nf, 1, device=dev)
    ops.fill_(yb[:nr], 1.0)
    ops.fill_(yb[nr:], 0.0)
    ones = ops.fill_(torch.empty(nr + nf, 1, device=dev), 1.0)
    eng.disc_step_body(xb, yb, ones, (nr, nf), False, lr=0.0)
    torch.cuda.synchronize()
    l4 = eng.loss4.cpu().numpy().astype(np.float64)
    assert abs(l4[0] / nr + l4[2] / nf - oloss) < TOL * oloss
    for k, g in ograds.items():
        assert fu.rel_err(eng.fp.g(k).cpu().numpy(), g.reshape(eng.fp.g(k).shape)) < TOL, k
    # and the metric end to end on sets of different sizes (100 real windows, 40 generated ones: 80 / 32 for training)
    rng = np.random.default_rng(6)
    a, b = rng.random((100, 12, 4), dtype=np.float32), np.clip(rng.random((40, 12, 4), dtype=np.float32) * 0.3, 0, 1)
    np.random.seed(0)
    losses = []
    s = Metrics.discriminative_score_metrics(torch.from_numpy(a), torch.from_numpy(b), 60, losses=losses)
    assert len(losses) == 60 and np.isfinite(losses).all() and losses[-1] < losses[0] and 0.0 <= s <= 0.5
