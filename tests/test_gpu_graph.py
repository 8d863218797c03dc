"""A training step replayed from its captured HIP graph equals the step enqueued kernel by kernel.

Both run the same kernels on the same inputs; what differs between ANY two runs of a step -- eager or replayed -- is
the order in which float atomics land (split reductions of the weight gradients, the loss sum), i.e. rounding.  The
bound is therefore the run-to-run bound of two eager runs: 1e-6 relative on the first loss (one step, no accumulated
drift), 2e-5 on the loss curve and 1e-4 on the weights after five steps (fp32)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _denoiser_run(graph, draws):
    from timeautodiff_amd import DIFF
    torch.manual_seed(21)
    N, T, K, H, E = 64, 8, 6, 24, 16
    g = torch.Generator().manual_seed(3)
    cond = torch.cat([torch.randint(0, 3, (N, T, 1), generator=g).float(), torch.rand(N, T, 4, generator=g)], -1)
    ti, lat = torch.randn(N, T, 8, generator=g), torch.randn(N, T, K, generator=g)
    model = DIFF.BiRNN_score(K, 5, H, 2, 100, 8, E, 0, 1, 4, [3])
    tr = DIFF.DiffusionTrainer(model, lat, ti, cond, batch_size=16, lr=1e-3, dtype="fp32", draws="reference", dev="cuda", graph=graph)
    hist = torch.zeros(len(draws), device="cuda")
    for s, dr in enumerate(draws):
        tr.step(draw=dr, loss_out=hist[s:s + 1])
    torch.cuda.synchronize()
    assert (tr._graph is not None) == bool(graph)
    return hist.cpu(), tr.eng.fp.flat[:tr.eng.fp.n_active].cpu().clone(), tr.eng.fp.step


def test_denoiser_step_graph_equals_eager():
    g = torch.Generator().manual_seed(5)
    draws = [dict(idx=torch.randperm(64, generator=g)[:16], t=torch.rand(16, 8, 1, generator=g).sort(1)[0],
                  i=torch.randint(0, 100, (16,), generator=g), noise=torch.randn(16, 8, 6, generator=g)) for _ in range(5)]
    le, we, se = _denoiser_run(False, draws)
    lg, wg, sg = _denoiser_run(True, draws)
    assert se == sg == 5                                   # the warm-up steps of the capture leave no trace in the state
    assert abs(float(le[0]) - float(lg[0])) < 1e-6 * abs(float(le[0]))
    assert float(((le - lg).abs() / le.abs()).max()) < 2e-5
    assert _rel(wg, we) < 1e-4


def _vae_run(graph):
    from timeautodiff_amd import VAE
    torch.manual_seed(23)
    random.seed(23)
    N, T, F = 40, 8, 6
    g = torch.Generator().manual_seed(7)
    x = torch.cat([torch.randint(0, 2, (N, T, 1), generator=g).float(), torch.randint(0, 4, (N, T, 1), generator=g).float(),
                   torch.rand(N, T, 4, generator=g)], -1)
    mask = (torch.rand(N, T, F, generator=g) < 0.8).float()
    dd = dict(response_train=x[:32] * mask[:32], target_mask_train=mask[:32], cond_train=x[:32] * (1 - mask[:32]),
              response_val=x[32:] * mask[32:], target_mask_val=mask[32:], cond_val=x[32:] * (1 - mask[32:]))
    ae = VAE.DeapStack(1, 1, 1, 4, [4], F, 24, 2, 16, 8, 6)
    tr = VAE.VAETrainer(ae, dd, batch_size=12, lr=2e-4, weight_decay=1e-6, dtype="fp32", dev="cuda", graph=graph)
    eg = torch.Generator().manual_seed(9)
    tl, vl = [], []
    for s in range(5):
        beta = 0.1 if s < 3 else 0.07                      # a change of beta takes another captured graph
        tl.append(tr.train_step(beta, eps=torch.randn(12, T, 6, generator=eg)))
        vl.append(tr.eval_loss(beta, eps=torch.randn(8, T, 6, generator=eg)))
    torch.cuda.synchronize()
    assert (len(tr._graphs) == 3) == bool(graph)           # two betas + the validation pass
    return np.array(tl), np.array(vl), tr.eng.fp.flat[:tr.eng.fp.n_active].cpu().clone()


def test_vae_step_and_validation_graph_equal_eager():
    te, ve, we = _vae_run(False)
    tg, vg, wg = _vae_run(True)
    assert abs(te[0] - tg[0]) < 1e-6 * abs(te[0])
    assert np.max(np.abs(te - tg) / np.abs(te)) < 2e-5 and np.max(np.abs(ve - vg) / np.abs(ve)) < 2e-5
    assert _rel(wg, we) < 1e-4


def test_captured_steps_survive_module_calls_on_other_batch_sizes():
    """Between two replays of a captured step the caller uses the model itself on batches of other sizes (ae.encoder /
    ae.decoder / full_forward; model(...) and sample() for the denoiser): those calls re-size the engine's default
    workspace, which must not be the memory the captured graph works in.  The interleaved run equals the eager one."""
    from timeautodiff_amd import DIFF, VAE
    # ---- VAE
    def vae_run(graph):
        torch.manual_seed(31)
        random.seed(31)
        N, T, F = 40, 8, 6
        g = torch.Generator().manual_seed(7)
        x = torch.cat([torch.randint(0, 2, (N, T, 1), generator=g).float(), torch.randint(0, 4, (N, T, 1), generator=g).float(),
                       torch.rand(N, T, 4, generator=g)], -1)
        mask = (torch.rand(N, T, F, generator=g) < 0.8).float()
        dd = dict(response_train=x[:32] * mask[:32], target_mask_train=mask[:32], cond_train=x[:32] * (1 - mask[:32]),
                  response_val=x[32:] * mask[32:], target_mask_val=mask[32:], cond_val=x[32:] * (1 - mask[32:]))
        ae = VAE.DeapStack(1, 1, 1, 4, [4], F, 24, 2, 16, 8, 6)
        tr = VAE.VAETrainer(ae, dd, batch_size=12, lr=2e-4, weight_decay=1e-6, dtype="fp32", dev="cuda", graph=graph)
        eg = torch.Generator().manual_seed(9)
        out = []
        for s in range(4):
            out.append(tr.train_step(0.1, eps=torch.randn(12, T, 6, generator=eg)))
            out.append(tr.eval_loss(0.1, eps=torch.randn(8, T, 6, generator=eg)))
            # the module surface on 5, then 40 windows: allocations of many sizes land where freed buffers were
            z, mu, lv = ae.encoder(x[:5 + s], eps=torch.zeros(5 + s, T, 6))
            dec = ae.decoder(z, mask[:5 + s], x[:5 + s] * (1 - mask[:5 + s]))
            outd, lat = tr.full_forward(eps=torch.zeros(32, T, 6))
            junk = [torch.full((257 * (k + 1),), float("nan"), device="cuda") for k in range(8)]      # re-use freed blocks
            out.append(float(mu.sum()) + float(dec["nums"].sum()) + float(lat.sum()))
            del junk
        torch.cuda.synchronize()
        return np.array(out), tr.eng.fp.flat[:tr.eng.fp.n_active].cpu().clone()
    (oe, we), (og, wg) = vae_run(False), vae_run(True)
    assert np.all(np.isfinite(og)) and np.max(np.abs(oe - og) / np.abs(oe)) < 2e-5 and _rel(wg, we) < 1e-4

    # ---- denoiser
    def den_run(graph):
        torch.manual_seed(33)
        N, T, K, H, E = 64, 8, 6, 24, 16
        g = torch.Generator().manual_seed(3)
        cond = torch.cat([torch.randint(0, 3, (N, T, 1), generator=g).float(), torch.rand(N, T, 4, generator=g)], -1)
        ti, lat = torch.randn(N, T, 8, generator=g), torch.randn(N, T, K, generator=g)
        model = DIFF.BiRNN_score(K, 5, H, 2, 100, 8, E, 0, 1, 4, [3])
        tr = DIFF.DiffusionTrainer(model, lat, ti, cond, batch_size=16, lr=1e-3, dtype="fp32", draws="reference", dev="cuda", graph=graph)
        dg = torch.Generator().manual_seed(5)
        hist = torch.zeros(4, device="cuda")
        extra = []
        for s in range(4):
            dr = dict(idx=torch.randperm(64, generator=dg)[:16], t=torch.rand(16, 8, 1, generator=dg).sort(1)[0],
                      i=torch.randint(0, 100, (16,), generator=dg), noise=torch.randn(16, 8, 6, generator=dg))
            tr.step(draw=dr, loss_out=hist[s:s + 1])
            nb = 3 + s
            y = model(lat[:nb], torch.rand(nb, T, 1, generator=dg), torch.full((nb, T, 1), 7.0), ti[:nb], cond[:nb], None)
            tg = torch.linspace(0, 1, T).view(1, -1, 1).repeat(nb, 1, 1)
            xs = DIFF.sample(tg, nb, T, K, 100, model, ti[:nb], cond[:nb], None,
                             noise=dict(x_T=torch.zeros(nb, T, K), z=torch.zeros(100, nb, T, K)))
            junk = [torch.full((131 * (k + 1),), float("nan"), device="cuda") for k in range(8)]
            extra.append(float(y.sum()) + float(xs.sum()))
            del junk
        torch.cuda.synchronize()
        return hist.cpu().numpy(), np.array(extra), tr.eng.fp.flat[:tr.eng.fp.n_active].cpu().clone()
    (he, xe, we), (hg, xg, wg) = den_run(False), den_run(True)
    assert np.max(np.abs(he - hg) / np.abs(he)) < 2e-5 and np.max(np.abs(xe - xg) / np.maximum(np.abs(xe), 1e-6)) < 1e-3
    assert _rel(wg, we) < 1e-4
