"""Device handling and argument guards of the Python layer (GPU): the engine is keyed by an indexed device, operands
on a device that is not current are rejected before any launch, short data tables raise the reference's IndexError,
and an optimizer step leaves parameters that received no gradient untouched."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _denoiser(H=24, E=16):
    from timeautodiff_amd import DIFF
    torch.manual_seed(3)
    return DIFF.BiRNN_score(6, 6, H, 2, 100, 8, E, 0, 0, 6, [])


def test_engine_is_not_rebuilt_for_an_equivalent_device_name():
    m = _denoiser()
    e0 = m.engine("cuda")
    assert e0.device == torch.device("cuda", torch.cuda.current_device())
    assert m.engine("cuda:0") is e0 and m.engine(torch.device("cuda")) is e0 and m.engine() is e0
    x = torch.randn(2, 8, 6, device="cuda:0")
    t = torch.rand(2, 8, 1, device="cuda:0").sort(1)[0]
    i = torch.full((2, 8, 1), 3.0, device="cuda:0")
    m(x, t, i, torch.randn(2, 8, 8, device="cuda:0"), torch.rand(2, 8, 6, device="cuda:0"))
    assert m.engine() is e0                       # a forward on cuda:0 tensors keeps the engine built for "cuda"


def test_cpu_operands_are_rejected_before_any_launch():
    from timeautodiff_amd import _lib, ops
    a = torch.zeros(4, 8)
    with pytest.raises(_lib.TadError):
        ops.silu_fwd(a, torch.zeros(4, 8, device="cuda"))


def test_trainer_rejects_tables_shorter_than_the_latent():
    from timeautodiff_amd import DIFF
    m = _denoiser()
    lat = torch.randn(10, 8, 6)
    with pytest.raises(IndexError):
        DIFF.DiffusionTrainer(m, lat, torch.randn(9, 8, 8), torch.rand(10, 8, 6), batch_size=4, dev="cuda")
    with pytest.raises(ValueError):
        DIFF.DiffusionTrainer(m, lat, torch.randn(10, 7, 8), torch.rand(10, 8, 6), batch_size=4, dev="cuda")


def test_vae_step_without_cond_leaves_the_conditioning_branch_untouched():
    """torch.optim.Adam skips parameters whose grad is None, weight decay included (reference VAE.py:315, 337-345
    with cond None): Emb_decoder.* and Emb_hidden_decoder.* must not decay."""
    from timeautodiff_amd import VAE
    torch.manual_seed(5)
    F, T = 6, 8
    ae = VAE.DeapStack(1, 0, 0, F, [], F, 24, 2, 16, 8, 6)
    before = {k: v.detach().cpu().clone() for k, v in ae.state_dict().items()}
    x = torch.rand(12, T, F)
    ones = torch.ones(12, T, F)
    dd = dict(response_train=x, target_mask_train=ones, cond_train=None, response_val=x[:4], target_mask_val=ones[:4], cond_val=None)
    tr = VAE.VAETrainer(ae, dd, batch_size=6, lr=1e-2, weight_decay=0.1, dtype="fp32", dev="cuda")
    import random
    random.seed(0)
    for _ in range(3):
        tr.train_step(0.1)
    torch.cuda.synchronize()
    after = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert any(k.startswith("fc_mu") for k in moved)
    frozen = [k for k in before if k.startswith("Emb_decoder.") or k.startswith("Emb_hidden_decoder.")]
    assert frozen and all(torch.equal(before[k], after[k]) for k in frozen)


def test_vae_bf16_with_a_hidden_size_that_is_not_a_multiple_of_8():
    """H = 100, two layers: heads-mode layers slice their input per head at multiples of H, which bf16 rows cannot
    do; the stack falls back to fp32 storage for its per-step tensors and must simply work."""
    from timeautodiff_amd import VAE
    torch.manual_seed(6)
    F, T = 6, 8
    ae = VAE.DeapStack(1, 0, 0, F, [], F, 100, 2, 16, 8, 6)
    ae.set_compute_dtype("bf16")
    x = torch.rand(5, T, F, device="cuda")
    out = ae(x, torch.ones_like(x))
    torch.cuda.synchronize()
    leaves = out[0]["nums"] if isinstance(out, tuple) else out["nums"]
    assert torch.isfinite(leaves).all()


def test_empty_shard_is_a_no_op_for_the_sequence_kernels():
    from timeautodiff_amd import _lib
    lib = _lib.load()
    dirs = (_lib.RnnDir * 1)()
    assert lib.tad_rnn_fwd(0, 0, 0, 0, 8, 24, 1, dirs, None) == 0
    bd = (_lib.RnnBwdDir * 1)()
    assert lib.tad_rnn_bwd(0, 0, 0, 0, 8, 24, 1, bd, None) == 0


def test_vae_trainer_on_windowed_tables_equals_stacked_windows():
    """response_train / cond_train as WindowedTable views of their tables (reference dataprovider_pypots.py:75-87 cuts
    and stacks them; VAE.py:326-348 re-copies and indexes the stacked form every iteration): same losses and weights
    as the trainer fed with the stacked tensors, on the same draws."""
    import random
    from timeautodiff_amd import VAE
    from timeautodiff_amd.dataprovider_pypots import WindowedTable, sliding_window
    g = torch.Generator().manual_seed(31)
    rows, F, T, stride = 60, 6, 8, 2
    table = torch.rand(rows, F, generator=g)
    ctab = torch.rand(rows, F, generator=g) * 0.1
    resp, cond = sliding_window(table, T, stride).cpu(), sliding_window(ctab, T, stride).cpu()
    N = resp.shape[0]
    mask = (torch.rand(N, T, F, generator=g) < 0.8).float()
    val = dict(response_val=resp[:5], target_mask_val=mask[:5], cond_val=cond[:5])
    runs = []
    for windowed in (False, True):
        torch.manual_seed(33)
        random.seed(33)
        ae = VAE.DeapStack(1, 0, 0, F, [], F, 24, 2, 16, 8, 6)
        dd = dict(response_train=WindowedTable(table, T, stride) if windowed else resp, target_mask_train=mask,
                  cond_train=WindowedTable(ctab, T, stride) if windowed else cond, **val)
        tr = VAE.VAETrainer(ae, dd, batch_size=8, lr=1e-3, weight_decay=0.0, dtype="fp32", dev="cuda", graph=False)
        assert tr.windowed == windowed and tr.N == N
        eg = torch.Generator().manual_seed(35)
        losses = [tr.train_step(0.1, eps=torch.randn(8, T, 6, generator=eg)) for _ in range(3)]
        out, latent = tr.full_forward(eps=torch.randn(N, T, 6, generator=eg))
        torch.cuda.synchronize()
        runs.append((losses, tr.eng.fp.flat[:tr.eng.fp.n_active].cpu().clone(), latent.cpu().clone()))
    assert max(abs(a - b) / abs(a) for a, b in zip(runs[0][0], runs[1][0])) < 2e-5
    assert float((runs[0][1] - runs[1][1]).norm() / runs[0][1].norm()) < 1e-4
    assert float((runs[0][2] - runs[1][2]).norm() / runs[0][2].norm()) < 1e-3
    with pytest.raises(IndexError):
        tr.train_step(0.1, idx=list(range(N - 7, N + 1)))


def test_launch_trace_bills_every_launch_its_own_work_and_adds_up_to_the_model():
    """The per-kernel figures of the bench come from ops.LaunchTrace: every launch is billed its own algorithmic FLOPs
    (the weight-gradient launches of a step's four recurrent layers share an entry point and leading arguments but not
    their work), a GRU's spare gate slot and tile padding count as issued, not as algorithmic work, and the launches of
    one training step add up to the model's 3 x forward FLOPs (a little less: no gradient is taken with respect to the
    model's inputs)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from timeautodiff_amd import ops
    from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer
    dev = torch.device("cuda:0")
    B = 512
    torch.manual_seed(3)
    latent, time_info, cond = bench.synth(2 * B, 3, dev)
    model = BiRNN_score(bench.K, bench.F, bench.H, 2, 100, 8, bench.E, 0, 1, 27, [8])
    tr = DiffusionTrainer(model, latent, time_info, cond, batch_size=B, dtype="bf16", draws="device", dev=dev)
    for _ in range(2):
        tr.step()
    with ops.LaunchTrace() as t:
        tr.step()
    recs = [r for r in t.records if r[0][0] == "tad_mm16_tn"]
    costs = sorted(r[4][0] for r in recs)
    assert len(recs) == 5 and len(set(costs)) >= 4, costs            # LSTM l0 / l1, GRU l0 / l1, the thin products
    by = t.summary_by_kernel()
    tn = by["mm16_tn_kernel"]
    assert abs(sum(c * fl for c, _, fl, _, _ in tn.values()) - sum(costs)) < 1e-6 * sum(costs)
    assert all(iss >= fl for c, _, fl, _, iss in tn.values())
    total = sum(c * fl for calls in by.values() for c, _, fl, _, _ in calls.values())
    model_flops = 3 * 2 * bench.fwd_macs_per_sample() * B
    print("traced / model FLOPs per step: %.4f" % (total / model_flops))
    assert 0.90 < total / model_flops < 1.005
