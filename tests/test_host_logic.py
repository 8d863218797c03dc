"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared
symbol, parameter flattening keeps the reference's state_dict contract, the data-parallel
plan is equivalent to a single process (world_size 2, gloo), and the product path refuses
to run without a GPU instead of falling back to the CPU."""
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixture_util as fu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubParser:
    """Stands in for process_edited.DataFrameParser: only datatype_info() is read on the path."""

    def __init__(self, schema):
        self.s = schema

    def datatype_info(self):
        return dict(n_bins=self.s["n_bins"], n_cats=self.s["n_cats"], n_nums=self.s["n_nums"], cards=list(self.s["cards"]))


def test_library_exports_every_declared_symbol(lib):
    from timeautodiff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tad_hip.h")).read()
    declared = set(re.findall(r"\b(tad_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tad_abi_version() == 16
    # bf16, H = 200, LSTM: the forward buffer holds the 32-row-form image (7 chunks x 13 k-blocks x 4 gates) and the
    # 16-row-form image (7 x 7 k-steps x 4 gates x 2 blocks of 16 units), the backward one 7 x 52 k-blocks; 1 KiB each
    assert lib.tad_rnn_packed_bytes(1, 200, 4) == max(7 * 13 * 4 + 7 * 7 * 4 * 2, 7 * (4 * 224 // 16)) * 1024
    assert lib.tad_rnn_packed_bytes(0, 200, 4) == max(7 * 25 * 4, 7 * (4 * 224 // 8)) * 1024
    assert lib.tad_rnn_packed_bytes(0, 200, 3) == 7 * (3 * 224 // 8) * 1024
    assert lib.tad_rnn_packed_bytes(0, 0, 4) == -1


def test_flat_params_keep_state_dict_contract():
    from timeautodiff_amd import DIFF
    from timeautodiff_amd.engine import FlatParams, RnnStack, EmbBlock
    z, meta = fu.load("den_mixed_small")
    sd = fu.sub(z, "w")
    s = meta["schema"]
    torch.manual_seed(1)
    m = DIFF.BiRNN_score(meta["K"], 6, meta["H"], 2, 100, 8, meta["E"], s["n_bins"], s["n_cats"], s["n_nums"], s["cards"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    assert all(tuple(m.state_dict()[k].shape) == sd[k].shape for k in sd)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    lstm_dirs = [("lstm.", "_l{l}"), ("lstm.", "_l{l}_reverse")]
    names = [n for n, _ in m.named_parameters()]
    layout = RnnStack.layout(lstm_dirs, 2)
    rest = [n for n in names if n not in dict(layout) and not n.startswith("Emb_hidden")]
    fp = FlatParams(m, layout + [(n, False) for n in rest], ["Emb_hidden.weight", "Emb_hidden.bias"], torch.device("cpu"))
    assert fp.owns(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    # glued groups are contiguous: both directions' W_ih form one [2*4H, H] matrix
    H = meta["H"]
    w = fp.pgroup("lstm.weight_ih_l0", 8 * H, H)
    assert torch.equal(w[:4 * H], m.lstm.weight_ih_l0) and torch.equal(w[4 * H:], m.lstm.weight_ih_l0_reverse)
    # unused parameters sit outside the optimizer / all-reduce range and have no gradient
    assert fp.offsets["Emb_hidden.weight"] >= fp.n_active and m.Emb_hidden.weight.grad is None
    assert m.fc.weight.grad is not None and m.fc.weight.grad.data_ptr() == fp.g("fc.weight").data_ptr()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert fp.owns(m) and np.array_equal(fp.p("fc.weight").numpy(), sd["fc.weight"])


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from timeautodiff_amd import DIFF, VAE
    s = fu.SCHEMAS["numeric"]
    m = DIFF.BiRNN_score(6, 6, 16, 2, 100, 8, 16, 0, 0, 6, [])
    x = torch.zeros(2, 4, 6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.zeros(2, 4, 1), torch.zeros(2, 4, 1), torch.zeros(2, 4, 8), x, None)
    data = fu.make_data(s, N=8, T=4, K=4, seed=0)
    dd = dict(parser=StubParser(s), cond_train=torch.from_numpy(data["cond"]), time_info_train=torch.from_numpy(data["time_info"]),
              target_mask_train=torch.from_numpy(data["mask"]))
    with pytest.raises(RuntimeError):
        DIFF.train_diffusion(torch.from_numpy(data["latent"]), None, dd, 16, 2, 100, 1, batch_size=4)
    with pytest.raises(RuntimeError):
        VAE.DeapStack(1, 0, 0, 6, [], 6, 16, 2, 16, 8, 4).encoder(x)
    with pytest.raises(ValueError):
        DIFF.train_diffusion(torch.from_numpy(data["latent"]), None, dd, 16, 2, 50, 1)   # schedule is 100 steps


def test_slice_bounds_cover_the_batch():
    from timeautodiff_amd.dist import slice_bounds
    for n in (100, 101, 7, 4096):
        for w in (1, 2, 3, 8):
            b = [slice_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import denoiser as oden, ops as oops
        from oracle.denoiser import Schema
        from timeautodiff_amd.dist import DataParallelPlan
        z, meta = fu.load("den_mixed_small")
        s = meta["schema"]
        sch = Schema(s["n_bins"], s["n_cats"], s["n_nums"], list(s["cards"]))
        sd = fu.sub(z, "w")
        B, T, K = meta["B"], meta["T"], meta["K"]              # B = 5: not divisible by 2
        data = fu.make_data(s, N=B, T=T, K=K, seed=meta["seed"], mask_kind="bern")
        _, alphas = oops.schedule(100)
        plan = DataParallelPlan(B, True)
        lo, hi = plan.lo, plan.hi
        loss, pred, grads = oden.loss_and_grads(sd, data["latent"][lo:hi], z["t"][lo:hi], z["i_idx"][lo:hi], z["noise"][lo:hi],
                                                data["time_info"][lo:hi], data["cond"][lo:hi], sch, alphas, global_numel=B * T * K)
        keys = sorted(grads)
        flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in keys]).astype(np.float32))
        loss_sum = torch.tensor([loss * (hi - lo) * T * K], dtype=torch.float64)
        # two buckets started asynchronously (the trainer's pattern), then the loss with a blocking call
        nh = flat.numel() // 3
        pending = [plan.all_reduce_async(flat[:nh]), plan.all_reduce_async(flat[nh:])]
        for w in pending:
            w.wait()
        plan.all_reduce(loss_sum)
        gathered = plan.all_gather_cat(torch.from_numpy(pred.astype(np.float32)))
        if rank == 0:
            ret["flat"] = flat.numpy()
            ret["loss"] = float(loss_sum.item() / (B * T * K))
            ret["pred"] = gathered.numpy()
            ret["keys"] = keys
    finally:
        dist.destroy_process_group()


def test_data_parallel_equals_single_process_gloo():
    """world_size 2 over gloo: sliced global batch + summed gradients == one process."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    from oracle import denoiser as oden, ops as oops
    from oracle.denoiser import Schema
    z, meta = fu.load("den_mixed_small")
    gref = fu.sub(z, "g")
    flat_ref = np.concatenate([gref[k].ravel() for k in ret["keys"]])
    assert fu.rel_err(ret["flat"], flat_ref) < 5e-5
    assert abs(ret["loss"] - float(z["loss"])) < 1e-5 * float(z["loss"])
    assert fu.rel_err(ret["pred"], z["pred"]) < 2e-5


def _dp8_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from oracle import denoiser as oden, ops as oops
        from oracle.denoiser import Schema
        from timeautodiff_amd.dist import DataParallelPlan
        sd, sch_d, data, t, i_idx, noise, B, Bs = _dp8_problem()
        sch = Schema(sch_d["n_bins"], sch_d["n_cats"], sch_d["n_nums"], list(sch_d["cards"]))
        T, K = data["latent"].shape[1:]
        _, alphas = oops.schedule(100)
        keys = sorted(k for k in sd if not k.startswith("Emb_hidden"))
        nflat = sum(sd[k].size for k in keys)
        # the trainers' buffer: gradients, then the loss partial sum in the slot right behind them -- ONE all-reduce
        flat = torch.zeros(nflat + 1, dtype=torch.float64)
        plan = DataParallelPlan(B, True)
        lo, hi = plan.lo, plan.hi
        assert hi - lo in (B // world, B // world + 1)
        loss, pred, grads = oden.loss_and_grads(sd, data["latent"][lo:hi], t[lo:hi], i_idx[lo:hi], noise[lo:hi], data["time_info"][lo:hi],
                                                data["cond"][lo:hi], sch, alphas, global_numel=B * T * K)
        flat[:nflat] = torch.from_numpy(np.concatenate([grads[k].ravel() for k in keys]).astype(np.float64))
        flat[nflat] = loss * (hi - lo) * T * K
        first = nflat // 2
        pending = [plan.all_reduce_async(flat[:first]), plan.all_reduce_async(flat[first:])]
        for w in pending:
            w.wait()
        # a batch smaller than the world (sampling / validation of Bs = 5 items over 8 ranks): three ranks own nothing,
        # still take part in the collectives, and contribute zeros / zero rows
        splan = DataParallelPlan(Bs, True)
        slo, shi = splan.lo, splan.hi
        part = torch.zeros(2, dtype=torch.float64)
        mine = torch.zeros(shi - slo, T, K)
        if shi > slo:
            i0 = np.zeros((shi - slo, T, 1), np.float32)
            p, _ = oden.forward(sd, data["latent"][slo:shi], t[slo:shi], i0, data["time_info"][slo:shi], data["cond"][slo:shi], sch)
            mine = torch.from_numpy(p.astype(np.float32))
            part[0], part[1] = float((p.astype(np.float64) ** 2).sum()), shi - slo
        plan.all_reduce(part)
        whole = splan.all_gather_cat(mine)
        ret[rank] = (shi - slo, tuple(whole.shape))
        if rank == world - 1:          # every rank holds the same results: take them from the last one
            ret["flat"] = flat.numpy()
            ret["part"] = part.numpy()
            ret["whole"] = whole.numpy()
    finally:
        dist.destroy_process_group()


def _dp8_problem():
    from timeautodiff_amd import DIFF
    sch_d = fu.SCHEMAS["mixed"]
    B, Bs, T, K, H, E = 100, 5, 4, 3, 8, 8
    torch.manual_seed(3)
    sd = {k: v.numpy().copy() for k, v in DIFF.init_denoiser_state(K, H, 2, 100, 8, E, sch_d["n_bins"], sch_d["n_cats"], sch_d["n_nums"],
                                                                   sch_d["cards"]).items()}
    data = fu.make_data(sch_d, N=B, T=T, K=K, seed=4, mask_kind="ones")
    rng = np.random.default_rng(5)
    t = np.sort(rng.random((B, T, 1), dtype=np.float32), 1)
    return sd, sch_d, data, t, rng.integers(0, 100, B), rng.standard_normal((B, T, K)).astype(np.float32), B, Bs


def test_data_parallel_world_8_with_uneven_and_empty_shards_gloo():
    """The reference's batch of 100 windows over 8 ranks (shards of 12 and 13), gradients and the loss slot reduced as the
    trainers reduce them (two asynchronous buckets of one flat buffer), and a batch of 5 over 8 ranks (three empty
    shards) through the partial-sum reduction and the padded all-gather: equal to one process."""
    from oracle import denoiser as oden, ops as oops
    from oracle.denoiser import Schema
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp8_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sd, sch_d, data, t, i_idx, noise, B, Bs = _dp8_problem()
    sch = Schema(sch_d["n_bins"], sch_d["n_cats"], sch_d["n_nums"], list(sch_d["cards"]))
    T, K = data["latent"].shape[1:]
    _, alphas = oops.schedule(100)
    loss, _, grads = oden.loss_and_grads(sd, data["latent"], t, i_idx, noise, data["time_info"], data["cond"], sch, alphas)
    keys = sorted(k for k in sd if not k.startswith("Emb_hidden"))
    ref = np.concatenate([grads[k].ravel() for k in keys])
    flat = ret["flat"]
    assert fu.rel_err(flat[:-1], ref) < 1e-5
    assert abs(flat[-1] / (B * T * K) - loss) < 1e-6 * abs(loss)
    assert sorted(ret[r][0] for r in range(world)) == [0, 0, 0, 1, 1, 1, 1, 1]
    assert all(ret[r][1] == (Bs, T, K) for r in range(world))
    p, _ = oden.forward(sd, data["latent"][:Bs], t[:Bs], np.zeros((Bs, T, 1), np.float32), data["time_info"][:Bs], data["cond"][:Bs], sch)
    assert fu.rel_err(ret["whole"], p) < 1e-5
    assert ret["part"][1] == Bs and abs(ret["part"][0] - float((p.astype(np.float64) ** 2).sum())) < 1e-5 * ret["part"][0]
