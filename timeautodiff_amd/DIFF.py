"""Drop-in for the reference's DIFF.py hot path on MI355X.

Same public names and call signatures as the reference (``BiRNN_score``,
``get_loss``, ``add_noise``, ``train_diffusion``, ``sample``, ``betas``, ``alphas``),
but every FLOP and byte on the path runs in hand-written gfx950 kernels behind the
C ABI of libtad_hip.so (see include/tad_hip.h).  torch provides parameter
initialisation (same constructors in the same order as the reference, so seeded
initial weights and state_dict keys are identical), device memory, streams and the
process group.  There is no CPU compute path: calling the model without the built
library or without a GPU raises.

Extra behaviour is reachable only through keyword-only arguments whose defaults
reproduce the reference.
"""
import math
import os
import random
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .dist import DataParallelPlan
from .engine import CounterNoise, DenoiserEngine, Workspace, prec_of, use_workspace

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


# ---------------------------------------------------------------------------
# schedule tables -- reference DIFF.py:88-95 (module globals, always 100 steps)
def get_betas(steps):
    beta_start, beta_end = 1e-5, 0.1
    u = torch.linspace(0, 1, steps)
    return beta_start * (1 - u) + beta_end * u


diffusion_steps = 100
betas = get_betas(diffusion_steps)
alphas = torch.cumprod(1 - betas, dim=0)


# ---------------------------------------------------------------------------
# parameter containers: same attribute names / constructor order as the reference
# modules so that state_dict keys and seeded initial values coincide.
class _FeedForward(nn.Module):                      # reference DIFF.py:148-167 (hidden_dims == [])
    def __init__(self, in_dim, out_dim, final_relu=False):
        super().__init__()
        layers = [nn.Linear(in_dim, out_dim)]
        if final_relu:
            layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers)


class _PositionalEncoding(nn.Module):               # reference DIFF.py:129-146
    def __init__(self, dim, max_value):
        super().__init__()
        self.max_value = max_value
        linear_dim = dim // 2
        periodic_dim = dim - linear_dim
        self.scale = torch.exp(-2 * torch.arange(0, periodic_dim).float() * math.log(self.max_value) / periodic_dim)
        self.shift = torch.zeros(periodic_dim)
        self.shift[::2] = 0.5 * math.pi
        self.linear_proj = nn.Linear(1, linear_dim)


class _EmbeddingData(nn.Module):                    # reference DIFF.py:35-63 / VAE.py:58-86
    def __init__(self, out_dim, emb_dim, n_bins, n_cats, n_nums, cards):
        super().__init__()
        n_disc = n_bins + n_cats
        if n_disc != 0:
            self.embeddings_list = nn.ModuleList([nn.Embedding(c, emb_dim) for c in [2] * n_bins + list(cards)])
        if n_nums != 0:
            self.mlp_nums = nn.Sequential(nn.Linear(16 * n_nums, 16 * n_nums), nn.SiLU(), nn.Linear(16 * n_nums, 16 * n_nums))
        self.mlp_output = nn.Sequential(nn.Linear(emb_dim * n_disc + 16 * n_nums, emb_dim), nn.ReLU(),
                                        nn.Linear(emb_dim, out_dim))


class BiRNN_score(nn.Module):
    """The diffusion denoiser (reference DIFF.py:170-224).  ``forward`` runs the HIP
    engine; ``target_mask`` is accepted and ignored exactly like the reference, and
    ``Emb_hidden`` exists only because the reference constructs (and never uses) it."""

    def __init__(self, input_size, org_feat_dim, hidden_size, num_layers, diffusion_steps, time_dim, emb_dim,
                 n_bins, n_cats, n_nums, cards):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.cfg = dict(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                        diffusion_steps=diffusion_steps, time_dim=time_dim, emb_dim=emb_dim, n_bins=n_bins,
                        n_cats=n_cats, n_nums=n_nums, cards=list(cards))
        self.input_proj = _FeedForward(input_size, hidden_size)
        self.t_enc = _PositionalEncoding(hidden_size, max_value=1)
        self.i_enc = _PositionalEncoding(hidden_size, max_value=diffusion_steps)
        self.proj = _FeedForward(4 * hidden_size, hidden_size, final_relu=True)
        self.lstm = nn.LSTM(hidden_size, hidden_size, num_layers, batch_first=True, bidirectional=True)
        self.layer_norm = nn.LayerNorm(2 * hidden_size)
        self.fc = nn.Linear(2 * hidden_size, input_size)
        self.Emb = _EmbeddingData(emb_dim, emb_dim, n_bins, n_cats, n_nums, cards)
        self.Emb_hidden = nn.Linear(emb_dim, hidden_size)
        self.cond_GRU = nn.GRU(emb_dim, hidden_size, num_layers, batch_first=True, bidirectional=True)
        self.cond_output = nn.Linear(2 * hidden_size, hidden_size)
        self.time_encode = nn.Sequential(nn.Linear(time_dim, hidden_size), nn.ReLU(), nn.Linear(hidden_size, hidden_size))
        self._eng = None
        self._eng_dtype = "fp32"
        self.register_load_state_dict_post_hook(lambda m, keys: m.mark_params_changed())

    # -- engine plumbing ----------------------------------------------------
    def set_compute_dtype(self, dtype):
        """'fp32' (exact fp32 MFMA; default) or 'bf16' (bf16 MFMA operands, fp32 accumulate)."""
        prec_of(dtype)
        if dtype != self._eng_dtype:
            self._eng_dtype = dtype
            self._eng = None
        return self

    def engine(self, dev=None) -> DenoiserEngine:
        dev = torch.device(dev) if dev is not None else (self._eng.device if self._eng is not None else device)
        if dev.type == "cuda":
            dev = ops.norm_device(dev)              # 'cuda' and 'cuda:0' name the same engine
        if dev.type != "cuda":
            raise RuntimeError("BiRNN_score runs only on an MI355X through libtad_hip.so; no GPU is visible "
                               "(there is no CPU fallback)")
        if self._eng is None or self._eng.device != dev or not self._eng.fp.owns(self):
            self._eng = DenoiserEngine(self, dev, self._eng_dtype)
        return self._eng

    def mark_params_changed(self):
        """Call after modifying parameter values in place outside of this package."""
        if self._eng is not None:
            self._eng.fp.version += 1

    @ops.device_scope(lambda self, x, *a, **k: ops._first_cuda(x, self))
    def forward(self, x, t, i, time_info=None, cond=None, target_mask=None):
        if time_info is None or cond is None:
            raise NotImplementedError("time_info and cond are required (every reference driver passes both)")
        shape = x.shape
        eng = self.engine(x.device if x.is_cuda else None)
        dev = eng.device
        eng.fp.version += 1          # parameters may have been edited in place by the caller: re-pack W_hh

        def prep(a, last):
            return a.to(device=dev, dtype=torch.float32).reshape(-1, shape[-2], last).contiguous()
        K = shape[-1]
        out = eng.forward(prep(x, K), prep(t, 1), prep(i, 1), prep(time_info, time_info.shape[-1]),
                          prep(cond, cond.shape[-1]), train=False)
        return out.clone()


def init_denoiser_state(input_size, hidden_size, num_layers, diffusion_steps, time_dim, emb_dim, n_bins, n_cats,
                        n_nums, cards):
    """state_dict (CPU tensors) of a freshly constructed BiRNN_score under the current torch seed."""
    m = BiRNN_score(input_size, 0, hidden_size, num_layers, diffusion_steps, time_dim, emb_dim, n_bins, n_cats,
                    n_nums, list(cards))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


# ---------------------------------------------------------------------------
def add_noise(x, t, i, *, noise=None):
    """reference DIFF.py:102-118.  x [B,S,D], i [B,S,1] (constant over S).  The GP covariance of
    the reference is the identity, so the noise is plain N(0, I).  Returns (x_noisy, noise)."""
    dev = x.device
    if dev.type != "cuda":
        raise RuntimeError("add_noise runs on the GPU only")
    x = x.float().contiguous()
    if noise is None:
        noise = torch.randn_like(x)
    noise = noise.to(dev).float().contiguous()
    i_idx = i.reshape(x.shape[0], -1)[:, 0].to(torch.int32).contiguous()
    xn = torch.empty_like(x)
    ops.add_noise(x, noise, i_idx, alphas.to(dev), xn, None)
    return xn, noise


@ops.device_scope(lambda model, x, *a, **k: ops._first_cuda(x, model))
def get_loss(model, x, t, time_info, cond=None, target_mask=None, *, i=None, noise=None):
    """reference DIFF.py:252-260 (forward only; the training loop uses the fused
    loss+backward of the engine).  Returns a 0-dim device tensor."""
    B = x.shape[0]
    if i is None:
        i = torch.randint(0, diffusion_steps, size=(B,))
    eng = model.engine(x.device if x.is_cuda else None)
    dev = eng.device
    i_f = i.view(-1, 1, 1).expand_as(x[..., :1]).to(device=dev, dtype=torch.float32)
    x_noisy, noise = add_noise(x.to(dev), t, i_f, noise=noise)
    pred = model(x_noisy, t, i_f, time_info, cond, target_mask)
    loss_sum = torch.zeros(1, device=dev)
    ops.mse_loss(pred.contiguous(), noise, 0.0, None, loss_sum)
    return (loss_sum / pred.numel())[0]


# ---------------------------------------------------------------------------
def _stage(data, d):
    """Device-resident form of a per-window dataset: a contiguous fp32 [N, T, F] tensor or a WindowedTable."""
    from .dataprovider_pypots import WindowedTable
    if isinstance(data, WindowedTable):
        di = lambda x: x.index if x.index is not None else torch.cuda.current_device()     # noqa: E731
        if data.device.type != torch.device(d).type or di(data.device) != di(torch.device(d)):
            raise ValueError(f"WindowedTable lives on {data.device}, the trainer on {d}")
        return data
    return data.to(device=d, dtype=torch.float32).contiguous()


def _gather(data, idx, out):
    """Minibatch rows idx of a staged dataset (indices come from a draw over range(N): in range by construction)."""
    if isinstance(data, torch.Tensor):
        ops.gather_rows(data, idx, out)
    else:
        data.gather_unchecked(idx, out)


class DiffusionTrainer:
    """One-GPU-per-process training state of the denoiser: device-resident dataset, flat
    parameters/gradients/Adam moments, per-step draws, optional data-parallel plan."""

    @ops.device_scope(lambda self, model, *a, **k: k.get("dev") if k.get("dev") is not None else ops._first_cuda(model))
    def __init__(self, model: BiRNN_score, latent, time_info, cond, *, batch_size=None, lr=1e-3, dtype="fp32",
                 process_group=None, draws="reference", dev=None, graph=None):
        model.set_compute_dtype(dtype)
        # graph=True: everything of a step after the draws (gathers, add_noise, forward, loss, backward on both streams,
        # Adam) is captured once as a HIP graph and replayed -- one launch per step for the host instead of some
        # sixty.  Default: off.  With the step's launches as few and as cheap for the host as they are now, the device
        # is the bound even at the reference's batch of 100 windows, and the replay measures 0-4 % slower than the
        # launches one by one at 100 / 256 / 512 windows (scripts/graph_vs_eager.py); TAD_TRAIN_GRAPH=0/1 overrides.
        self._graph_opt = graph
        self._graph = None
        self.model = model
        self.eng = model.engine(dev)
        d = self.eng.device
        self.x = latent.detach().to(device=d, dtype=torch.float32).contiguous()
        # time_info / cond: the reference's stacked [N, T, F] tensors, or WindowedTable views of the tables they
        # are cut from (dataprovider_pypots.py:302-308) -- then the stacked form never exists
        self.time_info = _stage(time_info, d)
        self.cond = _stage(cond, d)
        self.N, self.T, self.K = self.x.shape
        # the batch gathers index all three with the same window numbers, unchecked on the device: a shorter
        # time_info / cond would be read out of bounds where the reference raises IndexError (DIFF.py:309-311)
        for nm, a in (("time_info", self.time_info), ("cond", self.cond)):
            if len(a) < self.N:
                raise IndexError(f"{nm} holds {len(a)} windows, latent {self.N}: index out of range")
            if a.shape[1] != self.T:
                raise ValueError(f"{nm} has T = {a.shape[1]}, latent T = {self.T}")
        self.B = batch_size if batch_size is not None else model.cfg["diffusion_steps"]
        self.lr = lr
        self.plan = DataParallelPlan(self.B, process_group)
        self.draws = draws
        self.alphas = alphas.to(d)
        # the loss partial sum lives right behind the gradients in the flat buffer: the last all-reduce of a step
        # carries it along
        self.loss_sum = self.eng.fp.slots(1)
        if draws == "device":
            # counter-based draws on the device: seeded like torch's generators (torch.manual_seed decides the run)
            self._draw_seed, self._draw_count, self._draw_bufs = torch.initial_seed(), 0, [None, None]
        elif draws != "reference":
            raise ValueError("draws must be 'reference' (python random + torch CPU generator, the reference's "
                             "order) or 'device' (one counter-based draw kernel on the device)")
        self.all_indices = list(range(self.N))
        nb = self.plan.local_count
        # the steps run in a workspace of the trainer's own: model(...), get_loss and sample() between two steps re-size
        # the engine's default workspace, and a captured step holds the addresses of the buffers it was captured on
        ws = self.ws = Workspace(d)
        self.bx = ws.get("batch.x", (nb, self.T, self.K))
        self.bt = ws.get("batch.time_info", (nb, self.T, self.time_info.shape[2]))
        self.bc = ws.get("batch.cond", (nb, self.T, self.cond.shape[2]))

    @ops.device_scope(lambda self, *a, **k: self.eng.device)
    def draw(self):
        """The random draws of one iteration for the GLOBAL batch, in the reference's order
        (DIFF.py:306, 317, 253, 108): indices, t, i, noise."""
        B, T, K = self.B, self.T, self.K
        d = self.eng.device
        if self.draws == "reference":
            idx = torch.tensor(random.sample(self.all_indices, B), dtype=torch.int64)
            t = torch.rand(B, T, 1).sort(1)[0]
            i = torch.randint(0, diffusion_steps, size=(B,))
            noise = torch.randn(B, T, K)
        else:
            if self.N < B:
                raise ValueError("Sample larger than population or is negative")
            # one launch (tad_draw_batch): a function of (seed, iteration) alone, so every rank draws the same global
            # batch; two sets of buffers alternate, so a draw made ahead does not overwrite the one a step still reads
            bufs = self._draw_bufs[self._draw_count & 1]
            if bufs is None:
                bufs = dict(idx=torch.empty(B, dtype=torch.int64, device=d), t=torch.empty(B, T, 1, device=d),
                            i=torch.empty(B, dtype=torch.int32, device=d), noise=torch.empty(B, T, K, device=d))
                self._draw_bufs[self._draw_count & 1] = bufs
            ops.draw_batch(self._draw_seed, self._draw_count, self.N, B, T, K, diffusion_steps, **bufs)
            self._draw_count += 1
            return bufs
        return dict(idx=idx, t=t, i=i, noise=noise)

    def use_graph(self):
        env = os.environ.get("TAD_TRAIN_GRAPH")
        if self.plan.world > 1 or ops.tracing() is not None:
            return False                       # collectives and per-launch timing events are not captured
        if self._graph_opt is not None:
            return bool(self._graph_opt)
        if env is not None:
            return env == "1"
        return False

    @ops.device_scope(lambda self, *a, **k: self.eng.device)
    def step(self, draw=None, loss_out=None):
        """One optimizer step (reference DIFF.py:306-325).  ``loss_out`` (1-element device
        tensor view) receives the mean loss of the global batch without a host sync."""
        with use_workspace(self.eng, self.ws):
            if self.use_graph():
                return self._step_graph(draw, loss_out)
            return self._step_eager(draw, loss_out)

    def _step_graph(self, draw, loss_out):
        """The step as a replay of its captured graph; the draws (a handful of launches) stay outside, in the order
        and from the generators of the eager path, and are copied into the buffers the graph reads."""
        eng = self.eng
        d = eng.device
        dr = draw if draw is not None else self.draw()
        if self._graph is None:
            # two eager steps first: they size the workspace, load the code objects and set the kernels' LDS limits
            # (none of which may happen inside a capture) -- on copies of the state, which is restored afterwards
            fp = eng.fp
            keep = (fp.flat.clone(), fp.m.clone(), fp.v.clone(), fp.step)
            for _ in range(2):
                self._step_eager(dr, None)
            fp.flat.copy_(keep[0]); fp.m.copy_(keep[1]); fp.v.copy_(keep[2]); fp.step = keep[3]
            fp.version += 1
            B, T, K = self.B, self.T, self.K
            self._g_in = dict(idx=torch.zeros(B, dtype=torch.int64, device=d), t=torch.zeros(B, T, 1, device=d),
                              i=torch.zeros(B, dtype=torch.int32, device=d), noise=torch.zeros(B, T, K, device=d))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body(self._g_in["idx"], self._g_in["t"], self._g_in["i"], self._g_in["noise"], captured=True)
            self._graph = g
        gi = self._g_in
        gi["idx"].copy_(dr["idx"], non_blocking=True)
        gi["t"].copy_(dr["t"], non_blocking=True)
        gi["i"].copy_(dr["i"], non_blocking=True)
        gi["noise"].copy_(dr["noise"], non_blocking=True)
        eng.fp.adam_before_replay()
        self._graph.replay()
        if loss_out is not None:
            torch.div(self.loss_sum, float(self.B * self.T * self.K), out=loss_out)

    def _step_body(self, idx, t, i_idx, noise, captured=False):
        """Gathers, zeroing, add_noise -> forward -> loss -> backward, Adam (single process)."""
        eng = self.eng
        ops.gather_rows(self.x, idx, self.bx)
        _gather(self.time_info, idx, self.bt)
        _gather(self.cond, idx, self.bc)
        eng.fp.zero_grad()                  # the loss slot with it
        eng.loss_and_grads(self.bx, t, i_idx, noise, self.bt, self.bc, self.alphas, self.loss_sum,
                           global_numel=self.B * self.T * self.K)
        if captured:
            eng.fp.adam_captured(self.lr)
        else:
            eng.fp.adam(self.lr)

    def _step_eager(self, draw=None, loss_out=None):
        eng, plan = self.eng, self.plan
        d = eng.device
        dr = draw if draw is not None else self.draw()
        lo, hi = plan.lo, plan.hi
        fp = eng.fp
        pending = []
        if hi == lo:
            # an empty shard (more ranks than windows in the global batch): nothing to compute, but this rank still
            # contributes zeros to both buckets' reductions, in the order every other rank issues them
            fp.zero_grad()
            pending.append(plan.all_reduce_async(fp.gflat[:eng.first_bucket_end]))
            pending.append(plan.all_reduce_async(fp.gflat[eng.first_bucket_end:fp.n_active + 1]))
            for w in pending:
                w.wait()
            fp.adam(self.lr)
            if loss_out is not None:
                torch.div(self.loss_sum, float(self.B * self.T * self.K), out=loss_out)
            return
        idx = dr["idx"][lo:hi].to(d, non_blocking=True)
        t = dr["t"][lo:hi].to(d, dtype=torch.float32, non_blocking=True).contiguous()
        i_idx = dr["i"][lo:hi].to(d, dtype=torch.int32, non_blocking=True).contiguous()
        noise = dr["noise"][lo:hi].to(d, dtype=torch.float32, non_blocking=True).contiguous()
        ops.gather_rows(self.x, idx, self.bx)
        _gather(self.time_info, idx, self.bt)
        _gather(self.cond, idx, self.bc)
        fp.zero_grad()                  # the loss slot with it
        if plan.world > 1:
            # two buckets: the head of the flat buffer (input_proj .. fc) is complete once the backward has passed
            # `proj`; its all-reduce starts there, ordered after the gradient stream's launches so far, and runs
            # beside the backward of the conditioning path.  The rest (and the loss slot) follows at the end.
            def first_bucket():
                fp.release()
                if fp.side is not None:
                    fp._fork_ev.record()
                    fp.side.wait_event(fp._fork_ev)          # LayerNorm's gradients are written on the main stream
                    with torch.cuda.stream(fp.side):
                        pending.append(plan.all_reduce_async(fp.gflat[:eng.first_bucket_end]))
                    fp._side_busy = True
                else:
                    pending.append(plan.all_reduce_async(fp.gflat[:eng.first_bucket_end]))
            eng.on_first_bucket = first_bucket
        try:
            eng.loss_and_grads(self.bx, t, i_idx, noise, self.bt, self.bc, self.alphas, self.loss_sum,
                               global_numel=self.B * self.T * self.K)
        finally:
            eng.on_first_bucket = None
        if plan.world > 1:
            pending.append(plan.all_reduce_async(fp.gflat[eng.first_bucket_end:fp.n_active + 1]))
            for w in pending:
                w.wait()
        eng.fp.adam(self.lr)
        if loss_out is not None:
            torch.div(self.loss_sum, float(self.B * self.T * self.K), out=loss_out)


def train_diffusion(latent_features, cond_df, data_dict, hidden_dim, num_layers, diffusion_steps, n_epochs,
                    num_classes=None, *, batch_size=None, lr=1e-3, dtype="fp32", process_group=None,
                    draws="reference", losses=None, progress=False, graph=None):
    """reference DIFF.py:267-330.  ``cond_df`` and ``num_classes`` are accepted and ignored like
    the reference.  Keyword-only extras: batch_size (default = diffusion_steps, the reference's
    hard-coded choice), lr (Adam default 1e-3), dtype ('fp32' | 'bf16'), process_group (data
    parallel over RCCL), draws ('reference' | 'device'), losses (a list that receives the
    per-iteration losses, fetched once at the end), graph (replay each step as one captured HIP graph; default: off,
    the launches one by one measure as fast or faster)."""
    emb_dim = 128
    info = data_dict["parser"].datatype_info()
    n_bins, n_cats, n_nums, cards = info["n_bins"], info["n_cats"], info["n_nums"], info["cards"]
    cond_train = data_dict["cond_train"]
    time_info = data_dict["time_info_train"]
    if diffusion_steps != globals()["diffusion_steps"]:
        raise ValueError("the schedule tables are built for 100 diffusion steps (reference DIFF.py:93-95)")
    input_size = latent_features.shape[2]
    time_dim = time_info.shape[2]
    org_feat_dim = cond_train.shape[2]
    model = BiRNN_score(input_size, org_feat_dim, hidden_dim, num_layers, diffusion_steps, time_dim, emb_dim,
                        n_bins, n_cats, n_nums, cards)
    trainer = DiffusionTrainer(model, latent_features, time_info, cond_train, batch_size=batch_size, lr=lr, dtype=dtype,
                               process_group=process_group, draws=draws, graph=graph)
    hist = torch.zeros(max(n_epochs, 1), device=trainer.eng.device)
    bar = None
    if progress:
        from rich.progress import Progress
        bar = Progress()
        bar.start()
        task = bar.add_task("[red]Training...", total=n_epochs)
    for epoch in range(n_epochs):
        trainer.step(loss_out=hist[epoch:epoch + 1])
        if bar is not None:
            bar.update(task, advance=1)
    if bar is not None:
        bar.stop()
    if losses is not None:
        losses.extend(hist[:n_epochs].tolist())
    ops.check_rnn_workspaces(trainer.eng.device)       # (synchronises; raises if a sequence launch gave up waiting)
    return model


# ---------------------------------------------------------------------------
@torch.no_grad()
@ops.device_scope(lambda t, B, T, F, diffusion_steps, model, *a, **k: ops._first_cuda(model, t))
def sample(t, B, T, F, diffusion_steps, model, time_info, cond, target_mask, *, noise=None, dtype=None,
           process_group=None, noise_source="reference", keep=None, streams=None, graph=None, local_only=False,
           noise_call=0):
    """reference DIFF.py:333-352: x_T ~ N(0,I), then 100 reverse steps with the step-0 double
    update.  Loop-invariant work (cond path, t_enc, time_encode and their share of `proj`) is
    computed once; the per-step i_enc contribution is a precomputed [steps, H] table.

    Keyword-only extras: noise = dict(x_T=[B,T,F], z=[steps,B,T,F]) to inject the draws;
    noise_source 'reference' (torch CPU generator in the reference's order, staged to the
    device once) or 'device' (counter-based draws: a function of (torch.initial_seed(), noise_call, step, global row),
    NOT of how many calls were made before -- two calls under one seed return the same samples, whatever the sharding,
    chunking or graph replay; pass noise_call = 1, 2, ... (or reseed) for fresh noise; the reference's generator
    advances from call to call, DIFF.py:335-347, which noise_source='reference' reproduces); dtype overrides the model's compute dtype; process_group shards
    the batch over ranks with no communication inside the loop; every rank returns the whole batch [B,T,F] (one
    all-gather of the finished slices) unless local_only=True (then its own slice); keep = dict that
    receives x after the listed steps (keys = step numbers); streams = number of contiguous chunks of
    the batch that walk the loop concurrently on HIP streams of their own (default 2 for batches of
    at least 512 sequences; no value depends on it); graph = True replays the loop as one
    HIP graph captured per batch shape (first call of a shape runs it twice) instead of enqueueing it kernel by kernel."""
    if diffusion_steps != globals()["diffusion_steps"]:
        raise ValueError("the schedule tables are built for 100 diffusion steps (reference DIFF.py:93-95)")
    if dtype is not None:
        model.set_compute_dtype(dtype)
    eng = model.engine()
    d = eng.device
    plan = DataParallelPlan(B, process_group)
    lo, hi = plan.lo, plan.hi
    nb = hi - lo
    S = diffusion_steps
    if noise is not None:
        x = noise["x_T"][lo:hi].to(d, dtype=torch.float32).contiguous().clone()
        z_all = noise["z"][:, lo:hi].to(d, dtype=torch.float32).contiguous()
    elif noise_source == "reference":
        x = torch.randn(B, T, F)[lo:hi].to(d).contiguous()
        z_cpu = torch.empty(S, nb, T, F)
        for s in range(S - 1, -1, -1):
            z_cpu[s] = torch.randn(B, T, F)[lo:hi]
        z_all = z_cpu.to(d)
    elif noise_source == "device":
        # counter-based noise on the device, seeded like torch's generators: x_T now (counter = S), each step's z when
        # the step runs, and only the rows this rank (and each of its concurrent chunks) owns -- never the
        # [S, B, T, F] block.  A function of (seed, step, global row): the result does not depend on the sharding.
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * int(noise_call)) & 0xFFFFFFFFFFFFFFFF
        x = torch.empty(nb, T, F, device=d)
        ops.sampler_noise(seed, S, lo * T, nb * T, F, x)
        z_all = CounterNoise(seed, S, lo * T)
    else:
        raise ValueError("noise_source must be 'reference' or 'device'")

    def prep(a, last):
        return a[lo:hi].to(device=d, dtype=torch.float32).reshape(nb, T, last).contiguous()
    if nb > 0:          # (a rank whose slice of the batch is empty only takes part in the gather below)
        eng.sample_loop(x, prep(t, 1), prep(time_info, time_info.shape[-1]), prep(cond, cond.shape[-1]), z_all,
                        betas, alphas, keep=keep, streams=streams, graph=graph)
    if eng.H == 512:
        ops.check_rnn_workspaces(d)       # the eight-workgroup form of the sequence kernel: raises if a launch gave up waiting
    if plan.world > 1 and not local_only:
        return plan.all_gather_cat(x)
    return x
