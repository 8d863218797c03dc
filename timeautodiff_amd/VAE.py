"""Drop-in for the reference's VAE.py hot path on MI355X.

``DeapStack`` (encoder / decoder / forward), ``auto_loss`` and the two trainers
``train_autoencoder`` / ``train_autoencoder_best_train`` keep the reference's names,
signatures and return values; the arithmetic runs in libtad_hip.so.  Parameters are
created by the same torch constructors in the same order as the reference
(VAE.py:114-145), so seeded initial weights and state_dict keys coincide.
"""
import random

import torch
import torch.nn as nn

from . import ops
from .DIFF import _EmbeddingData
from .dist import DataParallelPlan
from .engine import VAEEngine, Workspace, prec_of, use_workspace

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class DeapStack(nn.Module):
    """Mixed-type sequence VAE (reference VAE.py:111-220).  ``time_encode`` exists only because
    the reference constructs (and never uses) it."""

    def __init__(self, channels, n_bins, n_cats, n_nums, cards, input_size, hidden_size, num_layers, cat_emb_dim,
                 time_dim, lat_dim):
        super().__init__()
        self.cfg = dict(n_bins=n_bins, n_cats=n_cats, n_nums=n_nums, cards=list(cards), input_size=input_size,
                        hidden_size=hidden_size, num_layers=num_layers, emb_dim=cat_emb_dim, time_dim=time_dim,
                        lat_dim=lat_dim)
        self.Emb = _EmbeddingData(input_size, cat_emb_dim, n_bins, n_cats, n_nums, cards)
        self.time_encode = nn.Sequential(nn.Linear(time_dim, input_size), nn.ReLU(), nn.Linear(input_size, input_size))
        self.encoder_mu = nn.GRU(input_size, hidden_size, num_layers, batch_first=True)
        self.encoder_logvar = nn.GRU(input_size, hidden_size, num_layers, batch_first=True)
        self.fc_mu = nn.Linear(hidden_size, lat_dim)
        self.fc_logvar = nn.Linear(hidden_size, lat_dim)
        self.decoder_mlp = nn.Sequential(nn.Linear(lat_dim, hidden_size), nn.ReLU(), nn.Linear(hidden_size, hidden_size))
        self.Emb_decoder = _EmbeddingData(input_size, cat_emb_dim, n_bins, n_cats, n_nums, cards)
        self.Emb_hidden_decoder = nn.Linear(input_size, hidden_size)
        self.channels = channels
        self.n_bins, self.n_cats, self.n_nums, self.cards = n_bins, n_cats, n_nums, cards
        self.disc = n_bins + n_cats
        self.bins_linear = nn.Linear(hidden_size, n_bins) if n_bins else None
        self.cats_linears = nn.ModuleList([nn.Linear(hidden_size, card) for card in cards]) if n_cats else None
        self.nums_linear = nn.Linear(hidden_size, n_nums) if n_nums else None
        self._eng = None
        self._eng_dtype = "fp32"
        self.register_load_state_dict_post_hook(lambda m, keys: m.mark_params_changed())

    # -- engine plumbing ----------------------------------------------------
    def set_compute_dtype(self, dtype):
        prec_of(dtype)
        if dtype != self._eng_dtype:
            self._eng_dtype = dtype
            self._eng = None
        return self

    def engine(self, dev=None) -> VAEEngine:
        dev = torch.device(dev) if dev is not None else (self._eng.device if self._eng is not None else device)
        if dev.type == "cuda":
            dev = ops.norm_device(dev)              # 'cuda' and 'cuda:0' name the same engine
        if dev.type != "cuda":
            raise RuntimeError("DeapStack runs only on an MI355X through libtad_hip.so; no GPU is visible "
                               "(there is no CPU fallback)")
        if self._eng is None or self._eng.device != dev or not self._eng.fp.owns(self):
            self._eng = VAEEngine(self, dev, self._eng_dtype)
        return self._eng

    def mark_params_changed(self):
        if self._eng is not None:
            self._eng.fp.version += 1

    def _dev_f32(self, a, eng):
        return a.to(device=eng.device, dtype=torch.float32).contiguous()

    # -- reference API ------------------------------------------------------
    @ops.device_scope(lambda self, x, *a, **k: ops._first_cuda(x, self))
    def encoder(self, x, *, eps=None):
        """reference VAE.py:152-160 -> (emb, mu_z, logvar_z); the reparametrisation noise is drawn
        with torch.randn_like on the device unless ``eps`` is given."""
        eng = self.engine(x.device if x.is_cuda else None)
        eng.fp.version += 1
        x = self._dev_f32(x, eng)
        B, T, _ = x.shape
        if eps is None:
            eps = device_normals(B * T, eng.lat, eng.device).view(B, T, eng.lat)
        z, mu, lv = eng.encoder(x, self._dev_f32(eps, eng), train=False)
        return z.clone(), mu.clone(), lv.clone()

    @ops.device_scope(lambda self, x, *a, **k: ops._first_cuda(x, self))
    def decoder(self, latent_feature, mask=None, cond=None):
        """reference VAE.py:193-215 -> dict with 'bins' / 'cats' / 'nums' (present when that
        column type exists)."""
        eng = self.engine(latent_feature.device if latent_feature.is_cuda else None)
        eng.fp.version += 1
        z = self._dev_f32(latent_feature, eng)
        B, T, _ = z.shape
        heads = eng.decoder(z, self._dev_f32(mask, eng), self._dev_f32(cond, eng) if cond is not None else None)
        return eng.split_heads(heads, B, T)

    @ops.device_scope(lambda self, x, *a, **k: ops._first_cuda(x, self))
    def decode_to_tensor(self, latent_feature, mask=None, cond=None):
        """decoder (VAE.py:193-215) followed by convert_to_tensor (process_edited.py:252-301) -- what every
        driver does with sampled latents -- without materialising the dict: [B,T,F] float32 of thresholded
        bins, argmax codes and numericals."""
        eng = self.engine(latent_feature.device if latent_feature.is_cuda else None)
        eng.fp.version += 1
        z = self._dev_f32(latent_feature, eng)
        B, T, _ = z.shape
        heads = eng.decoder(z, self._dev_f32(mask, eng), self._dev_f32(cond, eng) if cond is not None else None)
        out = torch.empty(B, T, eng.F, dtype=torch.float32, device=eng.device)
        ops.heads_to_tensor(heads, eng.nb, eng.nc, eng.nn, eng.cards, out.view(B * T, eng.F))
        return out

    @ops.device_scope(lambda self, x, *a, **k: ops._first_cuda(x, self))
    def forward(self, x, mask, cond=None, *, eps=None):
        emb, mu_z, logvar_z = self.encoder(x, eps=eps)
        outputs = self.decoder(emb, mask, cond)
        return outputs, emb, mu_z, logvar_z


_EPS_DRAWS = {}      # torch seed -> number of reparametrisation draws made under it in this process


def device_normals(rows, cols, dev, row0=0, advance=True, out=None):
    """[rows, cols] standard normals drawn on the device by the counter-based kernel (tad_sampler_noise: Philox4x32-10,
    Box-Muller): a function of (torch.initial_seed(), the number of draws made so far under that seed, global row, column)
    -- the reparametrisation noise of reference VAE.py:147-150 (torch.randn_like on the model's device there).  Every
    rank of a data-parallel run makes the same sequence of draws and takes the rows it owns (``row0``): the noise of a
    window does not depend on the sharding.  advance=False re-uses the current draw number (further rows of one draw)."""
    seed = torch.initial_seed()
    k = _EPS_DRAWS.get(seed, 0)
    if advance:
        _EPS_DRAWS[seed] = k + 1
    elif k > 0:
        k -= 1
    if out is None:
        out = torch.empty(rows, cols, dtype=torch.float32, device=dev)
    if rows:
        ops.sampler_noise(seed, k, row0, rows, cols, out)
    return out


def init_vae_state(n_bins, n_cats, n_nums, cards, input_size, hidden_size, num_layers, emb_dim, time_dim, lat_dim):
    """state_dict (CPU tensors) of a freshly constructed DeapStack under the current torch seed."""
    m = DeapStack(1, n_bins, n_cats, n_nums, list(cards), input_size, hidden_size, num_layers, emb_dim, time_dim, lat_dim)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def auto_loss(inputs, reconstruction, n_bins, n_nums, n_cats, cards):
    """reference VAE.py:222-253 on already-decoded outputs -> (disc_loss, num_loss) device scalars.
    (The trainers use the fused head+loss kernel instead; this entry point exists for drivers
    that call auto_loss directly.)"""
    dev = inputs.device
    if dev.type != "cuda":
        raise RuntimeError("auto_loss runs on the GPU only")
    B, L, F = inputs.shape
    rows = B * L
    parts = torch.zeros(2 + max(n_cats, 1), device=dev)
    cols = []
    if "bins" in reconstruction:
        cols.append(reconstruction["bins"].reshape(rows, -1))
    if "cats" in reconstruction:
        cols += [c.reshape(rows, -1) for c in reconstruction["cats"]]
    if "nums" in reconstruction:
        cols.append(reconstruction["nums"].reshape(rows, -1))
    # the head outputs side by side in one [rows, NH] matrix (tad_copy_2d per head)
    heads = torch.empty(rows, sum(c.shape[1] for c in cols), dtype=torch.float32, device=dev)
    o = 0
    for c in cols:
        ops.copy_2d(c.to(device=dev, dtype=torch.float32).contiguous(), heads[:, o:o + c.shape[1]])
        o += c.shape[1]
    # the outputs already carry the mask offsets / sigmoid: evaluate the loss terms directly
    ones = torch.empty(rows, F, device=dev)
    ops.fill_(ones, 1.0)
    nb = n_bins if "bins" in reconstruction else 0
    nc = n_cats if "cats" in reconstruction else 0
    disc = torch.zeros((), device=dev)
    num = torch.zeros((), device=dev)
    tgt = inputs.reshape(rows, F).float().contiguous()
    if nb or nc:
        sub = heads[:, :nb + sum(cards[:nc])].contiguous()
        ops.vae_heads_loss(sub, ones, tgt, nb, nc, 0, list(cards[:nc]), 0, 0, 0, None, parts)
        if nb:
            disc = disc + parts[0] / (rows * nb)
        if nc:
            disc = disc + parts[2:2 + nc].sum() / (rows * nc)
    if "nums" in reconstruction:
        nums = heads[:, heads.shape[1] - n_nums:].contiguous()
        t = tgt[:, n_bins + n_cats:n_bins + n_cats + n_nums].contiguous()
        s = torch.zeros(1, device=dev)
        ops.mse_loss(nums, t, 0.0, None, s)
        num = s[0] / nums.numel()
    return disc, num


# ---------------------------------------------------------------------------
class VAETrainer:
    """Device-resident training state of the VAE (one process per GPU)."""

    @ops.device_scope(lambda self, ae, *a, **k: k.get("dev") if k.get("dev") is not None else ops._first_cuda(ae))
    def __init__(self, ae: DeapStack, data_dict, *, batch_size, lr, weight_decay, dtype="fp32", process_group=None, dev=None,
                 graph=None):
        ae.set_compute_dtype(dtype)
        self.ae = ae
        self.eng = eng = ae.engine(dev)
        d = eng.device
        f32 = dict(device=d, dtype=torch.float32)
        from .dataprovider_pypots import WindowedTable
        from .DIFF import _stage

        def n(k, keep_windows=False):
            v = data_dict.get(k)
            if v is None:
                return None
            if isinstance(v, WindowedTable):
                # training response / cond may stay un-materialised (windows are cut from the table per minibatch);
                # everything else is small or needed whole
                return _stage(v, d) if keep_windows else _stage(v, d).materialize()
            return v.to(**f32).contiguous()
        self.resp_tr, self.mask_tr, self.cond_tr = n("response_train", True), n("target_mask_train"), n("cond_train", True)
        self.resp_va, self.mask_va, self.cond_va = n("response_val"), n("target_mask_val"), n("cond_val")
        self.windowed = isinstance(self.resp_tr, WindowedTable)
        if self.cond_tr is not None and isinstance(self.cond_tr, WindowedTable) != self.windowed:
            raise ValueError("response_train and cond_train must both be tensors or both be WindowedTable views")
        self.N, self.T, self.F = self.resp_tr.shape
        if tuple(self.mask_tr.shape) != (self.N, self.T, self.F):
            raise ValueError(f"target_mask_train has shape {tuple(self.mask_tr.shape)}, response_train {(self.N, self.T, self.F)}")
        self.B = batch_size
        self.lr, self.wd = lr, weight_decay
        self.plan = DataParallelPlan(batch_size, process_group)
        # inp = response + cond, target = inp * mask (reference VAE.py:331-336): once for the whole stacked set, or per
        # minibatch when the set is a WindowedTable (the stacked form then never exists)
        if self.windowed:
            self.inp_tr = self.tgt_tr = None
        else:
            self.inp_tr = torch.empty_like(self.resp_tr)
            self.tgt_tr = torch.empty_like(self.resp_tr)
            ops.vae_prepare(self.resp_tr, self.cond_tr, self.mask_tr, self.inp_tr, self.tgt_tr)
        self.inp_va = torch.empty_like(self.resp_va)
        self.tgt_va = torch.empty_like(self.resp_va)
        ops.vae_prepare(self.resp_va, self.cond_va, self.mask_va, self.inp_va, self.tgt_va)
        self.Nv = self.resp_va.shape[0]
        self.vplan = DataParallelPlan(self.Nv, process_group)
        # [bce, mse, ce_0.., kld] partial sums: right behind the gradients in the flat buffer when they fit there, so
        # that one all-reduce carries gradients and loss terms together
        npart = 3 + max(eng.nc, 1)
        self.parts_in_flat = npart <= eng.fp.SLOTS
        self.parts = eng.fp.slots(npart) if self.parts_in_flat else torch.zeros(npart, device=d)
        nb = self.plan.local_count
        # training steps and validation passes run in workspaces of the trainer's own (they work on batches of different
        # sizes, and so do ae.encoder / ae.decoder / full_forward through the engine's default workspace): no pass
        # re-allocates another's buffers, and captured graphs keep the addresses they were captured on
        ws = self._ws_train = Workspace(d)
        self._ws_eval = Workspace(d)
        self.b_inp = ws.get("batch.inp", (nb, self.T, self.F))
        self.b_tgt = ws.get("batch.tgt", (nb, self.T, self.F))
        self.b_mask = ws.get("batch.mask", (nb, self.T, self.F))
        self.b_cond = ws.get("batch.cond", (nb, self.T, self.F)) if self.cond_tr is not None else None
        self.b_resp = ws.get("batch.resp", (nb, self.T, self.F)) if self.windowed else None
        self.all_indices = list(range(self.N))
        self._next_idx = None
        # graph=True: a training step (after its draws) and a validation pass are captured as HIP graphs and replayed
        # (one graph per value of beta, which the kernels take as a launch argument).  Default: on for batches of at
        # most 512 windows in a single process; TAD_TRAIN_GRAPH=0/1 overrides.
        self._graph_opt = graph
        self._graphs = {}
        self._g_in = None
        self._eps_bufs = {}

    def use_graph(self):
        import os
        env = os.environ.get("TAD_TRAIN_GRAPH")
        if self.plan.world > 1 or ops.tracing() is not None:
            return False
        if self._graph_opt is not None:
            return bool(self._graph_opt)
        if env is not None:
            return env == "1"
        return self.B <= 512

    def _train_body(self, idx_d, eps_l, beta, captured=False):
        with use_workspace(self.eng, self._ws_train):
            self._train_body_ws(idx_d, eps_l, beta, captured)

    def _train_body_ws(self, idx_d, eps_l, beta, captured):
        eng = self.eng
        if idx_d.numel() == 0:
            # an empty shard of the global batch: zeros into the reduction, then the same optimizer step as every rank
            eng.fp.zero_grad()
            if not self.parts_in_flat:
                ops.zero_(self.parts)
            if self.parts_in_flat:
                self.plan.all_reduce(eng.fp.gflat[:eng.fp.n_active + self.parts.numel()])
            else:
                self.plan.all_reduce(eng.fp.gflat[:eng.fp.n_active], self.parts)
            eng.fp.adam(self.lr, weight_decay=self.wd, upto=None if self.cond_tr is not None else eng.n_without_cond)
            return
        ops.gather_rows(self.mask_tr, idx_d, self.b_mask)
        if self.windowed:
            self.resp_tr.gather_unchecked(idx_d, self.b_resp)          # indices come from a draw over range(N)
            if self.b_cond is not None:
                self.cond_tr.gather_unchecked(idx_d, self.b_cond)
            ops.vae_prepare(self.b_resp, self.b_cond, self.b_mask, self.b_inp, self.b_tgt)
        else:
            ops.gather_rows(self.inp_tr, idx_d, self.b_inp)
            ops.gather_rows(self.tgt_tr, idx_d, self.b_tgt)
            if self.b_cond is not None:
                ops.gather_rows(self.cond_tr, idx_d, self.b_cond)
        rows = self.B * self.T
        eng.fp.zero_grad()                 # clears the loss parts with the gradients when they live in the flat buffer
        if not self.parts_in_flat:
            ops.zero_(self.parts)
        nc = eng.nc
        kld = self.parts[2 + max(nc, 1):]
        z, mu, lv = eng.encoder(self.b_inp, eps_l, train=True, kld_sum=kld)
        eng.decoder(z, self.b_mask, self.b_cond, target=self.b_tgt, scales=eng.loss_scales(rows), loss_parts=self.parts,
                    train=True)
        eng.backward(beta, rows)
        upto = None if self.cond_tr is not None else eng.n_without_cond
        if captured:
            eng.fp.adam_captured(self.lr, weight_decay=self.wd, upto=upto)
        else:
            plan = self.plan
            if self.parts_in_flat:
                plan.all_reduce(eng.fp.gflat[:eng.fp.n_active + self.parts.numel()])
            else:
                plan.all_reduce(eng.fp.gflat[:eng.fp.n_active], self.parts)
            eng.fp.adam(self.lr, weight_decay=self.wd, upto=upto)

    def _eval_body(self, eps_l, lo, hi):
        eng = self.eng
        ops.zero_(self.parts)
        nc = eng.nc
        kld = self.parts[2 + max(nc, 1):]
        if hi == lo:
            return                              # an empty shard of the validation set adds nothing to the sums
        with use_workspace(eng, self._ws_eval):
            z, mu, lv = eng.encoder(self.inp_va[lo:hi], eps_l, train=False, kld_sum=kld)
            cond = self.cond_va[lo:hi] if self.cond_va is not None else None
            eng.decoder(z, self.mask_va[lo:hi], cond, target=self.tgt_va[lo:hi], loss_parts=self.parts, train=False)

    def _graph_for(self, key, build):
        """The captured graph for ``key`` (built on first use, after two eager runs on saved state)."""
        g = self._graphs.get(key)
        if g is None:
            fp = self.eng.fp
            keep = (fp.flat.clone(), fp.m.clone(), fp.v.clone(), fp.step)
            for _ in range(2):
                build(False)
            fp.flat.copy_(keep[0]); fp.m.copy_(keep[1]); fp.v.copy_(keep[2]); fp.step = keep[3]
            fp.version += 1
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                build(True)
            self._graphs[key] = g
        return g

    def _eps_buf(self, tag, n):
        """Persistent buffer for a pass's reparametrisation noise (no allocation per step)."""
        key = (tag, n)
        if key not in self._eps_bufs:
            self._eps_bufs[key] = torch.empty(n * self.T, self.eng.lat, dtype=torch.float32, device=self.eng.device)
        return self._eps_bufs[key]

    def _loss(self, rows, beta):
        """Combine the (all-reduced) partial sums on the host -> python float total."""
        p = self.parts.tolist()
        # (the read-back above has synchronised: every 16th time also ask whether a sequence launch whose workgroups wait
        # for each other -- hidden size 512 in bf16 -- gave up waiting)
        self._n_loss = getattr(self, "_n_loss", 0) + 1
        if self._n_loss % 16 == 1:
            ops.check_rnn_workspaces(self.eng.device)
        nc = self.eng.nc
        total, _ = self.eng.total_loss(p[:2 + nc], p[2 + max(nc, 1)], beta, rows)
        return total

    @ops.device_scope(lambda self, *a, **k: self.eng.device)
    def train_step(self, beta, idx=None, eps=None, prefetch=True):
        """One optimizer step (reference VAE.py:322-354) -> train loss (python float).  With idx=None the
        minibatch comes from ``random.sample`` like the reference's; prefetch=False on the last step of a run keeps
        the state of python's ``random`` where the reference leaves it (no draw for a step that never happens)."""
        eng, plan = self.eng, self.plan
        d = eng.device
        lo, hi = plan.lo, plan.hi
        own_draw = idx is None
        if own_draw:
            # drawn while the previous step was still running on the device (see the end of this function)
            idx_d = self._next_idx if self._next_idx is not None else self._draw_idx()
            self._next_idx = None
        else:
            idx_d = torch.as_tensor(idx[lo:hi], dtype=torch.int64).to(d)
            if idx_d.numel() and (int(idx_d.min()) < 0 or int(idx_d.max()) >= self.N):
                raise IndexError("index out of range")          # the gathers below do not check
        if eps is None:
            # this rank's rows of the global batch's noise, drawn on the device
            eps_l = device_normals((hi - lo) * self.T, eng.lat, d, row0=lo * self.T, out=self._eps_buf("train", hi - lo)).view(hi - lo, self.T, eng.lat)
        else:
            eps_l = eps[lo:hi].to(device=d, dtype=torch.float32).contiguous()
        rows = self.B * self.T
        if self.use_graph():
            if self._g_in is None:
                self._g_in = dict(idx=torch.zeros(hi - lo, dtype=torch.int64, device=d), eps=torch.zeros(hi - lo, self.T, eng.lat, device=d),
                                  veps=torch.zeros(self.vplan.hi - self.vplan.lo, self.T, eng.lat, device=d))
            gi = self._g_in
            gi["idx"].copy_(idx_d, non_blocking=True)
            gi["eps"].copy_(eps_l, non_blocking=True)
            g = self._graph_for(("train", float(beta)), lambda cap: self._train_body(gi["idx"], gi["eps"], beta, captured=cap))
            eng.fp.adam_before_replay()
            g.replay()
        else:
            self._train_body(idx_d, eps_l, beta)
        if own_draw and prefetch:
            # the next minibatch's indices (python `random`, the reference's generator for them, VAE.py:340) are
            # drawn now, while the device works on this step: reading the loss back below blocks the host
            self._next_idx = self._draw_idx()
        return self._loss(rows, beta)

    def _draw_idx(self):
        """random.sample of the global batch (reference VAE.py:340) -> this rank's slice, on the device."""
        idx = random.sample(self.all_indices, self.B)
        return torch.as_tensor(idx[self.plan.lo:self.plan.hi], dtype=torch.int64).to(self.eng.device)

    @ops.device_scope(lambda self, *a, **k: self.eng.device)
    def eval_loss(self, beta, eps=None):
        """Full validation-set loss under the current weights (reference VAE.py:357-368)."""
        eng, plan = self.eng, self.vplan
        d = eng.device
        lo, hi = plan.lo, plan.hi
        if eps is None:
            eps_l = device_normals((hi - lo) * self.T, eng.lat, d, row0=lo * self.T, out=self._eps_buf("val", hi - lo)).view(hi - lo, self.T, eng.lat)
        else:
            eps_l = eps[lo:hi].to(device=d, dtype=torch.float32).contiguous()
        if self.use_graph() and self._g_in is not None:
            gi = self._g_in
            gi["veps"].copy_(eps_l, non_blocking=True)
            self._graph_for("eval", lambda cap: self._eval_body(gi["veps"], lo, hi)).replay()
        else:
            self._eval_body(eps_l, lo, hi)
            plan.all_reduce(self.parts)
        return self._loss(self.Nv * self.T, beta)

    @ops.device_scope(lambda self, *a, **k: self.eng.device)
    def full_forward(self, eps=None):
        """Forward of the whole training set (reference VAE.py:395-402) -> (output dict, latent)."""
        eng = self.eng
        d = eng.device
        if eps is None:
            eps = device_normals(self.N * self.T, eng.lat, d).view(self.N, self.T, eng.lat)
        inp, cond = self.inp_tr, self.cond_tr
        if self.windowed:               # the one place that needs the whole training set at once (reference VAE.py:395-402)
            cond = self.cond_tr.materialize() if self.cond_tr is not None else None
            inp = torch.empty(self.N, self.T, self.F, device=d)
            ops.vae_prepare(self.resp_tr.materialize(), cond, self.mask_tr, inp, torch.empty_like(inp))
        z, mu, lv = eng.encoder(inp, eps.to(device=d, dtype=torch.float32).contiguous(), train=False)
        latent = z.clone()
        heads = eng.decoder(latent, self.mask_tr, cond)
        return eng.split_heads(heads, self.N, self.T), latent


def _train(real_df, channels, hidden_size, num_layers, lr, weight_decay, n_epochs, batch_size, threshold, min_beta,
           max_beta, emb_dim, time_dim, lat_dim, device, data_dict, select, dtype, process_group, noise_source,
           progress, graph=None):
    info = data_dict["parser"].datatype_info()
    n_bins, n_cats, n_nums, cards = info["n_bins"], info["n_cats"], info["n_nums"], info["cards"]
    N, _, input_size = data_dict["response_train"].shape
    ae = DeapStack(channels, n_bins, n_cats, n_nums, cards, input_size, hidden_size, num_layers, emb_dim, time_dim, lat_dim)
    dev = torch.device(device) if device is not None else None
    if dev is not None and dev.type != "cuda":
        raise RuntimeError("train_autoencoder runs only on an MI355X through libtad_hip.so (device=%r)" % (device,))
    tr = VAETrainer(ae, data_dict, batch_size=batch_size, lr=lr, weight_decay=weight_decay, dtype=dtype,
                    process_group=process_group, dev=dev, graph=graph)
    eng = tr.eng
    d = eng.device
    T, L = tr.T, eng.lat

    def draw(n):
        # "reference": the CPU generator, in the order the reference draws when it runs on a CPU.  "device" (what the
        # reference's randn_like does on a GPU): None -- each pass draws its rows with the counter-based kernel
        if noise_source == "reference":
            return torch.randn(n, T, L)
        if noise_source != "device":
            raise ValueError("noise_source must be 'reference' or 'device'")
        return None

    beta = max_beta
    patience, lambd = 0, 0.7
    best_val, best_train = float("inf"), float("inf")
    best_flat = eng.fp.flat.clone()
    history = {"train": [], "val": []}
    bar = None
    if progress:
        from rich.progress import Progress
        bar = Progress()
        bar.start()
        task = bar.add_task("[red]Training", total=n_epochs)
    for epoch in range(n_epochs):
        # indices: random.sample inside train_step (reference VAE.py:340), drawn one step ahead while the device works
        train_loss = tr.train_step(beta, eps=draw(batch_size), prefetch=epoch + 1 < n_epochs)
        if select == "train" and train_loss < best_train:
            best_train = train_loss
            best_flat.copy_(eng.fp.flat)
        val_loss = tr.eval_loss(beta, eps=draw(tr.Nv))
        if val_loss < best_val:
            best_val = val_loss
            if select == "val":
                best_flat.copy_(eng.fp.flat)
            patience = 0
        else:
            patience += 1
            if patience == 10 and beta > min_beta:
                beta *= lambd
                patience = 0
        history["train"].append(train_loss)
        history["val"].append(val_loss)
        if bar is not None:
            bar.update(task, advance=1)
    if bar is not None:
        bar.stop()
    eng.fp.flat.copy_(best_flat)
    eng.fp.version += 1
    ae.eval()
    output, latent = tr.full_forward(eps=draw(N))
    return ae, latent.detach(), output, history


def train_autoencoder(real_df, channels, hidden_size, num_layers, lr, weight_decay, n_epochs, batch_size, threshold,
                      min_beta, max_beta, emb_dim, time_dim, lat_dim, device, data_dict, *, dtype="fp32",
                      process_group=None, noise_source="device", progress=False, graph=None):
    """reference VAE.py:261-404: weights selected by lowest validation loss.  ``real_df``,
    ``channels`` and ``threshold`` are accepted and ignored like the reference.  Returns
    (ae, latent (sampled z of the training set under the best weights), output dict, history)."""
    return _train(real_df, channels, hidden_size, num_layers, lr, weight_decay, n_epochs, batch_size, threshold,
                  min_beta, max_beta, emb_dim, time_dim, lat_dim, device, data_dict, "val", dtype, process_group,
                  noise_source, progress, graph)


def train_autoencoder_best_train(real_df, channels, hidden_size, num_layers, lr, weight_decay, n_epochs, batch_size,
                                 threshold, min_beta, max_beta, emb_dim, time_dim, lat_dim, device, data_dict, *,
                                 dtype="fp32", process_group=None, noise_source="device", progress=False, graph=None):
    """reference VAE.py:416-568: weights selected by lowest *training* loss."""
    return _train(real_df, channels, hidden_size, num_layers, lr, weight_decay, n_epochs, batch_size, threshold,
                  min_beta, max_beta, emb_dim, time_dim, lat_dim, device, data_dict, "train", dtype, process_group,
                  noise_source, progress, graph)
