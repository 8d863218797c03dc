"""Host-side orchestration of the HIP kernels: flat parameter storage, workspaces and
the forward/backward launch sequences of the denoiser (BiRNN_score) and the VAE
(DeapStack).  All arithmetic happens in libtad_hip.so; torch only owns the memory."""
import math
import os
from typing import List, Optional

import torch

from . import ops
from .ops import ACT_DSILU, ACT_RELU, ACT_SILU_Z32, BF16, CELL_GRU, CELL_LSTM, F32

ALIGN = 64  # floats (256 B)
# TAD_GRAD_STREAM=1 moves the launches nothing waits for before the optimizer step (weight gradients, table gradients,
# weight packing) to a second stream (2: released only beside the next sequence kernel).  Off by default: since the
# weight gradients of a layer are one chip-filling launch the second stream measured within 1 % either way (denoiser
# 6.83 vs 6.79 ms, VAE at hidden 512 10.88 vs 11.18 ms) and on one stream every kernel's duration is its own.
GRAD_STREAM = os.environ.get("TAD_GRAD_STREAM", "0") != "0"
# chunks of the sampling batch that walk the reverse loop concurrently (1 = the whole batch on the current stream)
SAMPLE_STREAMS = int(os.environ.get("TAD_SAMPLE_STREAMS", "2"))
# TAD_SAMPLE_FUSED_TAIL=0: LayerNorm, fc, the sampler update and the next input_proj as four launches instead of one
SAMPLE_FUSED_TAIL = os.environ.get("TAD_SAMPLE_FUSED_TAIL", "1") != "0"
# the x share of `proj` folded into the tail of the previous reverse step (bf16 mode): one product and one launch fewer per step
SAMPLE_FOLD = os.environ.get("TAD_SAMPLE_FOLD", "1") != "0"
# counter-based noise of a reverse step drawn inside the folded tail instead of by a launch of its own (experiments: 0)
SAMPLE_TAIL_NOISE = os.environ.get("TAD_SAMPLE_TAIL_NOISE", "1") != "0"
# the denoiser keeps its product-only tensors as bf16 up to this 4 H (products with K > 1024 take the square-tiled kernel)
MM16_MAX_4H = int(os.environ.get("TAD_MM16_MAX_4H", "4096"))
# the VAE decoder's products on bf16-stored operands (bf16 mode)
VAE_DEC16 = os.environ.get("TAD_VAE_DEC16", "1") != "0"
# a first recurrent layer of at most 32 input columns at hidden size 512 (bf16 mode): x_t W_ih^T inside the sequence kernel
RNN_XIN = os.environ.get("TAD_RNN_XIN", "1") != "0"
# input-side products of a recurrent layer whose weight has at least this many elements take the square-tiled kernel
XPROJ_TILED_MIN = int(os.environ.get("TAD_XPROJ_TILED_MIN", str(1 << 21)))
# TAD_SAMPLE_GRAPH=1: replay the reverse loop as one captured HIP graph per batch shape instead of enqueueing it kernel
# by kernel (off by default: at 1k sequences the loop is device-bound and the graph measured no faster)
SAMPLE_GRAPH = os.environ.get("TAD_SAMPLE_GRAPH", "0") == "1"
GRAD_IMMEDIATE = os.environ.get("TAD_GRAD_STREAM", "0") == "1"
# HIP priority of the gradient stream (larger = lower priority): what runs there is off the critical path and must not
# take compute units from the main stream's kernels
SIDE_PRIORITY = int(os.environ.get("TAD_SIDE_PRIORITY", "0"))
# TAD_MM16=0: bf16 precision keeps every activation in fp32 and uses the fp32-operand product kernels throughout
# (the path of the fp32 mode with operands rounded inside the kernels; experiments and A/B timing)
MM16 = os.environ.get("TAD_MM16", "1") != "0"
# TAD_LN_FC_FUSED=0: layer_norm and fc of the denoiser as separate launches (forward and backward)
LN_FC_FUSED = os.environ.get("TAD_LN_FC_FUSED", "1") != "0"


def prec_of(dtype):
    """'fp32'/torch.float32 -> F32 (exact fp32 MFMA), 'bf16'/torch.bfloat16 -> BF16 operands."""
    if dtype in (F32, "fp32", "f32", torch.float32, None):
        return F32
    if dtype in (BF16, "bf16", torch.bfloat16):
        return BF16
    raise ValueError(f"unsupported compute dtype {dtype!r} (use 'fp32' or 'bf16')")


class FlatParams:
    """All parameters of a module in ONE contiguous fp32 device buffer (plus same-shaped
    gradient and Adam moment buffers).  The module's nn.Parameters are re-pointed to
    views of it, so state_dict()/load_state_dict() keep working with the reference's key
    names, while the kernels, the fused Adam and the gradient all-reduce address flat
    ranges.  ``layout`` is a list of (name, glue) in memory order; glue=True places the
    tensor directly after the previous one (groups the kernels read as one matrix).
    Names listed in ``inactive`` (parameters the forward never uses, which therefore
    never get a gradient in the reference) are stored after the active range and are
    excluded from the optimizer and the all-reduce."""

    SLOTS = 16

    def __init__(self, module, layout, inactive, device):
        params = dict(module.named_parameters())
        names = [n for n, _ in layout] + list(inactive)
        assert sorted(names) == sorted(params), (set(names) ^ set(params))
        self.offsets = {}
        off = 0
        for name, glue in layout:
            if not glue:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.offsets[name] = off
            off += params[name].numel()
        self.n_active = off
        # SLOTS floats behind the active gradients hold the loss partial sums of a step: one all-reduce then carries
        # gradients and loss together (the same floats of ``flat`` are unused padding)
        off = (off + self.SLOTS + ALIGN - 1) // ALIGN * ALIGN
        for name in inactive:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.offsets[name] = off
            off += params[name].numel()
        self.total = off
        self.device = device
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.gflat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.m = torch.zeros(self.n_active, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.n_active, dtype=torch.float32, device=device)
        self.step = 0
        # device-side step counter and bias-correction factors of captured optimizer steps (see adam_state); allocated
        # here, never inside a capture (a captured allocation would be re-zeroed by every replay)
        self._coef = torch.zeros(4, dtype=torch.float32, device=device)
        self._dev_step = 0
        self.shapes = {n: tuple(p.shape) for n, p in params.items()}
        self.version = 0          # bumped whenever parameter values change
        # weight-gradient products are off the critical path of a backward pass (nothing downstream reads them
        # before the optimizer step): they go to a second HIP stream and fill the compute units that the
        # latency-bound sequence kernels of the main stream leave idle
        self.side = torch.cuda.Stream(device=device, priority=SIDE_PRIORITY) if (GRAD_STREAM and torch.device(device).type == "cuda") else None
        self._fork_ev = torch.cuda.Event() if self.side is not None else None
        self._join_ev = torch.cuda.Event() if self.side is not None else None
        self._side_busy = False
        self._queued = []
        self._tn_jobs = []
        for name, p in params.items():
            view = self.p(name)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = self.g(name) if self.offsets[name] < self.n_active else None
        self._params = params

    def owns(self, module):
        """True while the module's parameters still alias this buffer (a .to()/.cuda()
        or a manual re-assignment breaks the aliasing)."""
        for name, p in module.named_parameters():
            if p.data_ptr() != self.flat.data_ptr() + 4 * self.offsets[name]:
                return False
        return True

    def _view(self, buf, name, n=None, shape=None):
        off = self.offsets[name]
        if n is None:
            shape = self.shapes[name]
            n = math.prod(shape)
        return buf[off:off + n].view(shape)

    def p(self, name):
        return self._view(self.flat, name)

    def g(self, name):
        return self._view(self.gflat, name)

    def pgroup(self, first, rows, cols):
        """A [rows, cols] matrix view starting at tensor ``first`` (a glued group)."""
        return self._view(self.flat, first, rows * cols, (rows, cols))

    def ggroup(self, first, rows, cols):
        return self._view(self.gflat, first, rows * cols, (rows, cols))

    def zero_grad(self):
        ops.zero_(self.gflat[:self.n_active + self.SLOTS])

    def slots(self, n=1):
        """n floats right behind the active gradient range (zeroed by zero_grad, reduced with the gradients)."""
        assert n <= self.SLOTS
        return self.gflat[self.n_active:self.n_active + n]

    def defer(self, fn, *args, **kwargs):
        """Queue a launch whose only result is an accumulation into the gradient buffer.  It is issued on the
        side stream at the next ``release()`` (ordered after everything enqueued on the current stream by then).
        ``join()`` must precede any reader of the gradients and any writer of the launch's operands."""
        if self.side is None:
            return fn(*args, **kwargs)
        self._queued.append((fn, args, kwargs))
        if GRAD_IMMEDIATE:
            self.release()

    def defer_until_release(self, fn, *args, **kwargs):
        """Queue any launch that is independent of what the current stream does until the next ``join()``; it is
        issued on the side stream at the next ``release()`` (i.e. beside the next sequence kernel)."""
        if self.side is None:
            return fn(*args, **kwargs)
        self._queued.append((fn, args, kwargs))

    def queue_tn(self, a16, job):
        """Queue one small weight-gradient product (ops.mm16_tn job with its own A operand).  The products of a
        backward pass that are queued this way leave together at ``flush_tn`` -- one launch per twelve jobs, in which
        every job gets a share of the chip and a dozen reduction splits instead of a launch of its own with hundreds."""
        j = dict(job)
        j["a"] = a16
        self._tn_jobs.append(j)

    def flush_tn(self):
        """Launch the queued weight-gradient products (deferred to the gradient stream like every weight gradient)."""
        jobs, self._tn_jobs = self._tn_jobs, []
        for i in range(0, len(jobs), ops.TN_MAXJOBS):
            self.defer(ops.mm16_tn, None, jobs[i:i + ops.TN_MAXJOBS])

    def release(self):
        """Issue the queued launches on the side stream; called right before a latency-bound kernel goes to the
        current stream, so that the two run side by side."""
        if not self._queued:
            return
        self._fork_ev.record()
        self.side.wait_event(self._fork_ev)
        with torch.cuda.stream(self.side):
            for fn, args, kwargs in self._queued:
                fn(*args, **kwargs)
        self._queued.clear()
        self._side_busy = True

    def join(self):
        """Issue what is still queued; the current stream then waits for everything deferred so far."""
        if self._tn_jobs:
            self.flush_tn()
        self.release()
        if self._side_busy:
            self._join_ev.record(self.side)
            torch.cuda.current_stream().wait_event(self._join_ev)
            self._side_busy = False

    def adam_state(self):
        """Device-side optimizer state of captured steps: float32[4] = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step),
        the step counter (int32 bits), unused}; ``_dev_step`` mirrors the counter on the host."""
        return self._coef

    def adam_captured(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, upto=None):
        """The launches of ``adam`` for a HIP-graph capture: a single-thread launch advances the step counter kept on the
        device and refreshes the two step-dependent factors, the Adam launch reads them from there -- a replay needs no
        host-to-device copy (``adam_replayed`` does the host-side bookkeeping)."""
        coef = self.adam_state()
        n = self.n_active if upto is None else min(int(upto), self.n_active)
        ops.adam_tick(coef, lr, betas[0], betas[1])
        ops.adam_step_dev(self.flat[:n], self.gflat[:n], self.m[:n], self.v[:n], betas[0], betas[1], eps, weight_decay, coef)

    def adam_before_replay(self):
        """Before a replay of a captured step: the device's step counter must equal the host's (eager steps in between,
        or a restored state, move the host's only) -- re-seeded with one small copy when it does not."""
        coef = self.adam_state()
        if self._dev_step != self.step:
            coef[2:3].view(torch.int32).copy_(torch.tensor([self.step], dtype=torch.int32), non_blocking=False)
            self._dev_step = self.step
        self.step += 1
        self._dev_step += 1
        self.version += 1

    def adam(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, upto=None):
        """One Adam step over the active range, or over its first ``upto`` elements (the rest holds parameters
        that received no gradient in this step: torch.optim.Adam skips those, weight decay included)."""
        self.step += 1
        n = self.n_active if upto is None else min(int(upto), self.n_active)
        ops.adam_step(self.flat[:n], self.gflat[:n], self.m[:n], self.v[:n], lr, betas[0], betas[1], eps, weight_decay, self.step)
        self.version += 1


class W16Images:
    """bf16 images of weight matrices for the bf16-stored-operand products (ops.mm16_nt): ``fwd`` = W [n, k] for
    y = x W^T, ``tr`` = W^T [k, n] for dx = dy W.  Storage is zero-initialised and padded (rows to a multiple of 64
    plus one spare block of 64, so that a row slice of an image is an image too; row stride to a multiple of 128);
    ``pack()`` refreshes all images in one launch when the parameters changed."""

    def __init__(self, fp):
        self.fp = fp
        self.items = []
        self.sums = []
        self.copies = []
        self._ver = -1

    @staticmethod
    def alloc(n, k, device):
        return torch.zeros(-(-n // 64) * 64 + 64, -(-k // 128) * 128, dtype=torch.bfloat16, device=device)

    def add(self, w, fwd=True, tr=False):
        """w: float32 [n, k] view of the flat parameter buffer -> dict(fwd=image or None, tr=image or None)."""
        n, k = w.shape
        ent = dict(fwd=self.alloc(n, k, w.device) if fwd else None, tr=self.alloc(k, n, w.device) if tr else None)
        if fwd:
            self.items.append((w, ent["fwd"], False))
        if tr:
            self.items.append((w, ent["tr"], True))
        return ent

    def add_side_by_side(self, ws_, biases):
        """One forward image [n, sum of the widths, each rounded up to 8] of several weights with the same number of rows,
        for y = [x_0 | x_1 | ...] [W_0 | W_1 | ...]^T = sum_i x_i W_i^T, and the sum of their biases (float32 [n]),
        refreshed with the images."""
        n = ws_[0].shape[0]
        widths = [-(-w.shape[1] // 8) * 8 for w in ws_]
        img = self.alloc(n, sum(widths), ws_[0].device)
        c0 = 0
        for w, wd in zip(ws_, widths):
            assert w.shape[0] == n
            self.items.append((w, img[:n, c0:c0 + wd], False))
            c0 += wd
        bsum = torch.zeros(n, dtype=torch.float32, device=ws_[0].device)
        assert len(biases) == 2
        self.sums.append((biases[0], biases[1], bsum))
        return dict(fwd=img, bias=bsum, widths=widths)

    def add_padded_bias(self, b, n_pad):
        """float32 copy of the bias vector b, zero-padded to n_pad entries (a product whose result is padded to a multiple of
        8 columns reads that many), refreshed with the images."""
        out = torch.zeros(n_pad, dtype=torch.float32, device=b.device)
        self.copies.append((b, out))
        return out

    def pack(self):
        if self._ver != self.fp.version:
            ops.pack_w16(self.items)
            for a, b, out in self.sums:
                ops.add_vec(a.reshape(-1), b.reshape(-1), out)
            for src, dst in self.copies:
                ops.copy_2d(src.reshape(1, -1), dst[:src.numel()].view(1, -1))
            self._ver = self.fp.version


def img_rows(img, r0, n):
    """Rows r0 .. r0+n of a weight image as an image of its own (n rounded up to 64 rows, the spare rows belong to
    the neighbouring slice or to the zero padding: their products land in columns that are never stored)."""
    return img[r0:r0 + -(-n // 64) * 64]


_CHUNK_STREAMS = {}      # device -> streams the sampler's concurrent chunks run on, shared by every engine of the process


class CounterNoise:
    """The noise of a reverse loop as a function instead of a [steps, B, T, K] tensor: step s of global rows row0 ..
    is drawn when the step needs it (tad_sampler_noise), by whoever owns those rows -- a rank, a concurrent chunk."""

    def __init__(self, seed, steps, row0=0):
        self.seed, self.steps, self.row0 = int(seed), int(steps), int(row0)
        self.shape = (self.steps,)

    def rows_from(self, first_row):
        return CounterNoise(self.seed, self.steps, self.row0 + int(first_row))

    def key(self):
        return ("counter", self.seed, self.steps, self.row0)


class Workspace:
    """Named device buffers, allocated on first use and reused while the shape is stable."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, name, shape, dtype=torch.float32, zero=False):
        """zero: True = cleared on every call, "new" = cleared only when (re)allocated (padding nobody writes)."""
        shape = tuple(int(s) for s in shape)
        t = self.bufs.get(name)
        fresh = t is None or tuple(t.shape) != shape or t.dtype != dtype
        if fresh:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs[name] = t
        if zero is True or (zero == "new" and fresh):
            ops.zero_(t)
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class use_workspace:
    """``with use_workspace(engine, ws):`` -- the engine's launch sequences work in ``ws`` inside the block.  A trainer
    runs its steps in a workspace of its own: a captured HIP graph holds the addresses of the buffers it was captured
    on, and the engine's default workspace is re-sized by whoever calls the module surface (model(...), ae.encoder,
    sample) with another batch size between two steps."""

    def __init__(self, eng, ws):
        self.eng, self.ws = eng, ws

    def __enter__(self):
        self.prev, self.eng.ws = self.eng.ws, self.ws
        return self.ws

    def __exit__(self, *exc):
        self.eng.ws = self.prev


# ---------------------------------------------------------------------------
class RnnStack:
    """Multi-layer LSTM/GRU with ``ndir`` directions ("bidir": every layer reads the
    concatenated output of all directions of the layer below, directions run
    forward/reverse in time) or independent heads ("heads": head d of layer l reads head
    d of layer l-1, all run forward in time; layer 0 shares the input).

    The per-timestep tensors (input-side pre-activations, gate stash, pre-activation
    gradients) are gate-interleaved: column d*4H + 4u + s holds slot s of unit u of
    direction d.  In bf16 precision a layer with an even H keeps them, and the gate-interleaved
    image of W_ih, in HBM as bfloat16; otherwise they are float32.  The 16-byte bf16 chunks need an
    input width that is a multiple of 8: a layer whose width is not (the VAE encoders read F = 28
    columns) works on a zero-padded copy of its input and a zero-padded weight image.""" 

    IN_SLOTS = {CELL_LSTM: (0, 1, 2, 3), CELL_GRU: (0, 1, 2, -1)}     # dgates slot -> gate of W_ih / b_ih
    HID_SLOTS = {CELL_LSTM: (0, 1, 2, 3), CELL_GRU: (0, 1, -1, 2)}    # dgates slot -> gate of W_hh / b_hh

    def __init__(self, fp: FlatParams, cell, H, in_dim, num_layers, dir_names, mode, prec):
        self.fp, self.cell, self.H, self.in_dim, self.L, self.mode, self.prec = fp, cell, H, in_dim, num_layers, mode, prec
        self.G = 4 if cell == CELL_LSTM else 3
        self.dir_names = dir_names              # list of (prefix, suffix_format)
        self.ndir = len(dir_names)
        self.rev = [False, True] if mode == "bidir" else [False] * self.ndir
        nbytes = ops.rnn_packed_bytes(prec, H, self.G)
        assert nbytes > 0
        dev = fp.device
        self.packed_f = [[torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.ndir)]
                         for _ in range(num_layers)]
        self.packed_b = [[torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.ndir)]
                         for _ in range(num_layers)]
        # heads mode slices a layer's input per head at multiples of H: bf16 rows then need H % 8 == 0
        ok16 = prec == BF16 and H % 2 == 0 and (mode == "bidir" or H % 8 == 0 or num_layers == 1)
        self.store = [torch.bfloat16 if ok16 else torch.float32 for layer in range(num_layers)]
        # bf16 layers whose widths are multiples of 8 take the bf16-stored-operand products: their gate-interleaved
        # W_ih image is padded the way ops.mm16_nt reads it (w_img; w_gl is its [rows, in_p] corner)
        self.mm16 = [MM16 and self.store[layer] == torch.bfloat16 and H % 8 == 0 for layer in range(num_layers)]
        self.w_img = [W16Images.alloc(self.ndir * 4 * H, self.in_p(layer), dev) if self.mm16[layer] else None
                      for layer in range(num_layers)]
        self.w_gl = [self.w_img[layer][:self.ndir * 4 * H] if self.mm16[layer] else
                     torch.empty(self.ndir * 4 * H, self.in_p(layer), dtype=self.store[layer], device=dev)
                     for layer in range(num_layers)]
        self.bias_gl = [torch.empty(self.ndir * 4 * H, dtype=torch.float32, device=dev) for _ in range(num_layers)]
        self._ver_f = self._ver_b = -1

    def name(self, kind, layer, d):
        prefix, sfx = self.dir_names[d]
        return f"{prefix}{kind}{sfx.format(l=layer)}"

    @staticmethod
    def layout(dir_names, num_layers):
        out = []
        for kind in ("weight_ih", "bias_ih", "bias_hh", "weight_hh"):
            for layer in range(num_layers):
                for d, (prefix, sfx) in enumerate(dir_names):
                    out.append((f"{prefix}{kind}{sfx.format(l=layer)}", d > 0))
        return out

    def in_l(self, layer):
        if layer == 0:
            return self.in_dim
        return self.ndir * self.H if self.mode == "bidir" else self.H

    def in_p(self, layer):
        """Input width the kernels see: padded to a multiple of 8 where the layer stores bf16."""
        n = self.in_l(layer)
        return -(-n // 8) * 8 if self.store[layer] == torch.bfloat16 else n

    def _padded(self, layer, x, ws, tag):
        """x [BT, in_l] -> [BT, in_p] with zero pad columns (x itself when no padding is needed)."""
        n, npad = self.in_l(layer), self.in_p(layer)
        if npad == n:
            return x
        key = f"{tag}.xpad{layer}"
        fresh = key not in ws.bufs or tuple(ws.bufs[key].shape) != (x.shape[0], npad)
        xp = ws.get(key, (x.shape[0], npad), zero=fresh)        # the pad columns are zeroed once and never written
        ops.copy_2d(x, xp[:, :n])
        return xp

    def pack(self, need_bwd=True):
        """Refresh the kernel-side images of the parameters (W_hh fragments, gate-interleaved W_ih
        and biases) if the parameters changed since the last pack."""
        v = self.fp.version
        do_f = self._ver_f != v
        do_b = need_bwd and self._ver_b != v
        if not (do_f or do_b):
            return
        fp, nd, GH = self.fp, self.ndir, self.G * self.H
        ops.rnn_pack_weights_multi(self.prec, self.H, self.G,
                                   [(fp.p(self.name("weight_hh", layer, d)), self.packed_f[layer][d] if do_f else None,
                                     self.packed_b[layer][d] if do_b else None) for layer in range(self.L) for d in range(nd)])
        for layer in range(self.L):
            if do_f:
                ops.rnn_pack_input(self.cell, self.H, fp.pgroup(self.name("weight_ih", layer, 0), nd * GH, self.in_l(layer)),
                                   fp.pgroup(self.name("bias_ih", layer, 0), 1, nd * GH).view(-1),
                                   fp.pgroup(self.name("bias_hh", layer, 0), 1, nd * GH).view(-1), self.w_gl[layer],
                                   self.bias_gl[layer])
        if do_f:
            self._ver_f = v
        if do_b:
            self._ver_b = v

    def pack_fwd_only(self):
        self.pack(need_bwd=False)

    def _xproj(self, layer, x, w_gl, bias_gl, xproj):
        """Input-side pre-activations from the rows ``w_gl`` of the layer's gate-interleaved weight image."""
        uf = self.G / 4.0           # a GRU's fourth slot is structural padding (zero weights)
        if x.dtype == torch.bfloat16:
            assert self.mm16[layer]
            n, k = xproj.shape[1], x.shape[1]
            if n * k >= XPROJ_TILED_MIN:
                # a weight of megabytes (hidden size 512: 4096 x 1024): mm16_nt would stream all of it through every
                # workgroup's LDS ring; the square-tiled kernel re-reads a 256-row panel of it per tile instead
                ops.linear_fwd_b16x(x, w_gl[:n, :k], bias_gl, xproj, useful=uf)
            else:
                ops.mm16_nt(x, w_gl, bias_gl, xproj, useful=uf)
        elif self.store[layer] == torch.bfloat16:
            ops.linear_fwd_b16(x, w_gl[:xproj.shape[1], :x.shape[1]], bias_gl, xproj, useful=uf)
        else:
            ops.linear_fwd(self.prec, x, w_gl, bias_gl, xproj, useful=uf)

    def forward(self, x, B, T, ws: Workspace, tag, train, last16=None):
        """x [B*T, in_dim] float32, or bfloat16 where layer 0 takes the bf16-stored-operand products.  Returns the
        float32 output of the last layer; ``self.out16`` holds the bfloat16 copies the sequence kernels wrote beside
        (per layer, None where a layer wrote none).  ``last16``: where the last layer's bfloat16 copy goes instead of a
        buffer of its own (a [B*T, ndir*H] view, e.g. the leading columns of a wider operand of the next product)."""
        fp, H, nd, prec = self.fp, self.H, self.ndir, self.prec
        BT = B * T
        H4 = 4 * H
        lstm = self.cell == CELL_LSTM
        self._in = x
        self._B, self._T = B, T
        self.out16 = [None] * self.L
        self._in16 = None
        if train and x.dtype == torch.float32 and self.mm16[0]:
            # the input-side weight gradient of layer 0 multiplies bf16 operands: a rounded copy of the input (padded to
            # a multiple of 8 columns with zeros) lets it ride in the layer's one weight-gradient launch
            self._in16 = ws.get(f"{tag}.in16", (BT, self.in_p(0)), dtype=torch.bfloat16, zero="new")
            ops.cvt_bf16(x, self._in16[:, :self.in_l(0)])
        inp = x
        for layer in range(self.L):
            sdt = self.store[layer]
            last = layer == self.L - 1
            # H = 512 in bf16: the form that keeps W_hh on chip, split over eight workgroups that exchange h_t through
            # device memory (one buffer per stack and workspace: the launches of a stream follow each other)
            wsb = ops.rnn_workspace_bytes(self.cell, prec, BF16 if sdt == torch.bfloat16 else F32, B, T, H, nd)
            # ... and a first layer that reads few columns (the VAE's encoders: the table's features) multiplies them by
            # W_ih inside that form: no pre-activation tensor [B*T, ndir*4H] is written or read
            fused = None
            # (that form addresses its tensors with 32-bit byte offsets: where the widest of them, the stash of nd * 4H bf16
            # columns, does not fit 2 GiB the launch is the plain one, which needs stored pre-activations)
            addressable = (B + 128) * T * nd * H4 * 2 < (1 << 31)
            if RNN_XIN and layer == 0 and wsb > 0 and addressable and self.mm16[0] and self.in_p(0) <= 32:
                x16 = x if x.dtype == torch.bfloat16 else self._in16
                if x16 is None:
                    x16 = ws.get(f"{tag}.in16", (BT, self.in_p(0)), dtype=torch.bfloat16, zero="new")
                    ops.cvt_bf16(x, x16[:, :self.in_l(0)])
                if x16.shape[1] == self.in_p(0):
                    fused = [dict(x16=x16, w_ih16=self.w_gl[0][d * H4:(d + 1) * H4], bias=self.bias_gl[0][d * H4:(d + 1) * H4])
                             for d in range(nd)]
            xproj = ws.get(f"{tag}.xproj{layer}", (BT, nd * H4), dtype=sdt) if fused is None else None
            w = self.w_img[layer] if self.mm16[layer] else self.w_gl[layer]
            if fused is not None:
                pass
            elif self.mode == "bidir" or layer == 0:
                xin = inp if inp.dtype == torch.bfloat16 else self._padded(layer, inp, ws, tag)
                self._xproj(layer, xin, w, self.bias_gl[layer], xproj)
            else:
                for d in range(nd):
                    sl = slice(d * H4, (d + 1) * H4)
                    assert self.in_p(layer) == self.in_l(layer)        # H even: no padding between heads
                    wd = img_rows(w, d * H4, H4) if self.mm16[layer] else w[sl]
                    self._xproj(layer, inp[:, d * H:(d + 1) * H], wd, self.bias_gl[layer][sl], xproj[:, sl])
            # h_t as bf16 for the products that read it (the next layer's input side, the weight gradients); as fp32
            # where something else reads it: the consumer of the stack's result and the backward of a GRU (h_{t-1})
            want16 = self.mm16[layer] and (train or (not last and self.mm16[layer + 1]))
            want32 = last or not want16 or (train and not lstm) or (not last and not self.mm16[layer + 1])
            out = ws.get(f"{tag}.out{layer}", (BT, nd * H)) if want32 else None
            out16 = ws.get(f"{tag}.out16_{layer}", (BT, nd * H), dtype=torch.bfloat16) if want16 else None
            if last and want16 and last16 is not None:
                assert tuple(last16.shape) == (BT, nd * H) and last16.dtype == torch.bfloat16
                out16 = last16
            self.out16[layer] = out16
            gates = ws.get(f"{tag}.gates{layer}", (BT, nd * H4), dtype=sdt) if train else None
            cst = ws.get(f"{tag}.c{layer}", (BT, nd * H)) if (train and lstm) else None
            dirs = []
            for d in range(nd):
                sl = slice(d * H4, (d + 1) * H4)
                hs = slice(d * H, (d + 1) * H)
                dirs.append(dict(xproj=xproj[:, sl] if xproj is not None else None, w_packed=self.packed_f[layer][d],
                                 b_hh=fp.p(self.name("bias_hh", layer, d)) if not lstm else None,
                                 out=out[:, hs] if out is not None else None,
                                 out16=out16[:, hs] if out16 is not None else None, gates=gates[:, sl] if train else None,
                                 cstate=cst[:, hs] if cst is not None else None, reverse=self.rev[d]))
            fp.release()
            cl_ws = ws.get(f"{tag}.rnn_ws", (wsb,), dtype=torch.uint8, zero="new") if wsb > 0 else None
            ops.rnn_fwd(self.cell, prec, B, T, H, dirs, workspace=cl_ws, xin=fused)
            inp = out16 if (not last and self.mm16[layer + 1] and out16 is not None) else out
        return inp

    def _weight_grads(self, layer, dg, inp, out, gb_ih, gb_hh, in_map, hid_map):
        """Weight and bias gradients of one layer from its pre-activation gradients ``dg``."""
        fp, H, nd, prec, T = self.fp, self.H, self.ndir, self.prec, self._T
        GH, H4 = self.G * H, 4 * H
        lstm = self.cell == CELL_LSTM
        shared = self.mode == "bidir" or layer == 0
        if self.mm16[layer] and self.out16[layer] is not None:
            # one launch: the input-side and both recurrent products read the pre-activation gradients once
            jobs = []
            if inp.dtype == torch.bfloat16 and (inp.shape[1] == (self.in_l(layer) if shared else nd * H) or
                                                (shared and inp.shape[1] == self.in_p(layer))):
                # (a zero-padded input: the product's spare columns are dropped, the gradient is in_l wide)
                if shared:
                    jobs.append(dict(b=inp, c=fp.ggroup(self.name("weight_ih", layer, 0), nd * GH, self.in_l(layer)), a0=0,
                                     na=nd * H4, gate_map=in_map, db=gb_ih, db2=gb_hh if lstm else None))
                else:
                    for d in range(nd):
                        sl = slice(d * GH, (d + 1) * GH)
                        jobs.append(dict(b=inp[:, d * H:(d + 1) * H], c=fp.g(self.name("weight_ih", layer, d)), a0=d * H4, na=H4,
                                         gate_map=in_map, db=gb_ih[sl], db2=gb_hh[sl] if lstm else None))
            elif layer == 0 and self._in16 is not None:
                jobs.append(dict(b=self._in16, c=fp.ggroup(self.name("weight_ih", layer, 0), nd * GH, self.in_l(layer)), a0=0,
                                 na=nd * H4, gate_map=in_map, db=gb_ih, db2=gb_hh if lstm else None))
            else:
                self._weight_grads_in(layer, dg, inp, gb_ih, gb_hh, in_map)
            o16 = self.out16[layer]
            for d in range(nd):
                jobs.append(dict(b=o16[:, d * H:(d + 1) * H], c=fp.g(self.name("weight_hh", layer, d)), a0=d * H4, na=H4,
                                 shift=(1 if self.rev[d] else -1), gate_map=hid_map,
                                 db=None if lstm else gb_hh[d * GH:(d + 1) * GH]))
            ops.mm16_tn(dg, jobs, T=T)
            return
        self._weight_grads_in(layer, dg, inp, gb_ih, gb_hh, in_map)
        # recurrent weights: dW_hh += dgates^T h_{t-1}
        for d in range(nd):
            ops.linear_bwd_weight(prec, dg[:, d * H4:(d + 1) * H4], out[:, d * H:(d + 1) * H],
                                  fp.g(self.name("weight_hh", layer, d)), db=None if lstm else gb_hh[d * GH:(d + 1) * GH],
                                  T=T, tshift=(1 if self.rev[d] else -1), gate_map=hid_map)

    def _weight_grads_in(self, layer, dg, inp, gb_ih, gb_hh, in_map):
        """Input-side weight and bias gradients of one layer through the fp32-operand product."""
        fp, H, nd, prec = self.fp, self.H, self.ndir, self.prec
        GH, H4 = self.G * H, 4 * H
        lstm = self.cell == CELL_LSTM
        if self.mode == "bidir" or layer == 0:
            ops.linear_bwd_weight(prec, dg, inp, fp.ggroup(self.name("weight_ih", layer, 0), nd * GH, self.in_l(layer)),
                                  db=gb_ih, db2=gb_hh if lstm else None, gate_map=in_map)
        else:
            for d in range(nd):
                sl = slice(d * GH, (d + 1) * GH)
                ops.linear_bwd_weight(prec, dg[:, d * H4:(d + 1) * H4], inp[:, d * H:(d + 1) * H],
                                      fp.g(self.name("weight_ih", layer, d)), db=gb_ih[sl], db2=gb_hh[sl] if lstm else None,
                                      gate_map=in_map)

    def backward(self, dout, ws: Workspace, tag, need_dx, relu_of=None, dx16=None):
        """BPTT through the stack; returns the gradient wrt the stack's input (None unless need_dx).
        dx16: a bfloat16 [B*T, in] buffer for that gradient when only matrix products read it (relu_of is then a
        bfloat16 mask operand): the product writes it directly where its operands are bf16-stored."""
        fp, H, G, nd, prec = self.fp, self.H, self.G, self.ndir, self.prec
        B, T = self._B, self._T
        BT = B * T
        GH, H4 = G * H, 4 * H
        lstm = self.cell == CELL_LSTM
        in_map = (H, G, self.IN_SLOTS[self.cell])
        hid_map = (H, G, self.HID_SLOTS[self.cell])
        for layer in range(self.L - 1, -1, -1):
            sdt = self.store[layer]
            out = ws.bufs.get(f"{tag}.out{layer}")          # fp32 h_t: absent where only bf16 copies were needed
            assert lstm or out is not None
            gates = ws.get(f"{tag}.gates{layer}", (BT, nd * H4), dtype=sdt)
            cst = ws.get(f"{tag}.c{layer}", (BT, nd * H)) if lstm else None
            if layer == 0:
                inp = self._in
            elif self.mm16[layer] and self.out16[layer - 1] is not None:
                inp = self.out16[layer - 1]
            else:
                inp = ws.bufs[f"{tag}.out{layer - 1}"]
            dg = ws.get(f"{tag}.dg{layer}", (BT, nd * H4), dtype=sdt)
            dirs = []
            for d in range(nd):
                sl = slice(d * H4, (d + 1) * H4)
                dirs.append(dict(dout=dout[:, d * H:(d + 1) * H], gates=gates[:, sl],
                                 cstate=cst[:, d * H:(d + 1) * H] if lstm else None,
                                 out=out[:, d * H:(d + 1) * H] if not lstm else None,
                                 w_packed_t=self.packed_b[layer][d], dgates=dg[:, sl], reverse=self.rev[d]))
            fp.release()
            ops.rnn_bwd(self.cell, prec, B, T, H, dirs)
            shared = self.mode == "bidir" or layer == 0
            in_l = self.in_l(layer)
            # input-side weights; the bias gradients ride along as row sums of the same dgates:
            # LSTM b_ih and b_hh both receive the sums; GRU b_hh receives the hidden-side sums with the W_hh pass
            gb_ih = fp.ggroup(self.name("bias_ih", layer, 0), 1, nd * GH).view(-1)
            gb_hh = fp.ggroup(self.name("bias_hh", layer, 0), 1, nd * GH).view(-1)
            fp.defer(self._weight_grads, layer, dg, inp, out, gb_ih, gb_hh, in_map, hid_map)
            # gradient wrt the layer input (the unused slot of a GRU meets a zero row of the weight image)
            if layer > 0 or need_dx:
                padded = shared and self.in_p(layer) != in_l
                wl = self.w_gl[layer][:, :self.in_p(layer)]
                if layer == 0 and dx16 is not None and sdt == torch.bfloat16 and shared and (not padded or
                                                                                             (dx16.shape[1] == self.in_p(0) and relu_of is None)):
                    # (a padded gradient: its spare columns meet the image's zero columns and hold zeros)
                    ops.linear_bwd_data_b16o(dg, wl, dx16, relu_of16=relu_of, useful=G / 4.0)
                    dout = dx16
                    continue
                dx = ws.get(f"{tag}.dx{layer}", (BT, (self.in_p(layer) if padded else in_l) if shared else nd * H))
                if layer == 0 and dx16 is not None:
                    relu_of, mask16 = None, relu_of            # the mask is applied with the rounding below
                ro = relu_of if layer == 0 else None
                assert not (padded and ro is not None), "a ReLU mask on a padded layer input is not supported"
                bwd_data = (lambda dy, w, o, r=None: ops.linear_bwd_data_b16(dy, w, o, relu_of=r, useful=G / 4.0)) if sdt == torch.bfloat16 else \
                    (lambda dy, w, o, r=None: ops.linear_bwd_data(prec, dy, w, o, relu_of=r, useful=G / 4.0))
                if shared:
                    bwd_data(dg, wl, dx, ro)
                else:
                    for d in range(nd):
                        sl = slice(d * H4, (d + 1) * H4)
                        bwd_data(dg[:, sl], wl[sl], dx[:, d * H:(d + 1) * H])
                dout = dx[:, :in_l] if padded else dx            # the pad columns meet zero weights: they hold zeros
                if layer == 0 and dx16 is not None:
                    ops.cvt_bf16(dout, dx16, relu_of=mask16)
                    dout = dx16
            else:
                dout = None
        return dout


# ---------------------------------------------------------------------------
class EmbBlock:
    """Embedding_data: per-column embedding lookups | sin/cos features -> mlp_nums, then mlp_output."""

    def __init__(self, fp: FlatParams, prefix, n_bins, n_cats, n_nums, cards, E, prec):
        self.fp, self.prefix, self.E, self.prec = fp, prefix, E, prec
        self.nd = n_bins + n_cats
        self.nn = n_nums
        self.cards = [2] * n_bins + list(cards)[:n_cats]
        self.tab_off = [0]
        for c in self.cards[:-1]:
            self.tab_off.append(self.tab_off[-1] + c)
        self.cat_dim = self.nd * E + 16 * n_nums
        self.mm16 = self.last16 = self.pad16 = False

    @staticmethod
    def layout(prefix, n_disc, n_nums):
        out = [(f"{prefix}embeddings_list.{j}.weight", j > 0) for j in range(n_disc)]
        if n_nums:
            out += [(f"{prefix}mlp_nums.0.weight", False), (f"{prefix}mlp_nums.0.bias", False),
                    (f"{prefix}mlp_nums.2.weight", False), (f"{prefix}mlp_nums.2.bias", False)]
        out += [(f"{prefix}mlp_output.0.weight", False), (f"{prefix}mlp_output.0.bias", False),
                (f"{prefix}mlp_output.2.weight", False), (f"{prefix}mlp_output.2.bias", False)]
        return out

    def use_mm16(self, imgs, pad_last=False):
        """Switch the block to the bf16-stored-operand products (bf16 precision, widths that are multiples of 8):
        registers its weight images with ``imgs`` (W16Images).  The last layer joins when its width allows, or, with
        ``pad_last``, on a result padded to a multiple of 8 columns (the spare columns meet zero weights and a zero bias:
        they hold zeros; forward(want16=True) then returns the padded bfloat16 result and backward takes a gradient of
        that width)."""
        fp, pre = self.fp, self.prefix
        if not (MM16 and self.prec == BF16 and self.E % 8 == 0):
            return
        self.imgs = imgs
        self.out_w = fp.p(pre + "mlp_output.2.weight").shape[0]
        self.out_p = -(-self.out_w // 8) * 8
        self.last16 = self.out_w % 8 == 0
        self.pad16 = bool(pad_last) and not self.last16
        self.im = {}
        if self.nn:
            self.im["n0"] = imgs.add(fp.p(pre + "mlp_nums.0.weight"), fwd=True, tr=False)
            self.im["n2"] = imgs.add(fp.p(pre + "mlp_nums.2.weight"), fwd=True, tr=True)
        self.im["o0"] = imgs.add(fp.p(pre + "mlp_output.0.weight"), fwd=True, tr=True)
        if self.last16 or self.pad16:
            self.im["o2"] = imgs.add(fp.p(pre + "mlp_output.2.weight"), fwd=True, tr=True)
        if self.pad16:
            self.im["o2_bias"] = imgs.add_padded_bias(fp.p(pre + "mlp_output.2.bias"), self.out_p)
        self.mm16 = True

    def forward(self, x, ws: Workspace, tag, want16=False):
        if self.mm16:
            return self._forward16(x, ws, tag, want16)
        fp, pre, E, prec, nd, nn = self.fp, self.prefix, self.E, self.prec, self.nd, self.nn
        rows = x.shape[0]
        self._x = x
        ecat = ws.get(f"{tag}.ecat", (rows, self.cat_dim))
        if nd:
            tables = fp.pgroup(f"{pre}embeddings_list.0.weight", sum(self.cards), E)
            ops.embed_fwd(x[:, :nd], tables, self.tab_off, self.cards, E, ecat[:, :nd * E])
        if nn:
            feats = ws.get(f"{tag}.feats", (rows, 16 * nn))
            z1 = ws.get(f"{tag}.z1", (rows, 16 * nn))
            a1 = ws.get(f"{tag}.a1", (rows, 16 * nn))
            ops.sincos_features(x[:, nd:nd + nn], feats)
            ops.linear_fwd(prec, feats, fp.p(pre + "mlp_nums.0.weight"), fp.p(pre + "mlp_nums.0.bias"), z1)
            ops.silu_fwd(z1, a1)
            ops.linear_fwd(prec, a1, fp.p(pre + "mlp_nums.2.weight"), fp.p(pre + "mlp_nums.2.bias"), ecat[:, nd * E:])
        a2 = ws.get(f"{tag}.a2", (rows, E))
        ops.linear_fwd(prec, ecat, fp.p(pre + "mlp_output.0.weight"), fp.p(pre + "mlp_output.0.bias"), a2, act=ACT_RELU)
        w2 = fp.p(pre + "mlp_output.2.weight")
        out = ws.get(f"{tag}.out", (rows, w2.shape[0]))
        ops.linear_fwd(prec, a2, w2, fp.p(pre + "mlp_output.2.bias"), out)
        return out

    def _forward16(self, x, ws, tag, want16):
        """The same launches with every tensor that only matrix products read kept as bfloat16 (the products round
        their operands to bf16 anyway, so the values are those of the fp32-stored path); z1, which SiLU and its
        derivative read, stays float32.  Returns bfloat16 if ``want16`` and the last layer's width allows."""
        fp, pre, E, nd, nn, im = self.fp, self.prefix, self.E, self.nd, self.nn, self.im
        rows = x.shape[0]
        b16 = torch.bfloat16
        self._x = x
        self.imgs.pack()
        ecat = ws.get(f"{tag}.ecat16", (rows, self.cat_dim), dtype=b16)
        if nd:
            tables = fp.pgroup(f"{pre}embeddings_list.0.weight", sum(self.cards), E)
            ops.embed_fwd(x[:, :nd], tables, self.tab_off, self.cards, E, ecat[:, :nd * E])
        if nn:
            feats = ws.get(f"{tag}.feats16", (rows, 16 * nn), dtype=b16)
            z1 = ws.get(f"{tag}.z1", (rows, 16 * nn))
            a1 = ws.get(f"{tag}.a1_16", (rows, 16 * nn), dtype=b16)
            ops.sincos_features(x[:, nd:nd + nn], feats)
            # z1 (float32: SiLU's backward reads it) and a1 = silu(z1) (bfloat16: the next product's operand) leave the
            # same launch
            ops.mm16_nt(feats, im["n0"]["fwd"], fp.p(pre + "mlp_nums.0.bias"), a1, act=ACT_SILU_Z32, z_out=z1)
            ops.mm16_nt(a1, im["n2"]["fwd"], fp.p(pre + "mlp_nums.2.bias"), ecat[:, nd * E:])
        w2 = fp.p(pre + "mlp_output.2.weight")
        if self.pad16:
            a2 = ws.get(f"{tag}.a2_16", (rows, E), dtype=b16)
            ops.mm16_nt(ecat, im["o0"]["fwd"], fp.p(pre + "mlp_output.0.bias"), a2, act=ACT_RELU)
            outp = ws.get(f"{tag}.out16p" if want16 else f"{tag}.outp", (rows, self.out_p), dtype=b16 if want16 else torch.float32)
            ops.mm16_nt(a2, im["o2"]["fwd"], im["o2_bias"], outp)
            out = outp if want16 else outp[:, :self.out_w]
        elif self.last16:
            a2 = ws.get(f"{tag}.a2_16", (rows, E), dtype=b16)
            ops.mm16_nt(ecat, im["o0"]["fwd"], fp.p(pre + "mlp_output.0.bias"), a2, act=ACT_RELU)
            out = ws.get(f"{tag}.out16" if want16 else f"{tag}.out", (rows, w2.shape[0]), dtype=b16 if want16 else torch.float32)
            ops.mm16_nt(a2, im["o2"]["fwd"], fp.p(pre + "mlp_output.2.bias"), out)
        else:
            a2 = ws.get(f"{tag}.a2", (rows, E))
            ops.mm16_nt(ecat, im["o0"]["fwd"], fp.p(pre + "mlp_output.0.bias"), a2, act=ACT_RELU)
            out = ws.get(f"{tag}.out", (rows, w2.shape[0]))
            ops.linear_fwd(self.prec, a2, w2, fp.p(pre + "mlp_output.2.bias"), out)
        return out

    def backward(self, dout, ws: Workspace, tag):
        if self.mm16:
            return self._backward16(dout, ws, tag)
        fp, pre, E, prec, nd, nn = self.fp, self.prefix, self.E, self.prec, self.nd, self.nn
        rows = dout.shape[0]
        x = self._x
        ecat = ws.get(f"{tag}.ecat", (rows, self.cat_dim))
        a2 = ws.get(f"{tag}.a2", (rows, E))
        da2 = ws.get(f"{tag}.da2", (rows, E))
        ops.linear_bwd_data(prec, dout, fp.p(pre + "mlp_output.2.weight"), da2, relu_of=a2)
        fp.defer(ops.linear_bwd_weight, prec, dout, a2, fp.g(pre + "mlp_output.2.weight"), db=fp.g(pre + "mlp_output.2.bias"))
        decat = ws.get(f"{tag}.decat", (rows, self.cat_dim))
        ops.linear_bwd_data(prec, da2, fp.p(pre + "mlp_output.0.weight"), decat)
        fp.defer(ops.linear_bwd_weight, prec, da2, ecat, fp.g(pre + "mlp_output.0.weight"), db=fp.g(pre + "mlp_output.0.bias"))
        if nd:
            gt = fp.ggroup(f"{pre}embeddings_list.0.weight", sum(self.cards), E)
            fp.defer(ops.embed_bwd, x[:, :nd], decat[:, :nd * E], gt, self.tab_off, self.cards, E)
        if nn:
            feats = ws.get(f"{tag}.feats", (rows, 16 * nn))
            z1 = ws.get(f"{tag}.z1", (rows, 16 * nn))
            a1 = ws.get(f"{tag}.a1", (rows, 16 * nn))
            dy1 = decat[:, nd * E:]
            da1 = ws.get(f"{tag}.da1", (rows, 16 * nn))
            ops.linear_bwd_data(prec, dy1, fp.p(pre + "mlp_nums.2.weight"), da1)
            fp.defer(ops.linear_bwd_weight, prec, dy1, a1, fp.g(pre + "mlp_nums.2.weight"), db=fp.g(pre + "mlp_nums.2.bias"))
            dz1 = ws.get(f"{tag}.dz1", (rows, 16 * nn))
            ops.silu_bwd(z1, da1, dz1)
            fp.defer(ops.linear_bwd_weight, prec, dz1, feats, fp.g(pre + "mlp_nums.0.weight"), db=fp.g(pre + "mlp_nums.0.bias"))

    def _backward16(self, dout, ws, tag):
        """dout: float32 or bfloat16 [rows, width of the block's result]."""
        fp, pre, E, prec, nd, nn, im = self.fp, self.prefix, self.E, self.prec, self.nd, self.nn, self.im
        rows = dout.shape[0]
        b16 = torch.bfloat16
        x = self._x
        ecat = ws.get(f"{tag}.ecat16", (rows, self.cat_dim), dtype=b16)
        da2 = ws.get(f"{tag}.da2_16", (rows, E), dtype=b16)
        gw2, gb2 = fp.g(pre + "mlp_output.2.weight"), fp.g(pre + "mlp_output.2.bias")
        if self.last16 or self.pad16:
            a2 = ws.get(f"{tag}.a2_16", (rows, E), dtype=b16)
            if dout.dtype != b16:
                # (padded result: the gradient's spare columns are zeros, written once)
                d16 = ws.get(f"{tag}.dout16", (rows, self.out_p), dtype=b16, zero="new")
                ops.cvt_bf16(dout, d16[:, :dout.shape[1]])
                dout = d16
            assert dout.shape[1] == self.out_p
            ops.mm16_nt(dout, im["o2"]["tr"], None, da2, relu_of=a2)
            fp.queue_tn(dout, dict(b=a2, c=gw2, db=gb2, a0=0, na=self.out_w))
        else:
            a2 = ws.get(f"{tag}.a2", (rows, E))
            da2f = ws.get(f"{tag}.da2", (rows, E))
            ops.linear_bwd_data(prec, dout, fp.p(pre + "mlp_output.2.weight"), da2f, relu_of=a2)
            fp.defer(ops.linear_bwd_weight, prec, dout, a2, gw2, db=gb2)
            ops.cvt_bf16(da2f, da2)
        fp.queue_tn(da2, dict(b=ecat, c=fp.g(pre + "mlp_output.0.weight"), db=fp.g(pre + "mlp_output.0.bias")))
        wt = im["o0"]["tr"]                       # [cat_dim, E]: rows = columns of ecat
        if nd:
            # the table gradients sum these values element by element: float32
            de = ws.get(f"{tag}.decat_d", (rows, nd * E))
            ops.mm16_nt(da2, img_rows(wt, 0, nd * E), None, de)
            gt = fp.ggroup(f"{pre}embeddings_list.0.weight", sum(self.cards), E)
            fp.defer(ops.embed_bwd, x[:, :nd], de, gt, self.tab_off, self.cards, E)
        if nn:
            feats = ws.get(f"{tag}.feats16", (rows, 16 * nn), dtype=b16)
            z1 = ws.get(f"{tag}.z1", (rows, 16 * nn))
            a1 = ws.get(f"{tag}.a1_16", (rows, 16 * nn), dtype=b16)
            dy1 = ws.get(f"{tag}.dy1_16", (rows, 16 * nn), dtype=b16)
            ops.mm16_nt(da2, img_rows(wt, nd * E, 16 * nn), None, dy1)
            fp.queue_tn(dy1, dict(b=a1, c=fp.g(pre + "mlp_nums.2.weight"), db=fp.g(pre + "mlp_nums.2.bias")))
            # d(z1) = (dy1 W) * silu'(z1): the SiLU's backward is the product's epilogue (d(a1) never reaches memory)
            dz1 = ws.get(f"{tag}.dz1_16", (rows, 16 * nn), dtype=b16)
            ops.mm16_nt(dy1, im["n2"]["tr"], None, dz1, act=ACT_DSILU, relu_of=z1)
            fp.queue_tn(dz1, dict(b=feats, c=fp.g(pre + "mlp_nums.0.weight"), db=fp.g(pre + "mlp_nums.0.bias")))


def _lin_bwd(fp, prec, dy, x, w, gw, gb, dx=None, relu_of=None, accumulate=False):
    """Backward of y = x W^T + b: weight/bias grads always (deferred to the gradient stream of ``fp``), input grad
    if dx is given."""
    if dx is not None:
        ops.linear_bwd_data(prec, dy, w, dx, relu_of=relu_of, accumulate=accumulate)
    fp.defer(ops.linear_bwd_weight, prec, dy, x, gw, db=gb)


# ---------------------------------------------------------------------------
class DenoiserEngine:
    """Launch sequences of BiRNN_score.forward (reference DIFF.py:196-224), its backward,
    and the fused loss step of get_loss (DIFF.py:252-260)."""

    def __init__(self, module, device, dtype="fp32"):
        self.prec = prec_of(dtype)
        self.device = device
        c = module.cfg
        self.K, self.H, self.E, self.L, self.steps = c["input_size"], c["hidden_size"], c["emb_dim"], c["num_layers"], c["diffusion_steps"]
        self.time_dim = c["time_dim"]
        nb, nc, nn, cards = c["n_bins"], c["n_cats"], c["n_nums"], c["cards"]
        self.F = nb + nc + nn
        lstm_dirs = [("lstm.", "_l{l}"), ("lstm.", "_l{l}_reverse")]
        gru_dirs = [("cond_GRU.", "_l{l}"), ("cond_GRU.", "_l{l}_reverse")]
        layout = [(n, False) for n in ("input_proj.net.0.weight", "input_proj.net.0.bias", "t_enc.linear_proj.weight",
                                       "t_enc.linear_proj.bias", "i_enc.linear_proj.weight", "i_enc.linear_proj.bias",
                                       "proj.net.0.weight", "proj.net.0.bias")]
        layout += RnnStack.layout(lstm_dirs, self.L)
        layout += [(n, False) for n in ("layer_norm.weight", "layer_norm.bias", "fc.weight", "fc.bias")]
        layout += EmbBlock.layout("Emb.", nb + nc, nn)
        layout += RnnStack.layout(gru_dirs, self.L)
        layout += [(n, False) for n in ("cond_output.weight", "cond_output.bias", "time_encode.0.weight",
                                        "time_encode.0.bias", "time_encode.2.weight", "time_encode.2.bias")]
        inactive = ["Emb_hidden.weight", "Emb_hidden.bias"]
        self.fp = FlatParams(module, layout, inactive, device)
        self.lstm = RnnStack(self.fp, CELL_LSTM, self.H, self.H, self.L, lstm_dirs, "bidir", self.prec)
        self.gru = RnnStack(self.fp, CELL_GRU, self.H, self.E, self.L, gru_dirs, "bidir", self.prec)
        self.emb = EmbBlock(self.fp, "Emb.", nb, nc, nn, cards, self.E, self.prec)
        self.imgs = W16Images(self.fp)
        self.emb.use_mm16(self.imgs)
        # the training step keeps what only matrix products read as bfloat16 when every width involved is a multiple of 8
        self.mm16 = bool(MM16 and self.prec == BF16 and self.H % 8 == 0 and 4 * self.H <= MM16_MAX_4H and self.emb.mm16 and self.emb.last16
                         and all(self.gru.mm16) and all(self.lstm.mm16))
        if self.mm16:
            p_ = self.fp.p
            # cond_output and input_proj add up (reference DIFF.py:197-198: x = input_proj(x) + cond_output(cond_out)): one
            # product over [h_gru | x] with the two weights side by side
            self.im = dict(co=self.imgs.add(p_("cond_output.weight"), fwd=False, tr=True),
                           coin=self.imgs.add_side_by_side([p_("cond_output.weight"), p_("input_proj.net.0.weight")],
                                                           [p_("cond_output.bias"), p_("input_proj.net.0.bias")]),
                           proj=self.imgs.add(p_("proj.net.0.weight"), fwd=True, tr=False),
                           te2=self.imgs.add(p_("time_encode.2.weight"), fwd=True, tr=True))
            if self.time_dim % 8 == 0:
                self.im["te0"] = self.imgs.add(p_("time_encode.0.weight"), fwd=True, tr=False)
            # d(cat) is needed for the first and the last H columns of cat only (input projection / cond_output share,
            # time-encoding share): their rows of proj's transposed weight, one below the other, make one product
            Hh, wp = self.H, p_("proj.net.0.weight")
            tr2 = self.imgs.alloc(2 * Hh, Hh, device)
            self.imgs.items.append((wp[:, :Hh], tr2[:Hh], True))
            self.imgs.items.append((wp[:, 3 * Hh:], tr2[Hh:2 * Hh], True))
            self.im["proj"]["tr2"] = tr2
            # the sampler's per-step share of `proj`: its first H input columns (the input projection of x)
            self.im["proj_x"] = self.imgs.add(p_("proj.net.0.weight")[:, :self.H], fwd=True, tr=False)
        self.ws = Workspace(device)
        # data parallel: the gradients of input_proj, t_enc, i_enc, proj, the LSTM, LayerNorm and fc occupy the head of
        # the flat buffer and are complete before the backward of the conditioning path starts: a trainer may reduce
        # that range over the ranks while the rest of the backward runs (callback, called once per backward)
        self.first_bucket_end = self.fp.offsets["Emb.embeddings_list.0.weight" if (nb + nc) else
                                                ("Emb.mlp_nums.0.weight" if nn else "Emb.mlp_output.0.weight")]
        self.on_first_bucket = None
        self.scale_t = module.t_enc.scale.to(device=device, dtype=torch.float32).contiguous()
        self.shift_t = module.t_enc.shift.to(device=device, dtype=torch.float32).contiguous()
        self.scale_i = module.i_enc.scale.to(device=device, dtype=torch.float32).contiguous()
        self.shift_i = module.i_enc.shift.to(device=device, dtype=torch.float32).contiguous()
        self.i_max = float(module.i_enc.max_value)
        self._chunks = []           # (stream, workspace, event) per concurrent sampler chunk
        self._graphs = {}           # (batch shape, chunking) -> (captured reverse loop, its static input buffers)

    # -- forward ------------------------------------------------------------
    def forward(self, x, t, i_f, time_info, cond, train=False):
        """x [B,T,K], t [B,T,1], i_f [B,T(,1)] float steps, time_info [B,T,td], cond [B,T,F] ->
        predicted noise [B,T,K] (a workspace buffer, valid until the next call)."""
        fp, ws, prec, H, K = self.fp, self.ws, self.prec, self.H, self.K
        B, T, _ = x.shape
        BT = B * T
        if cond.shape[1] != T:
            raise ValueError("cond must cover the same T steps as x (right-padding a shorter cond is not supported)")
        # the kernel-side weight images depend on the parameters only: a dozen small launches that run on the side
        # stream beside the embedding block; joined before the first recurrent layer needs them
        if train:
            fp.defer(self.lstm.pack)
            fp.defer(self.gru.pack)
        else:
            fp.defer(self.lstm.pack_fwd_only)
            fp.defer(self.gru.pack_fwd_only)
        x2 = x.reshape(BT, K)
        t1 = t.reshape(BT)
        i1 = i_f.reshape(BT)
        ti2 = time_info.reshape(BT, self.time_dim)
        c2 = cond.reshape(BT, self.F)
        self._saved = (x2, t1, i1, ti2, B, T)
        self._fwd16 = self.mm16 and train
        if self._fwd16:
            return self._forward16(x2, t1, i1, ti2, c2, B, T)
        cat = ws.get("cat", (BT, 4 * H))
        ops.linear_fwd(prec, x2, fp.p("input_proj.net.0.weight"), fp.p("input_proj.net.0.bias"), cat[:, :H])
        emb = self.emb.forward(c2, ws, "emb")
        fp.join()
        # the time / step encodings depend on nothing the conditioning path computes: they are queued for the side
        # stream and run beside the conditioning GRU's first sequence kernel
        ta = ws.get("ta", (BT, H))
        fp.defer_until_release(self._encodings, cat, ta, t1, i1, ti2, BT)
        gout = self.gru.forward(emb, B, T, ws, "gru", train)
        ops.linear_fwd(prec, gout, fp.p("cond_output.weight"), fp.p("cond_output.bias"), cat[:, :H], accumulate=True)
        fp.join()
        pa = ws.get("pa", (BT, H))
        ops.linear_fwd(prec, cat, fp.p("proj.net.0.weight"), fp.p("proj.net.0.bias"), pa, act=ACT_RELU)
        lout = self.lstm.forward(pa, B, T, ws, "lstm", train)
        return self._ln_fc(lout, BT, train).view(B, T, K)

    def _ln_fc(self, lout, BT, train):
        """layer_norm + fc (reference DIFF.py:221-223): one launch that keeps the normalised row in registers where the
        widths allow (the backward recomputes it), else LayerNorm and the projection as two."""
        fp, ws, prec, H, K = self.fp, self.ws, self.prec, self.H, self.K
        mr = ws.get("mr", (BT, 2))
        pred = ws.get("pred", (BT, K))
        self._lnfc = LN_FC_FUSED and ops.ln_fc_supported(2 * H, K)
        if self._lnfc:
            ops.ln_fc_fwd(prec, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), fp.p("fc.weight"), fp.p("fc.bias"), mr, pred)
        else:
            ln = ws.get("ln", (BT, 2 * H))
            ops.layernorm_fwd(lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), ln, mr)
            ops.linear_fwd(prec, ln, fp.p("fc.weight"), fp.p("fc.bias"), pred)
        return pred

    def _ln_fc_bwd(self, dpred, lout, BT):
        """-> d(loss)/d(recurrent output); the gradients of layer_norm and fc are accumulated."""
        fp, ws, prec, H = self.fp, self.ws, self.prec, self.H
        mr = ws.get("mr", (BT, 2))
        dlout = ws.get("dlout", (BT, 2 * H))
        if self._lnfc:
            ops.ln_fc_bwd(prec, dpred, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), mr, fp.p("fc.weight"), dlout,
                          fp.g("layer_norm.weight"), fp.g("layer_norm.bias"), fp.g("fc.weight"), fp.g("fc.bias"))
        else:
            ln = ws.get("ln", (BT, 2 * H))
            dln = ws.get("dln", (BT, 2 * H))
            _lin_bwd(fp, prec, dpred, ln, fp.p("fc.weight"), fp.g("fc.weight"), fp.g("fc.bias"), dx=dln)
            ops.layernorm_bwd(dln, lout, fp.p("layer_norm.weight"), mr, dlout, fp.g("layer_norm.weight"), fp.g("layer_norm.bias"))
        return dlout

    def _forward16(self, x2, t1, i1, ti2, c2, B, T):
        """The training forward with every tensor that only matrix products read kept as bfloat16: the products
        round their operands to bf16 anyway, so a tensor stored rounded yields the same values as one stored in
        float32.  What anything else reads stays float32: the GRU's h_t (its backward), the LSTM's last h_t (LayerNorm).
        input_proj(x) + cond_output(h_gru) is one product over [h_gru | x], summed in the float32 accumulators before the
        one rounding of `proj`'s operand."""
        fp, ws, prec, H, K, im = self.fp, self.ws, self.prec, self.H, self.K, self.im
        BT = B * T
        b16 = torch.bfloat16
        self.imgs.pack()
        # [h_gru | x | 0] as one bf16 operand: the GRU writes its part, x is rounded here (as the product would on load)
        Kp = im["coin"]["widths"][1]
        gx = ws.get("gx16", (BT, 2 * H + Kp), dtype=b16, zero="new")       # the pad columns stay zero
        ops.cvt_bf16(x2, gx[:, 2 * H:2 * H + K])
        emb = self.emb.forward(c2, ws, "emb", want16=True)
        fp.join()
        cat = ws.get("cat16", (BT, 4 * H), dtype=b16)
        ta = ws.get("ta16", (BT, H), dtype=b16)
        fp.defer_until_release(self._encodings16, cat, ta, t1, i1, ti2, BT)
        self.gru.forward(emb, B, T, ws, "gru", True, last16=gx[:, :2 * H])
        ops.mm16_nt(gx, im["coin"]["fwd"], im["coin"]["bias"], cat[:, :H])
        fp.join()
        pa = ws.get("pa16", (BT, H), dtype=b16)
        ops.mm16_nt(cat, im["proj"]["fwd"], fp.p("proj.net.0.bias"), pa, act=ACT_RELU)
        lout = self.lstm.forward(pa, B, T, ws, "lstm", True)
        return self._ln_fc(lout, BT, True).view(B, T, K)

    def _encodings16(self, cat, ta, t1, i1, ti2, BT):
        fp, ws, prec, H, im = self.fp, self.ws, self.prec, self.H, self.im
        ops.posenc_fwd(t1, 1, BT, H, 1.0, fp.p("t_enc.linear_proj.weight"), fp.p("t_enc.linear_proj.bias"), self.scale_t,
                       self.shift_t, cat[:, H:2 * H])
        ops.posenc_fwd(i1, 1, BT, H, 1.0 / self.i_max, fp.p("i_enc.linear_proj.weight"), fp.p("i_enc.linear_proj.bias"),
                       self.scale_i, self.shift_i, cat[:, 2 * H:3 * H])
        if "te0" in im:
            ti16 = ws.get("ti16", (BT, self.time_dim), dtype=torch.bfloat16)
            ops.cvt_bf16(ti2, ti16)
            ops.mm16_nt(ti16, im["te0"]["fwd"], fp.p("time_encode.0.bias"), ta, act=ACT_RELU)
        else:
            taf = ws.get("ta", (BT, H))
            ops.linear_fwd(prec, ti2, fp.p("time_encode.0.weight"), fp.p("time_encode.0.bias"), taf, act=ACT_RELU)
            ops.cvt_bf16(taf, ta)
        ops.mm16_nt(ta, im["te2"]["fwd"], fp.p("time_encode.2.bias"), cat[:, 3 * H:])

    def _encodings(self, cat, ta, t1, i1, ti2, BT):
        """t_enc | i_enc | time_encode -> columns H..4H of the input of `proj` (reference DIFF.py:199-204)."""
        fp, prec, H = self.fp, self.prec, self.H
        ops.posenc_fwd(t1, 1, BT, H, 1.0, fp.p("t_enc.linear_proj.weight"), fp.p("t_enc.linear_proj.bias"), self.scale_t,
                       self.shift_t, cat[:, H:2 * H])
        ops.posenc_fwd(i1, 1, BT, H, 1.0 / self.i_max, fp.p("i_enc.linear_proj.weight"), fp.p("i_enc.linear_proj.bias"),
                       self.scale_i, self.shift_i, cat[:, 2 * H:3 * H])
        ops.linear_fwd(prec, ti2, fp.p("time_encode.0.weight"), fp.p("time_encode.0.bias"), ta, act=ACT_RELU)
        ops.linear_fwd(prec, ta, fp.p("time_encode.2.weight"), fp.p("time_encode.2.bias"), cat[:, 3 * H:])

    # -- backward -----------------------------------------------------------
    def backward(self, dpred):
        """Accumulates d(loss)/d(params) into the flat gradient buffer given d(loss)/d(output)."""
        fp, ws, prec, H, K = self.fp, self.ws, self.prec, self.H, self.K
        x2, t1, i1, ti2, B, T = self._saved
        BT = B * T
        Lh = H // 2
        dpred = dpred.reshape(BT, K)
        if self._fwd16:
            return self._backward16(dpred)
        cat = ws.get("cat", (BT, 4 * H))
        ta = ws.get("ta", (BT, H))
        pa = ws.get("pa", (BT, H))
        lout = ws.get("lstm.out%d" % (self.L - 1), (BT, 2 * H))
        gout = ws.get("gru.out%d" % (self.L - 1), (BT, 2 * H))
        dlout = self._ln_fc_bwd(dpred, lout, BT)
        dpz = self.lstm.backward(dlout, ws, "lstm", need_dx=True, relu_of=pa)      # includes the ReLU of proj
        dcat = ws.get("dcat", (BT, 4 * H))
        _lin_bwd(fp, prec, dpz, cat, fp.p("proj.net.0.weight"), fp.g("proj.net.0.weight"), fp.g("proj.net.0.bias"), dx=dcat)
        dxh = dcat[:, :H]
        _lin_bwd(fp, prec, dxh, x2, None, fp.g("input_proj.net.0.weight"), fp.g("input_proj.net.0.bias"))
        fp.defer(ops.colsum, dcat[:, H:H + Lh], out_sum=fp.g("t_enc.linear_proj.bias"), row_scale=t1, scale_mul=1.0,
                 out_wsum=fp.g("t_enc.linear_proj.weight"))
        fp.defer(ops.colsum, dcat[:, 2 * H:2 * H + Lh], out_sum=fp.g("i_enc.linear_proj.bias"), row_scale=i1, scale_mul=1.0 / self.i_max,
                 out_wsum=fp.g("i_enc.linear_proj.weight"))
        if self.on_first_bucket is not None:
            self.on_first_bucket()          # gradients of input_proj .. fc (the head of the flat buffer) are all enqueued
        dtenc = dcat[:, 3 * H:]
        dta = ws.get("dta", (BT, H))
        _lin_bwd(fp, prec, dtenc, ta, fp.p("time_encode.2.weight"), fp.g("time_encode.2.weight"), fp.g("time_encode.2.bias"),
                 dx=dta, relu_of=ta)
        _lin_bwd(fp, prec, dta, ti2, None, fp.g("time_encode.0.weight"), fp.g("time_encode.0.bias"))
        dgout = ws.get("dgout", (BT, 2 * H))
        _lin_bwd(fp, prec, dxh, gout, fp.p("cond_output.weight"), fp.g("cond_output.weight"), fp.g("cond_output.bias"), dx=dgout)
        demb = self.gru.backward(dgout, ws, "gru", need_dx=True)
        self.emb.backward(demb, ws, "emb")
        fp.join()

    def _backward16(self, dpred):
        fp, ws, prec, H, K, im = self.fp, self.ws, self.prec, self.H, self.K, self.im
        x2, t1, i1, ti2, B, T = self._saved
        BT = B * T
        Lh = H // 2
        b16 = torch.bfloat16
        cat = ws.get("cat16", (BT, 4 * H), dtype=b16)
        ta = ws.get("ta16", (BT, H), dtype=b16)
        pa = ws.get("pa16", (BT, H), dtype=b16)
        lout = ws.get("lstm.out%d" % (self.L - 1), (BT, 2 * H))
        gout16 = self.gru.out16[self.L - 1]
        dlout = self._ln_fc_bwd(dpred, lout, BT)
        dpz = ws.get("dpz16", (BT, H), dtype=b16)
        self.lstm.backward(dlout, ws, "lstm", need_dx=True, relu_of=pa, dx16=dpz)      # includes the ReLU of proj
        fp.queue_tn(dpz, dict(b=cat, c=fp.g("proj.net.0.weight"), db=fp.g("proj.net.0.bias")))
        # d(cat): the input-projection / cond_output share and the time-encoding share (first and last H columns) feed
        # matrix products only: one bf16 product over their rows of proj's transposed weight; the t / step encodings'
        # share is needed summed over the rows only (enc_grad_b16 below)
        dxt = ws.get("dxt16", (BT, 2 * H), dtype=b16)
        ops.mm16_nt(dpz, im["proj"]["tr2"], None, dxt)
        dxh, dtenc = dxt[:, :H], dxt[:, H:]
        gx = ws.get("gx16", (BT, 2 * H + im["coin"]["widths"][1]), dtype=b16)
        fp.queue_tn(dxh, dict(b=gx[:, 2 * H:], c=fp.g("input_proj.net.0.weight"), db=fp.g("input_proj.net.0.bias")))
        # the t / step encodings have parameters in their linear halves only, and d(cat) of those columns is needed
        # only summed over the rows (plain: bias, weighted by t / i: weight): three weighted column sums of dpz times
        # proj's weight columns, instead of a [BT, 2H] product and its column sums
        fp.defer(ops.enc_grad_b16, dpz, t1, i1, fp.p("proj.net.0.weight"), H, 2 * H, Lh, 1.0, 1.0 / self.i_max,
                 ws.get("enc_sums", (3 * H,)), fp.g("t_enc.linear_proj.bias"), fp.g("t_enc.linear_proj.weight").view(-1),
                 fp.g("i_enc.linear_proj.bias"), fp.g("i_enc.linear_proj.weight").view(-1))
        if self.on_first_bucket is not None:
            fp.flush_tn()                   # proj's weight gradient is among the queued products: it belongs to this bucket
            self.on_first_bucket()          # gradients of input_proj .. fc (the head of the flat buffer) are all enqueued
        dta = ws.get("dta16", (BT, H), dtype=b16)
        ops.mm16_nt(dtenc, im["te2"]["tr"], None, dta, relu_of=ta)
        fp.queue_tn(dtenc, dict(b=ta, c=fp.g("time_encode.2.weight"), db=fp.g("time_encode.2.bias")))
        if "te0" in im:
            fp.queue_tn(dta, dict(b=ws.get("ti16", (BT, self.time_dim), dtype=b16), c=fp.g("time_encode.0.weight"),
                                  db=fp.g("time_encode.0.bias")))
        else:
            fp.defer(ops.linear_bwd_weight, prec, dta, ti2, fp.g("time_encode.0.weight"), db=fp.g("time_encode.0.bias"))
        dgout = ws.get("dgout", (BT, 2 * H))
        ops.mm16_nt(dxh, im["co"]["tr"], None, dgout)
        fp.queue_tn(dxh, dict(b=gout16, c=fp.g("cond_output.weight"), db=fp.g("cond_output.bias")))
        demb = self.gru.backward(dgout, ws, "gru", need_dx=True, dx16=ws.get("demb16", (BT, self.gru.in_l(0)), dtype=b16))
        self.emb.backward(demb, ws, "emb")
        fp.join()

    # -- reverse diffusion --------------------------------------------------
    def sample_loop(self, x, t, time_info, cond, z_all, betas, alphas, keep=None, streams=None, graph=None):
        """The reverse-diffusion loop of reference DIFF.py:340-351, in place on x [B,T,K].
        z_all[s] is the noise of step s (a [steps, B, T, K] tensor, or a CounterNoise that draws each step's rows on
        the device when the step runs).  Everything that does not depend on x or on the
        step (cond path, t_enc, time_encode and their share of `proj`) is evaluated once;
        i_enc's share of `proj` is a [steps, H] table used as a per-step bias.

        Sequences are independent, and at sampling batch sizes the sequence kernels are bound by the latency of
        their step-to-step chain, not by throughput: the batch is therefore cut into ``streams`` contiguous chunks
        (default SAMPLE_STREAMS, chunks of at least 256 sequences) that walk the loop on HIP streams of their own,
        so that one chunk's sequence kernel runs beside another chunk's matrix products.  A row's result does not
        depend on the rows it is computed with, so the chunking changes no value."""
        B, T, K = x.shape
        S = z_all.shape[0]
        self.lstm.pack_fwd_only()
        self.gru.pack_fwd_only()
        b32, a32 = betas.float().cpu(), alphas.float().cpu()
        coef = ((1 / (1 - b32).sqrt()).tolist(), (b32 / (1 - a32).sqrt()).tolist(), b32.sqrt().tolist())
        nch = SAMPLE_STREAMS if streams is None else int(streams)
        nch = max(1, min(nch, B // 256))
        if keep is not None:
            st = self._sample_setup(x, t, time_info, cond, z_all, self.ws)
            for s in range(S - 1, -1, -1):
                self._sample_step(st, s, coef)
                if s in keep:
                    keep[s] = x.clone()
            return x
        use_graph = (SAMPLE_GRAPH if graph is None else bool(graph)) and ops.tracing() is None
        if not use_graph:
            while len(self._chunks) < nch:
                self._chunks.append(self._new_chunk(len(self._chunks)))
            chunks = self._chunks[:nch] if nch > 1 else [(None, self.ws, None)]
            self._sample_run(x, t, time_info, cond, z_all, coef, chunks)
            return x
        # the whole loop (nch chunks x S steps x 8 launches) as ONE HIP graph per (batch shape, chunking): the host
        # enqueues a single graph launch instead of thousands of kernels (for callers whose host thread is busy).
        # A graph holds the addresses of everything it touches: its inputs are copied into buffers of its own and
        # its chunks work in workspaces of their own, which nothing else resizes.
        counter = isinstance(z_all, CounterNoise)
        key = (B, T, K, S, nch, self.prec, tuple(time_info.shape), tuple(cond.shape), z_all.key() if counter else None)
        ent = self._graphs.get(key)
        args = dict(x=x, t=t, time_info=time_info, cond=cond)
        if not counter:
            args["z"] = z_all
        if ent is None:
            static = {k: torch.empty_like(v) for k, v in args.items()}
            for k, v in args.items():
                static[k].copy_(v)
            chunks = [self._new_chunk(c) for c in range(nch)] if nch > 1 else [(None, Workspace(self.device), None)]
            run = lambda: self._sample_run(static["x"], static["t"], static["time_info"], static["cond"],  # noqa: E731
                                           z_all if counter else static["z"], coef, chunks)
            run()                               # sizes the workspaces (and loads the code objects) outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            ent = self._graphs[key] = (g, static, chunks)
        g, static, _ = ent
        for k, v in args.items():
            static[k].copy_(v)
        g.replay()
        x.copy_(static["x"])
        return x

    def _new_chunk(self, i):
        """Chunk i of a sampler call: a stream from the process-wide pool of its device (a process has few hardware
        queues -- four by default: streams of their own per engine made the second model of a process share queues,
        and its two-chunk loop ran at 39.6 instead of 25.4 ms per 1024 sequences), a workspace and an event of its own."""
        pool = _CHUNK_STREAMS.setdefault(str(self.device), [])
        while len(pool) <= i:
            pool.append(torch.cuda.Stream(device=self.device))
        return pool[i], Workspace(self.device), torch.cuda.Event()

    def _sample_run(self, x, t, time_info, cond, z_all, coef, chunks):
        """Enqueue the reverse loop for x [B,T,K] (in place): one chunk (stream, workspace, event) per contiguous
        share of the batch, forked from and joined to the current stream; a single chunk runs on the current stream."""
        B = x.shape[0]
        S = z_all.shape[0]
        nch = len(chunks)
        if nch == 1:
            st = self._sample_setup(x, t, time_info, cond, z_all, chunks[0][1])
            for s in range(S - 1, -1, -1):
                self._sample_step(st, s, coef)
            return
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        states = []
        for c, (stream, ws, _) in enumerate(chunks):
            lo, hi = (B * c) // nch, (B * (c + 1)) // nch
            stream.wait_event(fork)
            with torch.cuda.stream(stream):
                zc = z_all.rows_from(lo * x.shape[1]) if isinstance(z_all, CounterNoise) else z_all[:, lo:hi]
                states.append(self._sample_setup(x[lo:hi], t[lo:hi], time_info[lo:hi], cond[lo:hi], zc, ws))
        for s in range(S - 1, -1, -1):
            for c in range(nch):
                with torch.cuda.stream(chunks[c][0]):
                    self._sample_step(states[c], s, coef)
        for stream, _, done in chunks:
            done.record(stream)
            main.wait_event(done)

    def _sample_setup(self, x, t, time_info, cond, z_all, ws):
        """Step-invariant part of the sampler for the sequences x [B,T,K] (a contiguous view), on the current stream."""
        fp, prec, H, K = self.fp, self.prec, self.H, self.K
        B, T, _ = x.shape
        BT = B * T
        S = z_all.shape[0]
        x2 = x.view(BT, K)
        cat = ws.get("s.cat", (BT, 4 * H))
        emb = self.emb.forward(cond.reshape(BT, self.F), ws, "emb")
        gout = self.gru.forward(emb, B, T, ws, "gru", False)
        ops.linear_fwd(prec, gout, fp.p("cond_output.weight"), fp.p("cond_output.bias"), cat[:, :H])
        ops.posenc_fwd(t.reshape(BT), 1, BT, H, 1.0, fp.p("t_enc.linear_proj.weight"), fp.p("t_enc.linear_proj.bias"),
                       self.scale_t, self.shift_t, cat[:, H:2 * H])
        ops.zero_(cat[:, 2 * H:3 * H])
        ta = ws.get("ta", (BT, H))
        ops.linear_fwd(prec, time_info.reshape(BT, self.time_dim), fp.p("time_encode.0.weight"), fp.p("time_encode.0.bias"),
                       ta, act=ACT_RELU)
        ops.linear_fwd(prec, ta, fp.p("time_encode.2.weight"), fp.p("time_encode.2.bias"), cat[:, 3 * H:])
        pinv = ws.get("s.pinv", (BT, H))
        wp = fp.p("proj.net.0.weight")
        ops.linear_fwd(prec, cat, wp, fp.p("proj.net.0.bias"), pinv)
        steps_f = ws.get("s.steps", (S,))
        torch.arange(S, out=steps_f)
        ie = ws.get("s.ie", (S, H))
        ops.posenc_fwd(steps_f, 1, S, H, 1.0 / self.i_max, fp.p("i_enc.linear_proj.weight"), fp.p("i_enc.linear_proj.bias"),
                       self.scale_i, self.shift_i, ie)
        V = ws.get("s.V", (S, H))
        ops.linear_fwd(prec, ie, wp[:, 2 * H:3 * H], None, V)
        xh = ws.get("s.xh", (BT, H))
        fused = SAMPLE_FUSED_TAIL and ops.sampler_tail_supported(2 * H, K, H)
        if fused:       # the tail of every step leaves input_proj of the updated x behind; the first step needs it up front
            ops.linear_fwd(prec, x2, fp.p("input_proj.net.0.weight"), fp.p("input_proj.net.0.bias"), xh)
        # bf16 mode: what only the step's products read (the input projection, `proj`'s result) is kept as bf16
        b16 = self.mm16 and fused
        xh16 = pa16 = None
        if b16:
            self.imgs.pack()
            xh16 = ws.get("s.xh16", (BT, H), dtype=torch.bfloat16)
            pa16 = ws.get("s.pa16", (BT, H), dtype=torch.bfloat16)
            ops.cvt_bf16(xh, xh16)
        # input_proj has K inputs: its output's share of `proj`, x_h W_1^T, is the K-column product x (W_1 W_in)^T + W_1 b_in.
        # The tail of every step then leaves relu(that + V[s] + pinv) behind -- the LSTM's operand -- and the step's
        # [BT, H] x [H, H] product is gone (reference DIFF.py:203, 218).  Factors rounded to bf16 as the two products round them.
        fold = b16 and SAMPLE_FOLD
        wc = bc = None
        if fold:
            w1r = wp[:, :H].to(torch.bfloat16).float().contiguous()
            win_t = torch.cat([fp.p("input_proj.net.0.weight").to(torch.bfloat16).float().t(),
                               fp.p("input_proj.net.0.bias").view(1, H)], 0).contiguous()            # [K + 1, H]
            wcb = ws.get("s.wfold", (H, K + 1))
            ops.linear_fwd(F32, w1r, win_t, None, wcb)
            wc, bc = wcb[:, :K].contiguous(), wcb[:, K].contiguous()
            # the first step's operand the long way round
            ops.mm16_nt(xh16, self.im["proj_x"]["fwd"], V[S - 1], pa16, act=ACT_RELU, residual=pinv)
        return dict(x2=x2, B=B, T=T, ws=ws, pinv=pinv, V=V, wp=wp, z=z_all, xh=xh, pa=ws.get("pa", (BT, H)),
                    ln=ws.get("ln", (BT, 2 * H)), pred=ws.get("pred", (BT, K)), fused=fused, b16=b16, xh16=xh16, pa16=pa16,
                    fold=fold, wc=wc, bc=bc)

    def _step_noise(self, st, s):
        """Noise of reverse step s for the state's sequences: a row of the staged tensor, or drawn now."""
        z = st["z"]
        if isinstance(z, CounterNoise):
            buf = st["ws"].get("s.z", (st["B"], st["T"], self.K))
            ops.sampler_noise(z.seed, s, z.row0, st["B"] * st["T"], self.K, buf)
            return buf
        return z[s]

    def _sample_step(self, st, s, coef):
        """Reverse step s for the sequences of one ``_sample_setup`` state, on the current stream."""
        fp, prec, H = self.fp, self.prec, self.H
        x2, xh, pa, ln, pred = st["x2"], st["xh"], st["pa"], st["ln"], st["pred"]
        if st["fold"] and isinstance(st["z"], CounterNoise) and SAMPLE_TAIL_NOISE:
            # the tail draws the step's noise itself (the same values, no tensor, no launch)
            z = st["z"]
            lout = self.lstm.forward(st["pa16"], st["B"], st["T"], st["ws"], "lstm", False)
            ops.sampler_tail_proj_cn(prec, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), fp.p("fc.weight"), fp.p("fc.bias"),
                                     x2, z.seed, s, z.row0, coef[0][s], coef[1][s], coef[2][s], s == 0, st["wc"], st["bc"],
                                     st["V"][max(s - 1, 0)], st["pinv"], st["pa16"])
            return
        zs = self._step_noise(st, s)
        if st["fold"]:
            lout = self.lstm.forward(st["pa16"], st["B"], st["T"], st["ws"], "lstm", False)
            ops.sampler_tail_proj(prec, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), fp.p("fc.weight"), fp.p("fc.bias"),
                                  x2, zs, coef[0][s], coef[1][s], coef[2][s], s == 0, st["wc"], st["bc"], st["V"][max(s - 1, 0)],
                                  st["pinv"], st["pa16"])
            return
        if st["b16"]:
            ops.mm16_nt(st["xh16"], self.im["proj_x"]["fwd"], st["V"][s], st["pa16"], act=ACT_RELU, residual=st["pinv"])
            lout = self.lstm.forward(st["pa16"], st["B"], st["T"], st["ws"], "lstm", False)
            ops.sampler_tail(prec, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), fp.p("fc.weight"), fp.p("fc.bias"),
                             x2, zs, coef[0][s], coef[1][s], coef[2][s], s == 0, fp.p("input_proj.net.0.weight"),
                             fp.p("input_proj.net.0.bias"), st["xh16"])
            return
        if not st["fused"]:
            ops.linear_fwd(prec, x2, fp.p("input_proj.net.0.weight"), fp.p("input_proj.net.0.bias"), xh)
        ops.linear_fwd(prec, xh, st["wp"][:, :H], st["V"][s], pa, act=ACT_RELU, residual=st["pinv"])
        lout = self.lstm.forward(pa, st["B"], st["T"], st["ws"], "lstm", False)
        if st["fused"]:
            # LayerNorm -> fc -> update of x -> input_proj of the updated x for the next step, one pass over lout
            ops.sampler_tail(prec, lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), fp.p("fc.weight"), fp.p("fc.bias"),
                             x2, zs, coef[0][s], coef[1][s], coef[2][s], s == 0, fp.p("input_proj.net.0.weight"),
                             fp.p("input_proj.net.0.bias"), xh)
            return
        ops.layernorm_fwd(lout, fp.p("layer_norm.weight"), fp.p("layer_norm.bias"), ln, None)
        ops.linear_fwd(prec, ln, fp.p("fc.weight"), fp.p("fc.bias"), pred)
        ops.sampler_update(x2, pred, zs, coef[0][s], coef[1][s], coef[2][s], s == 0)

    # -- get_loss + backward ------------------------------------------------
    def loss_and_grads(self, x0, t, i_idx, noise, time_info, cond, alphas, loss_sum, global_numel=None):
        """add_noise -> forward -> MSE -> backward.  loss_sum[0] += sum of squared errors of
        this rank's slice; gradients are scaled by 2/global_numel so that a sum over ranks
        equals the gradient of the global mean loss."""
        B, T, K = x0.shape
        xn = self.ws.get("x_noisy", (B, T, K))
        i_f = self.ws.get("i_f", (B, T))
        ops.add_noise(x0, noise, i_idx, alphas, xn, i_f)
        pred = self.forward(xn, t, i_f, time_info, cond, train=True)
        dpred = self.ws.get("dpred", (B, T, K))
        n = global_numel if global_numel is not None else pred.numel()
        ops.mse_loss(pred, noise, 2.0 / n, dpred, loss_sum)
        self.backward(dpred)
        return pred


# ---------------------------------------------------------------------------
class VAEEngine:
    """Launch sequences of DeapStack.encoder/decoder/forward (reference VAE.py:152-220), the
    mixed masked loss (VAE.py:222-253, 350-352) and their backward."""

    def __init__(self, module, device, dtype="fp32"):
        self.prec = prec_of(dtype)
        self.device = device
        c = module.cfg
        self.F, self.H, self.E, self.L, self.lat = c["input_size"], c["hidden_size"], c["emb_dim"], c["num_layers"], c["lat_dim"]
        self.nb, self.nc, self.nn, self.cards = c["n_bins"], c["n_cats"], c["n_nums"], list(c["cards"])[:c["n_cats"]]
        nb, nc, nn = self.nb, self.nc, self.nn
        enc_dirs = [("encoder_mu.", "_l{l}"), ("encoder_logvar.", "_l{l}")]
        layout = EmbBlock.layout("Emb.", nb + nc, nn)
        layout += RnnStack.layout(enc_dirs, self.L)
        layout += [(n, False) for n in ("fc_mu.weight", "fc_mu.bias", "fc_logvar.weight", "fc_logvar.bias",
                                        "decoder_mlp.0.weight", "decoder_mlp.0.bias", "decoder_mlp.2.weight",
                                        "decoder_mlp.2.bias")]
        head_w, head_b = [], []
        if nb:
            head_w.append("bins_linear.weight"); head_b.append("bins_linear.bias")
        for j in range(nc):
            head_w.append(f"cats_linears.{j}.weight"); head_b.append(f"cats_linears.{j}.bias")
        if nn:
            head_w.append("nums_linear.weight"); head_b.append("nums_linear.bias")
        layout += [(n, k > 0) for k, n in enumerate(head_w)] + [(n, k > 0) for k, n in enumerate(head_b)]
        # the conditioning branch of the decoder comes last: without a cond it receives no gradient and the
        # optimizer step stops in front of it (torch.optim.Adam skips parameters whose grad is None)
        first_cond = len(layout)
        layout += EmbBlock.layout("Emb_decoder.", nb + nc, nn)
        layout += [("Emb_hidden_decoder.weight", False), ("Emb_hidden_decoder.bias", False)]
        self.head_w0, self.head_b0 = head_w[0], head_b[0]
        self.NH = nb + sum(self.cards) + nn
        inactive = ["time_encode.0.weight", "time_encode.0.bias", "time_encode.2.weight", "time_encode.2.bias"]
        self.fp = FlatParams(module, layout, inactive, device)
        self.n_without_cond = self.fp.offsets[layout[first_cond][0]]
        self.enc = RnnStack(self.fp, CELL_GRU, self.H, self.F, self.L, enc_dirs, "heads", self.prec)
        self.emb = EmbBlock(self.fp, "Emb.", nb, nc, nn, self.cards, self.E, self.prec)
        self.emb_dec = EmbBlock(self.fp, "Emb_decoder.", nb, nc, nn, self.cards, self.E, self.prec)
        # bf16 mode: the embedding blocks and the two latent projections' backward take the bf16-stored-operand products
        self.imgs = W16Images(self.fp)
        # (the encoder's embedding block hands its result to the GRUs as bfloat16, padded to a multiple of 8 columns)
        self.emb.use_mm16(self.imgs, pad_last=all(self.enc.mm16))
        self.emb_dec.use_mm16(self.imgs)
        self.im = {}
        if MM16 and self.prec == BF16 and self.H % 8 == 0 and all(self.enc.mm16):
            self.im = dict(fc_mu=self.imgs.add(self.fp.p("fc_mu.weight"), fwd=True, tr=True),
                           fc_logvar=self.imgs.add(self.fp.p("fc_logvar.weight"), fwd=True, tr=True))
            if VAE_DEC16:
                # the decoder's three products (reference VAE.py:200-219) on bf16-stored operands as well: z padded to 8 columns,
                # the fused head outputs to a multiple of 8 (the spare columns meet zero weights and are never read)
                self.NHp = -(-self.NH // 8) * 8
                self.im["d0"] = self.imgs.add(self.fp.p("decoder_mlp.0.weight"), fwd=True, tr=True)
                self.im["d2"] = self.imgs.add(self.fp.p("decoder_mlp.2.weight"), fwd=True, tr=True)
                self.im["hd"] = self.imgs.add(self.fp.pgroup(self.head_w0, self.NH, self.H), fwd=True, tr=True)
                self.im["hd_bias"] = self.imgs.add_padded_bias(self.fp.pgroup(self.head_b0, 1, self.NH).view(-1), self.NHp)
        self.ws = Workspace(device)

    def encoder(self, inp, eps, train=False, tag="enc", kld_sum=None):
        fp, ws, prec, H, Lt = self.fp, self.ws, self.prec, self.H, self.lat
        B, T, _ = inp.shape
        BT = B * T
        fp.defer(self.enc.pack if train else self.enc.pack_fwd_only)       # beside the embedding block
        x2 = inp.reshape(BT, self.F)
        self._e16 = bool(self.emb.mm16 and (self.emb.last16 or self.emb.pad16) and self.enc.mm16[0])
        e = self.emb.forward(x2, ws, "emb", want16=self._e16)
        fp.join()
        hh = self.enc.forward(e, B, T, ws, tag, train)
        mu = ws.get("mu", (BT, Lt))
        lv = ws.get("lv", (BT, Lt))
        h16 = self.enc.out16[-1]
        if "fc_mu" in self.im and h16 is not None:
            # the bf16 copy of h that the sequence kernel wrote for the weight gradients holds the operand values of the
            # bf16 mode: half the bytes of the fp32 rows
            self.imgs.pack()
            ops.linear_fwd_b16x(h16[:, :H], self.im["fc_mu"]["fwd"][:Lt, :H], fp.p("fc_mu.bias"), mu)
            ops.linear_fwd_b16x(h16[:, H:], self.im["fc_logvar"]["fwd"][:Lt, :H], fp.p("fc_logvar.bias"), lv)
        else:
            ops.linear_fwd(prec, hh[:, :H], fp.p("fc_mu.weight"), fp.p("fc_mu.bias"), mu)
            ops.linear_fwd(prec, hh[:, H:], fp.p("fc_logvar.weight"), fp.p("fc_logvar.bias"), lv)
        z = ws.get("z", (BT, Lt))
        eps2 = eps.reshape(BT, Lt)
        ops.vae_reparam_fwd(mu, lv, eps2, z, kld_sum)
        self._enc_saved = (hh, eps2, B, T)
        return z.view(B, T, Lt), mu.view(B, T, Lt), lv.view(B, T, Lt)

    def _dheads(self, ws, BT):
        """Gradient of the loss with respect to the head outputs, with the row stride of the head outputs (the loss kernel
        walks both with one stride); the pad columns of the bf16-stored form stay zero."""
        if "d0" in self.im:
            return ws.get("dheads_p", (BT, self.NHp), zero="new")[:, :self.NH]
        return ws.get("dheads", (BT, self.NH))

    def decoder(self, z, mask, cond, target=None, scales=None, loss_parts=None, train=False):
        """z [B,T,L], mask [B,T,F] float, cond [B,T,F] or None -> fused head outputs [B*T, NH]
        (bins | cats | nums, reference-visible values).  With ``target``/``scales``/``loss_parts``
        the loss partial sums are accumulated and, if train, d(loss)/d(head pre-activations)
        is left in the 'dheads' workspace buffer."""
        fp, ws, prec, H = self.fp, self.ws, self.prec, self.H
        B, T, Lt = z.shape
        BT = B * T
        z2 = z.reshape(BT, Lt)
        m2 = mask.reshape(BT, self.F)
        ce = None
        if "d0" in self.im:
            # d1, lat and their gradients are read by matrix products only: bf16 where they are produced
            b16 = torch.bfloat16
            im = self.im
            self.imgs.pack()
            Lp = -(-Lt // 8) * 8
            z16 = ws.get("z16", (BT, Lp), dtype=b16, zero="new")        # the pad columns stay zero
            ops.cvt_bf16(z2, z16[:, :Lt])
            d1 = ws.get("d1_16", (BT, H), dtype=b16)
            ops.mm16_nt(z16, im["d0"]["fwd"], fp.p("decoder_mlp.0.bias"), d1, act=ACT_RELU)
            lat = ws.get("lat16", (BT, H), dtype=b16)
            if cond is not None:
                ce = self.emb_dec.forward(cond.reshape(BT, self.F), ws, "embdec")
                latc = ws.get("latc", (BT, H))
                ops.linear_fwd(prec, ce, fp.p("Emb_hidden_decoder.weight"), fp.p("Emb_hidden_decoder.bias"), latc)
                ops.mm16_nt(d1, im["d2"]["fwd"], fp.p("decoder_mlp.2.bias"), lat, residual=latc)
            else:
                ops.mm16_nt(d1, im["d2"]["fwd"], fp.p("decoder_mlp.2.bias"), lat)
            heads_p = ws.get("heads_p", (BT, self.NHp))
            ops.mm16_nt(lat, im["hd"]["fwd"], im["hd_bias"], heads_p)
            heads = heads_p[:, :self.NH]
        else:
            d1 = ws.get("d1", (BT, H))
            ops.linear_fwd(prec, z2, fp.p("decoder_mlp.0.weight"), fp.p("decoder_mlp.0.bias"), d1, act=ACT_RELU)
            lat = ws.get("lat", (BT, H))
            ops.linear_fwd(prec, d1, fp.p("decoder_mlp.2.weight"), fp.p("decoder_mlp.2.bias"), lat)
            if cond is not None:
                ce = self.emb_dec.forward(cond.reshape(BT, self.F), ws, "embdec")
                ops.linear_fwd(prec, ce, fp.p("Emb_hidden_decoder.weight"), fp.p("Emb_hidden_decoder.bias"), lat, accumulate=True)
            heads = ws.get("heads", (BT, self.NH))
            ops.linear_fwd(prec, lat, fp.pgroup(self.head_w0, self.NH, H), fp.pgroup(self.head_b0, 1, self.NH), heads)
        dheads = self._dheads(ws, BT) if train else None
        t2 = target.reshape(BT, self.F) if target is not None else m2
        s = scales if scales is not None else (0.0, 0.0, 0.0)
        ops.vae_heads_loss(heads, m2, t2, self.nb, self.nc, self.nn, self.cards, s[0], s[1], s[2], dheads, loss_parts)
        self._dec_saved = (z2, ce, B, T)
        return heads

    def split_heads(self, heads, B, T):
        """Fused [B*T, NH] head buffer -> the reference's dict {'bins','cats','nums'} (copies)."""
        out = {}
        o = 0
        if self.nb:
            out["bins"] = heads[:, o:o + self.nb].reshape(B, T, self.nb).clone()
            o += self.nb
        if self.nc:
            out["cats"] = []
            for c in self.cards:
                out["cats"].append(heads[:, o:o + c].reshape(B, T, c).clone())
                o += c
        if self.nn:
            out["nums"] = heads[:, o:o + self.nn].reshape(B, T, self.nn).clone()
        return out

    def loss_scales(self, global_rows):
        """(scale_bce, scale_ce, scale_mse): d(mean-reduced loss terms)/d(sum terms) for a global
        batch of ``global_rows`` = B*T rows (VAE.py:241-251)."""
        s_bce = 1.0 / (global_rows * self.nb) if self.nb else 0.0
        s_ce = 1.0 / (global_rows * self.nc) if self.nc else 0.0
        s_mse = 2.0 / (global_rows * self.nn) if self.nn else 0.0
        return s_bce, s_ce, s_mse

    def total_loss(self, parts, kld_sum, beta, rows):
        """Host-side combination of the partial sums (parts: [2+nc] list, kld_sum float)."""
        bce = parts[0] / (rows * self.nb) if self.nb else 0.0
        mse = parts[1] / (rows * self.nn) if self.nn else 0.0
        ce = sum(parts[2:2 + self.nc]) / (rows * self.nc) if self.nc else 0.0
        kld = -0.5 * kld_sum / (rows * self.lat)
        return mse + bce + ce + beta * kld, dict(bce=bce, ce=ce, mse=mse, kld=kld)

    def backward(self, beta, global_rows):
        """Backward of loss = mse + bce + ce + beta*kld through decoder and encoder (after
        encoder(train=True) and decoder(train=True))."""
        fp, ws, prec, H, Lt = self.fp, self.ws, self.prec, self.H, self.lat
        z2, ce, B, T = self._dec_saved
        hh, eps2, _, _ = self._enc_saved
        BT = B * T
        dheads = self._dheads(ws, BT)
        mu = ws.get("mu", (BT, Lt))
        lv = ws.get("lv", (BT, Lt))
        dz = ws.get("dz", (BT, Lt))
        if "d0" in self.im:
            b16 = torch.bfloat16
            im = self.im
            Lp = -(-Lt // 8) * 8
            lat = ws.get("lat16", (BT, H), dtype=b16)
            d1 = ws.get("d1_16", (BT, H), dtype=b16)
            z16 = ws.get("z16", (BT, Lp), dtype=b16)
            dh16 = ws.get("dheads16", (BT, self.NHp), dtype=b16)
            ops.cvt_bf16(ws.get("dheads_p", (BT, self.NHp)), dh16)
            fp.queue_tn(dh16, dict(b=lat, c=fp.ggroup(self.head_w0, self.NH, H), db=fp.ggroup(self.head_b0, 1, self.NH).view(-1),
                                   a0=0, na=self.NH))
            dlat = ws.get("dlat16", (BT, H), dtype=b16)
            if ce is not None:
                # the conditioning branch reads d(lat) through the fp32-operand products: an fp32 copy for it
                dlat32 = ws.get("dlat", (BT, H))
                ops.mm16_nt(dh16, im["hd"]["tr"], None, dlat32)
                ops.cvt_bf16(dlat32, dlat)
                dce = ws.get("dce", (BT, self.F))
                _lin_bwd(fp, prec, dlat32, ce, fp.p("Emb_hidden_decoder.weight"), fp.g("Emb_hidden_decoder.weight"),
                         fp.g("Emb_hidden_decoder.bias"), dx=dce)
                self.emb_dec.backward(dce, ws, "embdec")
            else:
                ops.mm16_nt(dh16, im["hd"]["tr"], None, dlat)
            fp.queue_tn(dlat, dict(b=d1, c=fp.g("decoder_mlp.2.weight"), db=fp.g("decoder_mlp.2.bias")))
            dd1 = ws.get("dd1_16", (BT, H), dtype=b16)
            ops.mm16_nt(dlat, im["d2"]["tr"], None, dd1, relu_of=d1)
            fp.queue_tn(dd1, dict(b=z16, c=fp.g("decoder_mlp.0.weight"), db=fp.g("decoder_mlp.0.bias")))
            dzp = ws.get("dz_p", (BT, Lp))
            ops.mm16_nt(dd1, im["d0"]["tr"], None, dzp)
            ops.copy_2d(dzp[:, :Lt], dz)
        else:
            lat = ws.get("lat", (BT, H))
            d1 = ws.get("d1", (BT, H))
            dlat = ws.get("dlat", (BT, H))
            _lin_bwd(fp, prec, dheads, lat, fp.pgroup(self.head_w0, self.NH, H), fp.ggroup(self.head_w0, self.NH, H),
                     fp.ggroup(self.head_b0, 1, self.NH), dx=dlat)
            if ce is not None:
                dce = ws.get("dce", (BT, self.F))
                _lin_bwd(fp, prec, dlat, ce, fp.p("Emb_hidden_decoder.weight"), fp.g("Emb_hidden_decoder.weight"),
                         fp.g("Emb_hidden_decoder.bias"), dx=dce)
                self.emb_dec.backward(dce, ws, "embdec")
            dd1 = ws.get("dd1", (BT, H))
            _lin_bwd(fp, prec, dlat, d1, fp.p("decoder_mlp.2.weight"), fp.g("decoder_mlp.2.weight"), fp.g("decoder_mlp.2.bias"),
                     dx=dd1, relu_of=d1)
            _lin_bwd(fp, prec, dd1, z2, fp.p("decoder_mlp.0.weight"), fp.g("decoder_mlp.0.weight"), fp.g("decoder_mlp.0.bias"), dx=dz)
        dmu = ws.get("dmu", (BT, Lt))
        dlv = ws.get("dlv", (BT, Lt))
        ops.vae_reparam_bwd(dz, mu, lv, eps2, beta / (global_rows * Lt), dmu, dlv)
        dhh = ws.get("dhh", (BT, 2 * H))
        o16 = self.enc.out16[self.L - 1] if self.im else None
        if o16 is not None:
            # Linear(H -> lat): a weight gradient of lat rows over all B*T samples and an input gradient of reduction
            # length lat; as bf16-stored products they read [dmu | dlogvar] (rounded, each padded to 8 columns) and the
            # bf16 copy of the encoder's output
            Lp = -(-Lt // 8) * 8
            self.imgs.pack()
            d16 = ws.get("dml16", (BT, 2 * Lp), dtype=torch.bfloat16, zero="new")
            ops.cvt_bf16(dmu, d16[:, :Lt])
            ops.cvt_bf16(dlv, d16[:, Lp:Lp + Lt])
            fp.queue_tn(d16, dict(b=o16[:, :H], c=fp.g("fc_mu.weight"), db=fp.g("fc_mu.bias"), a0=0, na=Lt))
            fp.queue_tn(d16, dict(b=o16[:, H:], c=fp.g("fc_logvar.weight"), db=fp.g("fc_logvar.bias"), a0=Lp, na=Lt))
            ops.mm16_nt(d16[:, :Lp], self.im["fc_mu"]["tr"], None, dhh[:, :H])
            ops.mm16_nt(d16[:, Lp:], self.im["fc_logvar"]["tr"], None, dhh[:, H:])
        else:
            _lin_bwd(fp, prec, dmu, hh[:, :H], fp.p("fc_mu.weight"), fp.g("fc_mu.weight"), fp.g("fc_mu.bias"), dx=dhh[:, :H])
            _lin_bwd(fp, prec, dlv, hh[:, H:], fp.p("fc_logvar.weight"), fp.g("fc_logvar.weight"), fp.g("fc_logvar.bias"), dx=dhh[:, H:])
        # (the gradient of the embedding block's bfloat16 result leaves the input-gradient product as bfloat16)
        dx16 = ws.get("enc.dx16", (BT, self.enc.in_p(0)), dtype=torch.bfloat16) if self._e16 else None
        de = self.enc.backward(dhh, ws, "enc", need_dx=True, dx16=dx16)
        self.emb.backward(de, ws, "emb")
        fp.join()
