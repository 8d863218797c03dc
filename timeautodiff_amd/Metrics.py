"""Post-hoc evaluation networks of the reference on the HIP kernels: the GRU discriminator behind
``discriminative_score_metrics`` (reference Evaluation/Metrics.py:54-126) and the one-layer GRU predictor behind
``predictive_score_metrics`` (reference Evaluation/predictive_metrics.py:8-54).

Both train on minibatches of 128 windows of at most a few dozen features with hidden sizes of 50 and dim // 2:
a step is bound by launch latency, so the training step is replayed as one captured HIP graph after its first use.
Same call signatures and return values as the reference; the random draws (numpy permutations) are taken in the
reference's order.  Keyword-only extras: dtype ('fp32' | 'bf16'), losses (a list that receives the loss curve)."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .engine import FlatParams, RnnStack, Workspace, prec_of
from .ops import CELL_GRU


def train_test_divide(data_x, data_x_hat, train_rate=0.8):
    """reference Metrics.py:11-35: two numpy permutations, the first 80 % of each for training."""
    no = len(data_x)
    idx = np.random.permutation(no)
    tr, te = idx[:int(no * train_rate)], idx[int(no * train_rate):]
    train_x, test_x = data_x[tr], data_x[te]
    no = len(data_x_hat)
    idx = np.random.permutation(no)
    tr, te = idx[:int(no * train_rate)], idx[int(no * train_rate):]
    return train_x, data_x_hat[tr], test_x, data_x_hat[te]


def batch_generator(data, batch_size):
    """reference Metrics.py:37-52: the first batch_size entries of a numpy permutation."""
    idx = np.random.permutation(len(data))[:batch_size]
    return data[idx]


class PosthocEngine:
    """A unidirectional GRU stack (parameters ``<prefix>weight_ih_l0`` ...) followed by fc (hidden -> 1), either on
    the last step's hidden state (discriminator) or on every step (predictor); forward, loss, backward and Adam on
    the flat parameter buffer."""

    def __init__(self, module, prefix, hidden, in_dim, num_layers, per_step, device, dtype="fp32"):
        self.prec = prec_of(dtype)
        self.device = ops.norm_device(device)
        self.H, self.in_dim, self.L, self.per_step = hidden, in_dim, num_layers, per_step
        dirs = [(prefix, "_l{l}")]
        layout = RnnStack.layout(dirs, num_layers) + [("fc.weight", False), ("fc.bias", False)]
        self.fp = FlatParams(module, layout, [], self.device)
        self.rnn = RnnStack(self.fp, CELL_GRU, hidden, in_dim, num_layers, dirs, "heads", self.prec)
        self.ws = Workspace(self.device)
        # loss partial sums right behind the gradients (cleared with them): [0] the step's loss sum (both halves of a
        # discriminator batch when they are equally long), [2] the generated half's when the halves differ in length
        self.loss4 = self.fp.slots(4)
        self.loss = self.loss4[:1]
        self._graphs = {}

    def forward(self, x, train=False):
        """x [B,T,in] (device, float32, contiguous) -> fc output: [B,1] of the last step, or [B*T,1] of every step."""
        fp, ws, H = self.fp, self.ws, self.H
        B, T, _ = x.shape
        self.rnn.pack(need_bwd=train)
        out = self.rnn.forward(x.reshape(B * T, self.in_dim), B, T, ws, "rnn", train)
        self._saved = (out, B, T)
        if self.per_step:
            y = ws.get("y", (B * T, 1))
            ops.linear_fwd(self.prec, out, fp.p("fc.weight"), fp.p("fc.bias"), y)
        else:
            y = ws.get("y", (B, 1))
            ops.linear_fwd(self.prec, out.view(B, T, H)[:, T - 1, :], fp.p("fc.weight"), fp.p("fc.bias"), y)
        return y

    def backward(self, dy):
        fp, ws, H = self.fp, self.ws, self.H
        out, B, T = self._saved
        if self.per_step:
            dout = ws.get("dout", (B * T, H))
            ops.linear_bwd_data(self.prec, dy, fp.p("fc.weight"), dout)
            fp.defer(ops.linear_bwd_weight, self.prec, dy, out, fp.g("fc.weight"), db=fp.g("fc.bias"))
        else:
            dout = ws.get("dout", (B * T, H), zero=True)                       # only the last step receives a gradient
            last = out.view(B, T, H)[:, T - 1, :]
            ops.linear_bwd_data(self.prec, dy, fp.p("fc.weight"), dout.view(B, T, H)[:, T - 1, :])
            fp.defer(ops.linear_bwd_weight, self.prec, dy, last, fp.g("fc.weight"), db=fp.g("fc.bias"))
        self.rnn.backward(dout, ws, "rnn", need_dx=False)
        fp.join()

    # ---- one optimizer step each (everything after the draws; capturable)
    def disc_step_body(self, xb, yb, ones, n_each, captured, coef=None, lr=1e-3):
        """xb [n_real + n_fake, T, dim] = real windows then generated ones, yb = their labels: d_loss = mean BCE(real)
        + mean BCE(fake), each mean over its own count (reference Metrics.py:101-106) -> backward -> Adam.  n_each: the
        common count, or (n_real, n_fake) -- then loss4[0] / n_real + loss4[2] / n_fake is the step's loss."""
        self.fp.zero_grad()
        logit = self.forward(xb, train=True)
        dlog = self.ws.get("dlogit", tuple(logit.shape))
        nr, nf = (n_each, n_each) if isinstance(n_each, int) else n_each
        if nr == nf:
            ops.vae_heads_loss(logit, ones, yb, 1, 0, 0, [], 1.0 / nr, 0.0, 0.0, dlog, self.loss4[:2])
        else:
            ops.vae_heads_loss(logit[:nr], ones[:nr], yb[:nr], 1, 0, 0, [], 1.0 / nr, 0.0, 0.0, dlog[:nr], self.loss4[:2])
            ops.vae_heads_loss(logit[nr:], ones[nr:], yb[nr:], 1, 0, 0, [], 1.0 / nf, 0.0, 0.0, dlog[nr:], self.loss4[2:])
        self.backward(dlog)
        if captured:
            self.fp.adam_captured(lr)
        else:
            self.fp.adam(lr)

    def pred_step_body(self, xb, yb, captured, coef=None, lr=1e-3):
        """xb [n,T-1,dim-1], yb [n*(T-1),1]: L1 loss (mean) -> backward -> Adam (predictive_metrics.py:36-41)."""
        self.fp.zero_grad()
        p = self.forward(xb, train=True)
        dp = self.ws.get("dp", tuple(p.shape))
        ops.l1_loss(p, yb, 1.0 / p.numel(), dp, self.loss)
        self.backward(dp)
        if captured:
            self.fp.adam_captured(lr)
        else:
            self.fp.adam(lr)

    def replay(self, key, body, lr=1e-3):
        """Run ``body(captured, coef)`` as a captured graph (built on first use after two eager runs on saved state)."""
        ent = self._graphs.get(key)
        if ent is None:
            fp = self.fp
            keep = (fp.flat.clone(), fp.m.clone(), fp.v.clone(), fp.step)
            for _ in range(2):
                body(False, None)
            fp.flat.copy_(keep[0]); fp.m.copy_(keep[1]); fp.v.copy_(keep[2]); fp.step = keep[3]
            fp.version += 1
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body(True, None)
            ent = self._graphs[key] = (g, None)
        g, _ = ent
        self.fp.adam_before_replay()
        g.replay()


def _engine_of(module, build):
    eng = getattr(module, "_eng", None)
    if eng is None or not eng.fp.owns(module):
        eng = module._eng = build()
    return eng


class Discriminator(nn.Module):
    """reference Metrics.py:54-65: nn.GRU(input_size, hidden_size, num_layers, batch_first=True) + Linear(hidden, 1)
    on the last layer's final hidden state -> (logit, sigmoid(logit))."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.RNN = nn.GRU(input_size, hidden_size, num_layers, batch_first=True)
        self.fc = nn.Linear(hidden_size, 1)
        self._cfg = (input_size, hidden_size, num_layers)
        self._dtype = "fp32"

    def engine(self, dev=None):
        i, h, n = self._cfg
        return _engine_of(self, lambda: PosthocEngine(self, "RNN.", h, i, n, False, dev, self._dtype))

    @ops.device_scope(lambda self, x: ops._first_cuda(x, self))
    def forward(self, x):
        eng = self.engine(x.device if x.is_cuda else None)
        eng.fp.version += 1
        logit = eng.forward(x.to(device=eng.device, dtype=torch.float32).contiguous()).clone()
        return logit, ops.sigmoid_(logit.clone())


class PosthocRNN(nn.Module):
    """reference predictive_metrics.py:8-17: nn.GRU(input_dim - 1, hidden_dim, batch_first=True) + Linear(hidden, 1)
    on every step."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.rnn = nn.GRU(input_size=input_dim - 1, hidden_size=hidden_dim, batch_first=True)
        self.fc = nn.Linear(hidden_dim, 1)
        self._cfg = (input_dim - 1, hidden_dim)
        self._dtype = "fp32"

    def engine(self, dev=None):
        i, h = self._cfg
        return _engine_of(self, lambda: PosthocEngine(self, "rnn.", h, i, 1, True, dev, self._dtype))

    @ops.device_scope(lambda self, x: ops._first_cuda(x, self))
    def forward(self, x):
        eng = self.engine(x.device if x.is_cuda else None)
        eng.fp.version += 1
        B, T, _ = x.shape
        return eng.forward(x.to(device=eng.device, dtype=torch.float32).contiguous()).view(B, T, 1).clone()


def discriminative_score_metrics(ori_data, generated_data, iterations, *, dtype="fp32", losses=None, graph=True):
    """reference Metrics.py:67-126: train a 2-layer GRU(50) discriminator for ``iterations`` Adam steps on batches of
    128 real and 128 generated windows, then |0.5 - accuracy| on the held-out 20 %."""
    dev = ops.norm_device(None)
    ori = torch.as_tensor(ori_data, dtype=torch.float32).to(dev).contiguous()
    gen = torch.as_tensor(generated_data, dtype=torch.float32).to(dev).contiguous()
    no, seq_len, dim = ori.shape
    hidden_dim, batch_size, num_layers = 50, 128, 2
    disc = Discriminator(dim, hidden_dim, num_layers)
    disc._dtype = dtype
    eng = disc.engine(dev)
    train_x, train_x_hat, test_x, test_x_hat = train_test_divide(ori, gen)
    # the reference's batch_generator takes the first batch_size entries of a permutation of EACH set: the two halves
    # of a batch differ in length when one set holds fewer than batch_size training windows, and each BCE mean runs
    # over its own count (Metrics.py:96-106)
    n_real, n_fake = min(batch_size, len(train_x)), min(batch_size, len(train_x_hat))
    xb = torch.empty(n_real + n_fake, seq_len, dim, device=dev)
    yb = torch.empty(n_real + n_fake, 1, device=dev)
    ops.fill_(yb[:n_real], 1.0)
    ops.fill_(yb[n_real:], 0.0)
    ones = ops.fill_(torch.empty(n_real + n_fake, 1, device=dev), 1.0)
    hist = torch.zeros(max(iterations, 1), 4, device=dev)
    n_each = n_real if n_real == n_fake else (n_real, n_fake)
    with torch.cuda.device(dev):
        for itt in range(iterations):
            ir = torch.from_numpy(np.random.permutation(len(train_x))[:batch_size]).to(dev)
            ih = torch.from_numpy(np.random.permutation(len(train_x_hat))[:batch_size]).to(dev)
            ops.gather_rows(train_x, ir, xb[:n_real])
            ops.gather_rows(train_x_hat, ih, xb[n_real:])
            body = lambda cap, coef: eng.disc_step_body(xb, yb, ones, n_each, cap, coef)     # noqa: E731
            if graph and ops.tracing() is None:
                eng.replay("disc", body)
            else:
                body(False, None)
            hist[itt].copy_(eng.loss4)
        if losses is not None:
            h = hist[:iterations].cpu().double()
            losses.extend((h[:, 0] / n_real if n_real == n_fake else h[:, 0] / n_real + h[:, 2] / n_fake).tolist())
        _, p_real = disc(test_x)
        _, p_fake = disc(test_x_hat)
    y_pred = np.squeeze(np.concatenate((p_real.cpu().numpy(), p_fake.cpu().numpy()), axis=0))
    y_label = np.concatenate((np.ones([len(p_real), ]), np.zeros([len(p_fake), ])), axis=0)
    acc = float(np.mean((y_pred > 0.5) == (y_label > 0.5)))             # sklearn.metrics.accuracy_score
    return np.abs(0.5 - acc)


def predictive_score_metrics(ori_data, generated_data, col_pred, iterations=5000, batch_size=128, *, dtype="fp32", losses=None,
                             graph=True):
    """reference predictive_metrics.py:19-54: train GRU(dim-1 -> dim//2) + fc to predict column ``col_pred`` one step
    ahead from the other columns on generated windows (L1 loss), report the mean absolute error on the original ones
    (the mean over windows of each window's MAE)."""
    dev = ops.norm_device(None)
    ori = torch.as_tensor(ori_data, dtype=torch.float32).to(dev).contiguous()
    gen = torch.as_tensor(generated_data, dtype=torch.float32).to(dev).contiguous()
    dim = ori.shape[2]
    T = gen.shape[1]
    model = PosthocRNN(dim, int(dim / 2))
    model._dtype = dtype
    eng = model.engine(dev)
    others = [c for c in range(dim) if c != col_pred]
    gx = gen[:, :-1, others].contiguous()                                 # inputs / targets of every window, cut once
    gy = gen[:, 1:, col_pred].contiguous()
    nb = min(batch_size, len(gen))
    xb = torch.zeros(nb, T - 1, dim - 1, device=dev)
    yb = torch.zeros(nb, T - 1, device=dev)
    hist = torch.zeros(max(iterations, 1), device=dev)
    with torch.cuda.device(dev):
        for it in range(iterations):
            idx = torch.from_numpy(np.random.permutation(len(gen))[:batch_size]).to(dev)
            ops.gather_rows(gx, idx, xb)
            ops.gather_rows(gy, idx, yb)
            body = lambda cap, coef: eng.pred_step_body(xb, yb.view(-1, 1), cap, coef)      # noqa: E731
            if graph and ops.tracing() is None:
                eng.replay("pred", body)
            else:
                body(False, None)
            hist[it:it + 1].copy_(eng.loss)
        if losses is not None:
            losses.extend((hist[:iterations] / float(nb * (T - 1))).tolist())
        ox = ori[:, :-1, others].contiguous()
        oy = ori[:, 1:, col_pred]
        pred = model(ox).squeeze(2)
        mae = (pred - oy).abs().mean(1).mean().item()                     # mean over windows of the per-window MAE
    return mae
