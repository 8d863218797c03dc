"""TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's post-hoc evaluation networks -- the GRU
discriminator of Evaluation/Metrics.py:54-65 with its loss (Metrics.py:101-106) and the one-layer GRU predictor of
Evaluation/predictive_metrics.py:8-17 with its L1 loss (:27, 38-40).  Imported by tests, smoke() and the bench's
CPU baseline only; the product path never touches it."""
import numpy as np

from . import ops
from .rnn import rnn_bwd, rnn_fwd


def disc_forward(sd, x, num_layers):
    """x [B,T,dim] -> (logits [B,1], (caches, out))  -- Discriminator.forward, Metrics.py:60-64: fc of the last layer's h_T."""
    out, caches = rnn_fwd("gru", sd, "RNN.", x, num_layers, False)
    logit = ops.linear_fwd(out[:, -1, :], sd["fc.weight"], sd["fc.bias"])
    return logit, (caches, out)


def disc_loss_and_grads(sd, x_real, x_fake, num_layers=2):
    """d_loss = BCEWithLogits(D(real), 1) + BCEWithLogits(D(fake), 0), each a mean over its batch (Metrics.py:101-106).
    Returns (loss, logits_real, logits_fake, grads by torch name)."""
    x = np.concatenate([x_real, x_fake], 0).astype(np.float32)
    nr, nf = len(x_real), len(x_fake)
    y = np.concatenate([np.ones((nr, 1), np.float32), np.zeros((nf, 1), np.float32)], 0)
    w = np.concatenate([np.full((nr, 1), 1.0 / nr, np.float32), np.full((nf, 1), 1.0 / nf, np.float32)], 0)
    logit, (caches, out) = disc_forward(sd, x, num_layers)
    z = logit.astype(np.float64)
    bce = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))                 # torch's stable form
    loss = float((bce * w).sum())
    dlogit = ((ops.sigmoid(logit) - y) * w).astype(np.float32)
    dh, dW, db = ops.linear_bwd(out[:, -1, :], sd["fc.weight"], dlogit)
    grads = {"fc.weight": dW, "fc.bias": db}
    dout = np.zeros_like(out)
    dout[:, -1, :] = dh
    rnn_bwd("gru", sd, "RNN.", caches, dout, num_layers, False, grads, need_dx=False)
    return loss, logit[:nr], logit[nr:], grads


def pred_forward(sd, x):
    """x [B,T-1,dim-1] -> predictions [B,T-1,1] -- PosthocRNN.forward, predictive_metrics.py:14-17."""
    out, caches = rnn_fwd("gru", sd, "rnn.", x, 1, False)
    y = ops.linear_fwd(out, sd["fc.weight"], sd["fc.bias"])
    return y, (caches, out)


def pred_loss_and_grads(sd, x, y):
    """nn.L1Loss() (mean) of the predictor and its gradients (predictive_metrics.py:36-41)."""
    x = x.astype(np.float32)
    p, (caches, out) = pred_forward(sd, x)
    d = p - y
    loss = float(np.abs(d.astype(np.float64)).mean())
    dp = (np.sign(d) / d.size).astype(np.float32)
    B, T, H = out.shape
    do, dW, db = ops.linear_bwd(out.reshape(B * T, H), sd["fc.weight"], dp.reshape(B * T, 1))
    grads = {"fc.weight": dW, "fc.bias": db}
    rnn_bwd("gru", sd, "rnn.", caches, do.reshape(B, T, H), 1, False, grads, need_dx=False)
    return loss, p, grads
