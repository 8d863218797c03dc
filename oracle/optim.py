"""torch.optim.Adam (coupled L2 weight decay, no amsgrad) in numpy.

Oracle code: test infrastructure only.  Restates the update the reference's
``Adam(model.parameters())`` (DIFF.py:291) and
``Adam(ae.parameters(), lr=lr, weight_decay=weight_decay)`` (VAE.py:295) apply.
Parameters without a gradient are skipped, like torch does for ``grad is None``.
"""
import numpy as np


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.p = params                      # dict name -> array, updated in place
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = {}
        self.v = {}
        self.t = {}

    def step(self, grads):
        for k, g in grads.items():
            p = self.p[k]
            dt = p.dtype
            g = g.astype(dt).reshape(p.shape)
            if self.wd:
                g = g + dt.type(self.wd) * p
            if k not in self.m:
                self.m[k] = np.zeros_like(p)
                self.v[k] = np.zeros_like(p)
                self.t[k] = 0
            self.t[k] += 1
            t = self.t[k]
            m, v = self.m[k], self.v[k]
            m += (g - m) * dt.type(1 - self.b1)            # exp_avg.lerp_(grad, 1-beta1)
            v *= dt.type(self.b2)
            v += dt.type(1 - self.b2) * g * g
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            step_size = self.lr / bc1
            denom = np.sqrt(v) / dt.type(np.sqrt(bc2)) + dt.type(self.eps)
            p -= dt.type(step_size) * (m / denom)
