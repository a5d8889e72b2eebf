"""CPU restatement of the reference optimizer step - test infrastructure only.

Follows: ``OneCycle`` / ``LRSchedulerStep.step`` / ``annealing_cos``
(tools/train_utils/optimization/learning_schedules_fastai.py:13-77), ``build_optimizer`` 'adam_onecycle'
(tools/train_utils/optimization/__init__.py:19-32: Adam betas (0.9, 0.99), wd = WEIGHT_DECAY, true_wd,
bn_wd), ``OptimWrapper.step`` (tools/train_utils/optimization/fastai_optim.py:135-152: p *= 1 - wd*lr for
every trainable parameter incl. BN, then torch Adam with weight_decay forced to 0) and
``clip_grad_norm_(params, GRAD_NORM_CLIP)`` (tools/train_utils/train_utils.py:52).
Pinned by tests/golden/optimizer.npz (captured from the imported reference classes).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def one_cycle(step: int, total_step: int, lr_max: float, moms, div_factor: float, pct_start: float):
    """(lr, beta1) at ``step``: both phases are evaluated whenever step >= start and the later one wins
    (learning_schedules_fastai.py:44-50)."""
    a1 = int(total_step * pct_start)
    low = lr_max / div_factor
    phases = [(0, a1, (low, lr_max), (moms[0], moms[1])), (a1, total_step, (lr_max, low / 1e4), (moms[1], moms[0]))]
    lr, mom = low, moms[0]
    for start, end, (l0, l1), (m0, m1) in phases:
        if step >= start:
            pct = (step - start) / (end - start)
            lr, mom = annealing_cos(l0, l1, pct), annealing_cos(m0, m1, pct)
    return float(lr), float(mom)


class AdamOneCycle:
    """fp32 reference trajectory of the full optimizer step on a list of tensors."""

    def __init__(self, params, wd=0.01, beta2=0.99, eps=1e-8, max_norm=10.0):
        self.params = list(params)
        self.wd, self.beta2, self.eps, self.max_norm = wd, beta2, eps, max_norm
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, lr, beta1):
        self.t += 1
        grads = [p.grad for p in self.params]
        total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
        coef = torch.clamp(self.max_norm / (total + 1e-6), max=1.0)
        bc1 = 1 - beta1 ** self.t
        bc2 = 1 - self.beta2 ** self.t
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            g = g * coef
            p.mul_(1 - self.wd * lr)
            m.mul_(beta1).add_(g, alpha=1 - beta1)
            v.mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-lr / bc1)
