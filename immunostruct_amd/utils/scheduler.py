"""``LinearWarmupCosineAnnealingLR`` (reference ``utils/scheduler.py:9-138``, used by ``train_Cancer_wFT.py:146-149``).

Linear warm-up from ``warmup_start_lr`` to the base learning rate over ``warmup_epochs`` steps, then half a cosine down
to ``eta_min`` at ``max_epochs``.  Written in closed form (the value depends only on the step index); the reference's
chainable recurrence produces the same sequence, which ``tests/golden/metrics.npz`` pins for three configurations.
"""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import LRScheduler

__all__ = ["LinearWarmupCosineAnnealingLR"]


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def _value(self, base_lr, t):
        if t < self.warmup_epochs:
            return self.warmup_start_lr + t * (base_lr - self.warmup_start_lr) / max(1, self.warmup_epochs - 1)
        phase = math.pi * (t - self.warmup_epochs) / (self.max_epochs - self.warmup_epochs)
        return self.eta_min + 0.5 * (base_lr - self.eta_min) * (1.0 + math.cos(phase))

    def get_lr(self):
        return [self._value(b, self.last_epoch) for b in self.base_lrs]

    _get_closed_form_lr = get_lr
