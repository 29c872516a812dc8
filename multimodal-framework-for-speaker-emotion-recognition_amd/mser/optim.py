"""Fused flat-buffer Adam + closed-form StepLR: the optimiser half of the reference's trainer
(model_trainer.py:82-83 ``torch.optim.Adam(lr, weight_decay=2e-5)`` + ``StepLR(step_size=test_step, gamma=lr_decay)``,
stepped with an explicit epoch at :92)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .flat import FlatStore


class FlatAdam:
    """One HIP launch over the flat parameter buffer.  Parameters that never receive a gradient in the reference
    (``.grad is None`` -> skipped by torch, no weight decay either) are masked out through ``store.live``."""

    def __init__(self, store: FlatStore, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.store = store
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.step_count = 0
        self.grad_scale = 1.0        # set by the data-parallel wrapper (1 / world_size after a SUM all-reduce)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.store.zero_grad()

    def _ensure_state(self) -> None:
        d = self.store.data
        if self.m is None or self.m.device != d.device or self.m.numel() != d.numel():
            self.m = torch.zeros_like(d)
            self.v = torch.zeros_like(d)

    def step(self) -> None:
        if self.store.data is None:
            raise RuntimeError("FlatAdam.step() before the first forward/backward")
        self._ensure_state()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_flat(self.store.data, self.store.grad, self.m, self.v, self.store.live, self.step_count, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], g["weight_decay"], self.grad_scale)

    def state_dict(self):
        return dict(step=self.step_count, m=self.m, v=self.v, param_groups=self.param_groups)


class StepLR:
    """Closed form used by the reference: ``scheduler.step(epoch - 1)`` -> lr = lr0 * gamma ** ((epoch-1) // step_size)."""

    def __init__(self, optimizer: FlatAdam, step_size: int, gamma: float):
        self.optimizer, self.step_size, self.gamma = optimizer, step_size, gamma
        self.last_epoch = 0

    def step(self, epoch: Optional[int] = None) -> None:
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self.gamma ** (self.last_epoch // self.step_size)
