"""Fused flat-buffer Adam + closed-form StepLR: the optimiser half of the reference's trainer
(model_trainer.py:82-83 ``torch.optim.Adam(lr, weight_decay=2e-5)`` + ``StepLR(step_size=test_step, gamma=lr_decay)``,
stepped with an explicit epoch at :92)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .flat import FlatStore


class FlatAdam:
    """Two HIP launches over the flat parameter buffer (bias-correction scalars, then the update).  The step counter and the
    learning rate live on the device so the launches can sit inside a captured hipGraph.  Parameters that never receive a
    gradient in the reference (``.grad is None`` -> skipped by torch, no weight decay either) are masked out through
    ``store.live``."""

    def __init__(self, store: FlatStore, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.store = store
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.step_dev: Optional[torch.Tensor] = None
        self.hp_dev: Optional[torch.Tensor] = None
        self.sched_dev: Optional[torch.Tensor] = None
        self._hp_host = None

    @property
    def step_count(self) -> int:
        return 0 if self.step_dev is None else int(self.step_dev.item())

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.store.zero_grad()

    def _ensure_state(self) -> None:
        d = self.store.data
        if self.m is None or self.m.device != d.device or self.m.numel() != d.numel():
            self.m = torch.zeros_like(d)
            self.v = torch.zeros_like(d)
            self.step_dev = torch.zeros(1, device=d.device, dtype=torch.int32)
            self.hp_dev = torch.zeros(3, device=d.device)
            self.sched_dev = torch.zeros(2, device=d.device)
            self._hp_host = None

    def sync_hyperparams(self) -> None:
        """Push {lr, beta1, beta2} to the device if they changed (outside any graph capture)."""
        self._ensure_state()
        g = self.param_groups[0]
        hp = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
        if hp != self._hp_host:
            self.hp_dev.copy_(torch.tensor(hp, dtype=torch.float32))
            self._hp_host = hp

    def step(self, sync_hp: bool = True, grad: Optional[torch.Tensor] = None, grad_div: Optional[torch.Tensor] = None,
             gfault: Optional[torch.Tensor] = None) -> None:
        """``grad`` / ``grad_div`` (device tensors) override the flat gradient: the data-parallel path passes the all-reduced
        buffer and the global mask count so that no copy-back or separate scaling pass is needed.  The launch skips the update on the
        device while the device's fault word (mser.fault) or ``gfault`` (the all-reduced fault flags) is non-zero."""
        if self.store.data is None:
            raise RuntimeError("FlatAdam.step() before the first forward/backward")
        if sync_hp:
            self.sync_hyperparams()
        g = self.param_groups[0]
        ops.adam_flat_dev(self.store.data, self.store.grad if grad is None else grad, self.m, self.v, self.store.live,
                          self.step_dev, self.hp_dev, self.sched_dev, g["eps"], g["weight_decay"], grad_div, 1.0, gfault)

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
