"""Flat fp32 parameter / gradient storage.

Every ``nn.Parameter`` of a model keeps its reference name and shape (so ``state_dict`` files interchange with the
reference, model_trainer.py:170-187) but its storage is a view into ONE flat buffer; gradients live in a second flat
buffer of the same layout.  That gives the data-parallel step a single RCCL all-reduce (8.8 MB at reference width) and
the optimiser a single fused Adam launch, instead of ~100 per-tensor calls.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # floats (256 B): keeps every tensor 16-byte aligned for the float4 loads of the recurrent kernels


class FlatStore:
    def __init__(self, module: nn.Module, dead: Iterable[str] = ()):
        self.module = module
        self.names: List[str] = []
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        off = 0
        for name, p in module.named_parameters():
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(p.shape)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.dead = set(dead)
        self.data: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.live: Optional[torch.Tensor] = None
        self._views: Dict[str, torch.Tensor] = {}
        self._gviews: Dict[str, torch.Tensor] = {}
        self._params: Dict[str, nn.Parameter] = dict(module.named_parameters())

    # ------------------------------------------------------------------------------------------
    def is_attached(self, device: torch.device) -> bool:
        if self.data is None or self.data.device != device:
            return False
        base = self.data.data_ptr()
        for name in (self.names[0], self.names[-1]):
            if self._params[name].data_ptr() != base + 4 * self.offsets[name]:
                return False
        return True

    def attach(self, device: torch.device) -> None:
        """(Re)build the flat buffers on ``device`` from the module's current parameter values and re-point every parameter."""
        data = torch.zeros(self.total, device=device, dtype=torch.float32)
        live = torch.zeros(self.total, device=device, dtype=torch.uint8)
        self._views.clear()
        self._gviews.clear()
        grad = torch.zeros(self.total, device=device, dtype=torch.float32)
        with torch.no_grad():
            for name in self.names:
                p = self._params[name]
                if p.dtype != torch.float32:
                    raise RuntimeError(f"parameter {name} is {p.dtype}; the MI355X path computes in float32")
                off, n = self.offsets[name], p.numel()
                view = data[off:off + n].view(self.shapes[name])
                view.copy_(p.data.to(device))
                p.data = view
                self._views[name] = view
                self._gviews[name] = grad[off:off + n].view(self.shapes[name])
                if name not in self.dead:
                    live[off:off + n] = 1
                p.grad = None
        self.data, self.grad, self.live = data, grad, live

    def p(self, name: str) -> torch.Tensor:
        return self._views[name]

    def g(self, name: str) -> Optional[torch.Tensor]:
        return self._gviews[name]

    def publish_grads(self) -> None:
        """Expose the flat gradient through ``param.grad`` views (dead parameters keep ``None`` like in the reference)."""
        for name in self.names:
            if name in self.dead:
                continue
            p = self._params[name]
            if p.grad is None or p.grad.data_ptr() != self._gviews[name].data_ptr():
                p.grad = self._gviews[name]

    def grads_were_reset(self) -> bool:
        """True when the caller ran ``zero_grad(set_to_none=True)`` (torch's default) since the last backward."""
        for name in self.names:
            if name not in self.dead:
                return self._params[name].grad is None
        return False

    def zero_grad(self) -> None:
        if self.grad is not None:
            self.grad.zero_()
