"""A tiny bridge between torch.autograd and the hand-written forward/backward pairs."""
from __future__ import annotations

import torch


class ModuleFn(torch.autograd.Function):
    """``impl.fwd(*tensors) -> (tuple_of_outputs, saved)``; ``impl.bwd(saved, tensors, *douts) -> grads aligned with tensors``."""

    @staticmethod
    def forward(ctx, impl, *tensors):
        outs, saved = impl.fwd(*[t.detach() if isinstance(t, torch.Tensor) else t for t in tensors])
        ctx.impl, ctx.saved, ctx.tensors = impl, saved, tensors
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        grads = ctx.impl.bwd(ctx.saved, ctx.tensors, *douts)
        return (None, *grads)


def require_gpu(*tensors) -> None:
    for t in tensors:
        if isinstance(t, torch.Tensor) and not t.is_cuda:
            raise RuntimeError("this module runs on the MI355X HIP path only: move the module and its inputs to a GPU "
                               "(there is deliberately no CPU fallback; the CPU restatement lives in oracle/ for tests)")
