"""Data-parallel step: one process per GPU, ONE RCCL all-reduce over the flat gradient buffer (backend "nccl" is RCCL on
ROCm; "gloo" works on CPU tensors for tests).  The reference has no distributed code at all; this is
the MI355X-native addition BASELINE.json's north_star asks for.

Sharding: contiguous split of the B dialogues; every rank runs the full time dimension of its dialogues.  MARN1_sps couples
dialogues inside a batch through speaker-slot compaction, so the N-GPU step equals the reference run on each shard
separately with the gradients combined -- that is what the gloo test checks.

Exact loss weighting: the reference divides by the local sum(mask) (loss.py:21).  ``combine`` weights rank r's gradient by
n_r / sum_r n_r, which makes the combined gradient equal to d/dtheta of  sum_r(n_r * loss_r) / sum_r n_r.  The mask counts ride in one
extra slot of the same all-reduce buffer, so there is still exactly one collective per step.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def shard_batch(x, qmask, umask, label, rank: int, world: int):
    """Contiguous split along the dialogue axis (B): x [L,B,F], qmask [L,B,2], umask [B,L], label [B,L]."""
    B = x.shape[1]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    s = slice(rank * (B // world), (rank + 1) * (B // world))
    return x[:, s].contiguous(), qmask[:, s].contiguous(), umask[s].contiguous(), label[s].contiguous()


class FlatAllReduce:
    """Owns a [total + 1] buffer: gradients (pre-scaled by the local mask count) + the mask count itself."""

    def __init__(self, total: int, device, group=None):
        self.buf = torch.zeros(total + 1, device=device, dtype=torch.float32)
        self.group = group

    def combine(self, grad_flat: torch.Tensor, n_local: torch.Tensor) -> torch.Tensor:
        """grad_flat <- sum_r n_r * grad_r / sum_r n_r (in place).  ``n_local`` is a 0-d / 1-element device tensor."""
        n = self.buf.numel() - 1
        torch.mul(grad_flat, n_local.reshape(1), out=self.buf[:n])
        self.buf[n:].copy_(n_local.reshape(1))
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        torch.div(self.buf[:n], self.buf[n:], out=grad_flat)
        return grad_flat
