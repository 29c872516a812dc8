"""Data-parallel step: one process per GPU, ONE RCCL all-reduce over the flat gradient buffer (backend "nccl" is RCCL on
ROCm; "gloo" works on CPU tensors for the protocol tests).  The reference has no distributed code at all (SURVEY.md 2 rows
19-20); this is the MI355X-native addition BASELINE.json's north_star asks for.

Sharding: contiguous split of the B dialogues; every rank runs the full time dimension of its dialogues.  MARN1_sps couples
dialogues inside a batch through speaker-slot compaction, so the N-GPU step equals the reference run on each shard
separately with the gradients combined.

Exact loss weighting: the reference divides by the local sum(mask) (loss.py:21).  Rank r's gradient is weighted by
n_r / sum_r n_r, which makes the combined gradient that of  sum_r(n_r * loss_r) / sum_r n_r.  The mask counts ride in one extra
slot of the same buffer, so there is exactly one collective per step; the division by the global count is folded into the
fused Adam kernel (``gscale_div_dev``).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


def shard_batch(x, qmask, umask, label, rank: int, world: int):
    """Contiguous split along the dialogue axis (B): x [L,B,F], qmask [L,B,2], umask [B,L], label [B,L]."""
    B = x.shape[1]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    s = slice(rank * (B // world), (rank + 1) * (B // world))
    return x[:, s].contiguous(), qmask[:, s].contiguous(), umask[s].contiguous(), label[s].contiguous()


class FlatAllReduce:
    """Owns a [total + 1] buffer: gradients pre-scaled by the local mask count, plus the mask count itself."""

    def __init__(self, total: int, device, group=None):
        self.total = total
        self.buf = torch.zeros(total + 1, device=device, dtype=torch.float32)
        self.group = group

    @property
    def grad(self) -> torch.Tensor:          # sum_r n_r * g_r after reduce()
        return self.buf[:self.total]

    @property
    def count(self) -> torch.Tensor:         # sum_r n_r after reduce()
        return self.buf[self.total:]

    def reduce(self, grad_flat: torch.Tensor, n_local: torch.Tensor) -> None:
        n_local = n_local.reshape(1).to(torch.float32)
        if grad_flat.is_cuda:
            ops.dp_pack(self.buf, grad_flat, n_local)
        else:
            # CPU tensors only occur in the gloo protocol tests (no model arithmetic is involved here)
            torch.mul(grad_flat, n_local, out=self.buf[:self.total])
            self.buf[self.total:].copy_(n_local)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
