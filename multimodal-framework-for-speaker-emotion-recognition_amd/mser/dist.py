"""Data-parallel step: one process per GPU, ONE RCCL all-reduce over the flat gradient buffer (backend "nccl" is RCCL on
ROCm; "gloo" works on CPU tensors for the protocol tests).  The reference has no distributed code at all (SURVEY.md 2 rows
19-20); this is the MI355X-native addition BASELINE.json's north_star asks for.

Sharding: contiguous split of the B dialogues; every rank runs the full time dimension of its dialogues.  MARN1_sps couples
dialogues inside a batch through speaker-slot compaction, so the N-GPU step equals the reference run on each shard
separately with the gradients combined.

Exact loss weighting: the reference divides by the local sum(mask) (loss.py:21).  Rank r's gradient is weighted by
n_r / sum_r n_r, which makes the combined gradient that of  sum_r(n_r * loss_r) / sum_r n_r.  The mask counts ride in one extra
slot of the same buffer, so there is exactly one collective per step; the division by the global count is folded into the
fused Adam kernel (``gscale_div_dev``).  A rank whose shard has no valid utterance (sum(mask) == 0) contributes a zero gradient
and a zero count (``masked_loss_bwd`` emits zeros for it), not NaN.
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
    """Owns a [total + 2] buffer: gradients pre-scaled by the local mask count, the mask count itself, and this rank's fault
    flag (mser.fault): after the SUM all-reduce the last slot counts the ranks whose step faulted, and the fused Adam launch of
    EVERY rank skips its update together (replicas never diverge)."""

    def __init__(self, total: int, device, group=None, host_staging: bool = False):
        """host_staging: reduce through a pinned host copy -- only for backends without device support (gloo, in the tests that
        run two ranks on one GPU); RCCL ("nccl") reduces the device buffer in place."""
        self.total = total
        self.buf = torch.zeros(total + 2, device=device, dtype=torch.float32)
        self.group = group
        self.host = torch.zeros(total + 2, dtype=torch.float32).pin_memory() if host_staging else None

    @property
    def grad(self) -> torch.Tensor:          # sum_r n_r * g_r after reduce()
        return self.buf[:self.total]

    @property
    def count(self) -> torch.Tensor:         # sum_r n_r after reduce()
        return self.buf[self.total:self.total + 1]

    @property
    def faults(self) -> torch.Tensor:        # number of ranks whose fault word was set, after reduce()
        return self.buf[self.total + 1:]

    def reduce(self, grad_flat: torch.Tensor, n_local: torch.Tensor) -> None:
        n_local = n_local.reshape(1).to(torch.float32)
        if grad_flat.is_cuda:
            ops.dp_pack(self.buf, grad_flat, n_local)
        else:
            # CPU tensors only occur in the gloo protocol tests (no model arithmetic is involved here)
            torch.mul(grad_flat, n_local, out=self.buf[:self.total])
            self.buf[self.total:self.total + 1].copy_(n_local)
            self.buf[self.total + 1:].zero_()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if self.host is not None:
                self.host.copy_(self.buf)
                dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=self.group)
                self.buf.copy_(self.host)
            else:
                dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)


def broadcast_replica(store, optim=None, src: int = 0, group=None) -> None:
    """Make every rank's replica identical to rank ``src``'s: ONE broadcast of the flat parameter buffer (and of Adam's moments and
    step counter when they exist).  Called by ``ModelTrainer`` before the first data-parallel step, so that identical replicas do
    not depend on every rank having seeded its initialiser alike."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ts = [store.data]
    if optim is not None and optim.m is not None:
        ts += [optim.m, optim.v, optim.step_dev]
    staged = store.data.is_cuda and dist.get_backend(group) == "gloo"      # (tests: two ranks on one GPU over gloo)
    for t in ts:
        if staged:
            h = t.cpu()
            dist.broadcast(h, src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=group)


def _small_all_reduce(t: torch.Tensor, op, group=None) -> torch.Tensor:
    """All-reduce of a few words.  RCCL ("nccl") reduces device tensors; gloo (the tests' backend, also two ranks on one GPU) reduces
    a host copy."""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=op, group=group)
    return t


def agree_on_fault(local_bits: int, device="cpu", group=None) -> int:
    """The OR of every rank's fault bits (mser.fault), identical on all ranks: ONE tiny MAX all-reduce over the three bit flags.
    Every rank must take the same control flow after a fault (raise, or fall back to per-step launches and replay) -- a rank that
    carried on alone would block at the next collective.  World 1 / no process group: the local bits."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_bits)
    dev = device if dist.get_backend(group) != "gloo" else "cpu"
    flags = torch.tensor([float(bool(local_bits & (1 << k))) for k in range(3)], device=dev)
    flags = _small_all_reduce(flags, dist.ReduceOp.MAX, group)
    return sum((1 << k) for k in range(3) if float(flags[k]) > 0)
