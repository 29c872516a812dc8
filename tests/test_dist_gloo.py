"""CPU, world_size = 2, gloo: the data-parallel protocol (contiguous dialogue sharding + ONE flat all-reduce carrying the
mask-count weights) reproduces the gradient of the globally mask-weighted loss.  Model gradients come from the CPU oracle
(test infrastructure) -- the collective / packing logic under test is the product's (mser.dist)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grad(P0, x, qmask, umask, label, d_r):
    P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    lp, _, _ = O.marn1_sps_forward(P, x, qmask, umask, d_r=d_r)
    loss = O.masked_nll(lp, label.view(-1), umask)
    loss.backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in P.values()])
    return flat, float(loss), float(umask.sum())


def _worker(rank, world, port, d_r, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mser.dist import FlatAllReduce, shard_batch
    P0 = O.seeded_params(seed=21, d_r=d_r)
    x, qmask, umask, label = O.seeded_batch(4, 6, d_r=d_r, seed=22, ragged=True)
    xs, qs, us, ls = shard_batch(x, qmask, umask, label, rank, world)
    g, loss, n = _shard_grad(P0, xs, qs, us, ls, d_r)
    ar = FlatAllReduce(g.numel(), "cpu")
    ar.reduce(g, torch.tensor(n))
    combined = ar.grad / ar.count
    if rank == 0:
        torch.save(dict(combined=combined, count=float(ar.count)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_matches_global_weighted_loss(tmp_path):
    d_r, world = 64, 2
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), d_r, out), nprocs=world, join=True)
    res = torch.load(out, weights_only=True)
    # single-process ground truth: sum_r n_r * loss_r / sum_r n_r, each shard run on its own (per-shard slot compaction!)
    P0 = O.seeded_params(seed=21, d_r=d_r)
    x, qmask, umask, label = O.seeded_batch(4, 6, d_r=d_r, seed=22, ragged=True)
    from mser.dist import shard_batch
    gs, ns = [], []
    for r in range(world):
        g, _, n = _shard_grad(P0, *shard_batch(x, qmask, umask, label, r, world), d_r)
        gs.append(g)
        ns.append(n)
    ref = sum(n * g for n, g in zip(ns, gs)) / sum(ns)
    assert res["count"] == sum(ns)
    assert float((res["combined"] - ref).abs().max()) < 1e-6 * max(1.0, float(ref.abs().max()))


def test_shard_batch_contiguous_and_validates():
    from mser.dist import shard_batch
    x, qmask, umask, label = O.seeded_batch(6, 5, d_r=8, seed=1)
    parts = [shard_batch(x, qmask, umask, label, r, 3) for r in range(3)]
    assert torch.equal(torch.cat([p[0] for p in parts], 1), x)
    assert torch.equal(torch.cat([p[2] for p in parts], 0), umask)
    with pytest.raises(ValueError):
        shard_batch(x, qmask, umask, label, 0, 4)


def _fault_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mser.dist import agree_on_fault
    res = [agree_on_fault(0),                                # nobody faulted
           agree_on_fault(1 if rank == 1 else 0),            # only rank 1 timed out: EVERY rank must see it
           agree_on_fault(4 if rank == 0 else 2),            # different bits on different ranks: the OR
           agree_on_fault(0)]
    torch.save(res, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_fault_bits_are_agreed_by_all_ranks(tmp_path):
    """only the rank that faulted used to raise at epoch end; the others carried on into the next collective.  The
    decision is collective now (mser.dist.agree_on_fault: one MAX all-reduce over the bit flags): every rank gets the same bits."""
    out = str(tmp_path / "bits")
    mp.spawn(_fault_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=True), torch.load(out + ".1", weights_only=True)
    assert r0 == r1 == [0, 1, 6, 0]
    from mser.dist import agree_on_fault
    assert agree_on_fault(5) == 5                            # no process group: the local bits
