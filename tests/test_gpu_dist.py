"""GPU side of the data-parallel step and of the sticky fault word (run with -m gpu):

* world 1: ``FlatAllReduce.reduce`` on CUDA tensors (``mser_dp_pack``) + ``FlatAdam.step(grad=, grad_div=, gfault=)`` against the
  plain step and against ``oracle.adam_step``;
* world 2 on ONE GPU (two processes, gloo through a pinned host copy): shard -> the product's forward/backward per rank -> dp_pack ->
  all-reduce -> fused Adam with the global count, against the same two shards run in one process and combined by hand;
* world 2 over RCCL ("nccl") when the box has two GPUs (skipped otherwise);
* the fault word: a set word makes the fused Adam skip on the device and the trainer raise; out-of-range labels set it; an
  empty shard yields zero gradients, not NaN.
"""
import os
import socket

import numpy as np
import pytest
import torch

from gpu_util import load_params, maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ref_cpu
    return ref_cpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_pack_and_adam_grad_div_world1(O):
    """The device branch of FlatAllReduce.reduce (mser_dp_pack) and the grad / grad_div / gfault branch of the fused Adam."""
    from mser import fault
    from mser.dist import FlatAllReduce
    from mser.flat import FlatStore
    from mser.optim import FlatAdam
    fault.word("cuda:0").zero_()
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
    lin2 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).cuda()
    lin2.load_state_dict(lin.state_dict())
    sa, sb = FlatStore(lin), FlatStore(lin2)
    sa.attach(torch.device("cuda:0"))
    sb.attach(torch.device("cuda:0"))
    oa, ob = FlatAdam(sa, 1e-3, weight_decay=2e-5), FlatAdam(sb, 1e-3, weight_decay=2e-5)
    ar = FlatAllReduce(sa.total, "cuda:0")
    ref = {n: (sa.p(n).detach().cpu().clone(), torch.zeros(sa.shapes[n]), torch.zeros(sa.shapes[n])) for n in sa.names}
    for step in (1, 2, 3):
        g = torch.tensor(np.random.RandomState(step).standard_normal(sa.total).astype(np.float32)).cuda()
        n_local = torch.tensor(7.0 + step, device="cuda")
        sa.grad.copy_(g)
        sb.grad.copy_(g)
        ar.reduce(sa.grad, n_local)                                   # buf = g * n | n | 0
        assert torch.equal(ar.grad, g * n_local) and float(ar.count) == 7.0 + step and float(ar.faults) == 0.0
        oa.step(grad=ar.grad, grad_div=ar.count, gfault=ar.faults)    # (g * n) / n
        ob.step()
        for name in sa.names:
            off, shp = sa.offsets[name], sa.shapes[name]
            numel = int(np.prod(shp))
            O.adam_step(ref[name][0], g[off:off + numel].cpu().view(shp), ref[name][1], ref[name][2], step, 1e-3, wd=2e-5)
    assert maxabs(sa.data, sb.data) < 2e-7                            # g*n/n rounds once more than g
    for name in sa.names:
        assert maxabs(sa.p(name), ref[name][0]) < 1e-6, name
    assert oa.step_count == 3


def test_fault_word_skips_adam_and_trainer_raises(O):
    """A set fault word: the fused Adam leaves parameters, moments and the step counter untouched (on the device, no host round trip),
    dp_pack forwards the flag, and ModelTrainer.train_network raises at its epoch-end synchronisation."""
    from model_trainer import ModelTrainer
    from mser import fault
    from mser.dist import FlatAllReduce
    dev = torch.device("cuda:0")
    w = fault.word(dev)
    w.zero_()
    tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=64, quiet=True, dropout=False)
    load_params(tr.model, O.seeded_params(seed=5, d_r=64))
    x, qmask, umask, label = (t.cuda() for t in O.seeded_batch(3, 5, d_r=64, seed=6, ragged=True))
    tr.train()
    tr.scheduler.step(0)
    tr.train_step(x, qmask, umask, label)
    before = tr.model.flat_store.data.clone()
    m_before = tr.optim.m.clone()
    assert tr.optim.step_count == 1
    w.fill_(1)                                           # as a timed-out chain would leave it
    tr.train_step(x, qmask, umask, label)
    assert torch.equal(tr.model.flat_store.data, before) and torch.equal(tr.optim.m, m_before) and tr.optim.step_count == 1
    ar = FlatAllReduce(8, dev)
    ar.reduce(torch.ones(8, device=dev), torch.tensor(2.0, device=dev))
    assert float(ar.faults) == 1.0
    r = x[:, :, :64]
    batch = [r, r, r, r, torch.zeros(5, 3, 4), x[:, :, 64:], qmask, umask, label, ["v"] * 3]
    # round 3: a chain time-out is retried ONCE with one launch per time step -- the trainer switches the launch mode,
    # replays the steps whose updates the device skipped, and the epoch completes; a second time-out (or any other fault) raises
    from mser import ops
    try:
        tr.train_network(1, [batch])
        assert fault.peek(dev) == 0 and tr._fell_back
        assert not torch.equal(tr.model.flat_store.data, before)     # the replayed step did update
        w.fill_(1)
        with pytest.raises(RuntimeError, match="device fault"):
            tr.train_network(1, [batch])
        assert fault.peek(dev) == 0                          # check() cleared it: the next epoch trains again
        w.fill_(4)                                           # a bad label is never retried
        with pytest.raises(RuntimeError, match="label"):
            tr.train_network(1, [batch])
    finally:
        ops.set_option(ops.MSER_OPT_PERSISTENT, 1)
        ops.set_option(ops.MSER_OPT_DRNN_PERSISTENT, 1)
    before = tr.model.flat_store.data.clone()
    tr.train_network(1, [batch])
    assert not torch.equal(tr.model.flat_store.data, before)
    # all-reduced flag alone (another rank faulted): this rank skips too
    snap = tr.model.flat_store.data.clone()
    tr.optim.step(gfault=torch.ones(1, device=dev))
    assert torch.equal(tr.model.flat_store.data, snap)


def test_bad_label_sets_fault_and_ignore_index(O):
    """loss.py:19-24: torch's lossers raise on a target outside [0, C) and skip ignore_index = -100.  Here: the row is skipped and the
    fault word says so / the row is skipped silently but still counts in sum(mask)."""
    from loss import MaskedLoss
    from mser import fault
    dev = torch.device("cuda:0")
    fault.word(dev).zero_()
    rs = np.random.RandomState(3)
    pred = torch.log_softmax(torch.tensor(rs.standard_normal((12, 6)).astype(np.float32)), 1)
    target = torch.tensor(rs.randint(0, 6, 12))
    mask = torch.ones(3, 4)
    mask[2, 3] = 0
    for is_ce, losser in ((False, torch.nn.NLLLoss), (True, torch.nn.CrossEntropyLoss)):
        t2 = target.clone()
        t2[1] = -100
        p = pred.clone().cuda().requires_grad_(True)
        out = MaskedLoss(losser)(p, t2.cuda(), mask.cuda())
        out.backward()
        pr = pred.clone().requires_grad_(True)
        ref = losser(reduction="sum")(pr * mask.view(-1, 1), t2) / mask.sum()
        ref.backward()
        assert abs(float(out) - float(ref)) < 1e-6 and maxabs(p.grad, pr.grad) < 1e-7
        assert fault.peek(dev) == 0
    t3 = target.clone()
    t3[5] = 6
    p = pred.clone().cuda().requires_grad_(True)
    out = MaskedLoss(torch.nn.NLLLoss)(p, t3.cuda(), mask.cuda())
    out.backward()
    assert bool(torch.isfinite(out)) and float(p.grad[5].abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="label"):
        fault.check(dev, "test")
    assert fault.peek(dev) == 0


def test_empty_shard_contributes_zero_gradient(O):
    """A rank whose shard has no valid utterance: the loss is 0/0 like the reference's, its gradient must be zeros (n_r * g_r
    rides the all-reduce), not NaN."""
    from loss import MaskedLoss
    pred = torch.log_softmax(torch.randn(8, 6), 1).cuda().requires_grad_(True)
    target = torch.randint(0, 6, (8,)).cuda()
    out = MaskedLoss(torch.nn.NLLLoss)(pred, target, torch.zeros(2, 4).cuda())
    out.backward()
    assert not bool(torch.isfinite(out)) and float(pred.grad.abs().sum()) == 0.0


# ---- two ranks -------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, backend, out_path, d_r, Bg, Ln):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", rank % ndev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_cpu as O
    from model_trainer import ModelTrainer
    from mser.dist import shard_batch
    if ndev < world:
        # ranks SHARE a GPU here: two persistent launches of 160-208 workgroups each cannot both be resident on 256 CUs, and if the
        # dispatcher interleaves them each can hold CUs the other waits for until the bounded waits give up.  The data-parallel
        # device path under test (dp_pack, flag, fused Adam) does not depend on the launch mode: one launch per step here.
        from mser import ops
        ops.set_option(ops.MSER_OPT_PERSISTENT, 0)
    torch.manual_seed(100 + rank)                        # DIFFERENT initial draws per rank: the trainer must broadcast rank 0's
    tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=d_r, quiet=True, dropout=False)
    if rank == 0:
        load_params(tr.model, O.seeded_params(seed=31, d_r=d_r))
    x, qmask, umask, label = O.seeded_batch(Bg, Ln, d_r=d_r, seed=32, ragged=True)
    xs, qs, us, ls = (t.to(dev) for t in shard_batch(x, qmask, umask, label, rank, world))
    tr.train()
    tr.scheduler.step(0)
    for _ in range(2):
        tr.train_step(xs, qs, us, ls)
    torch.cuda.synchronize()
    flat = tr.model.flat_store.data.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    # ---- a fault on ONE rank must raise on EVERY rank (train_network and eval_network decide collectively); rank 1 plants
    # a bad-label bit in its fault word, rank 0's word stays clean
    from mser import fault
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    r = xs[:, :, :d_r].contiguous()
    batch = [r.cpu(), r.cpu(), r.cpu(), r.cpu(), torch.zeros(Ln, xs.shape[1], 4), xs[:, :, d_r:].cpu(), qs.cpu(), us.cpu(), ls.cpu(),
             ["v"] * xs.shape[1]]
    raised = []
    for fn in (lambda: tr.train_network(1, [batch]), lambda: tr.eval_network([batch])):
        if rank == 1:
            fault.word(dev).fill_(4)
        try:
            fn()
            raised.append("")
        except RuntimeError as e:
            raised.append(str(e))
    clean = True
    try:
        tr.train_network(1, [batch])                      # and the next epoch runs on both ranks (no rank was left behind)
    except RuntimeError:
        clean = False
    torch.save(dict(raised=raised, clean=clean), out_path + f".fault{rank}")
    if rank == 0:
        torch.save(dict(flat=flat, same=bool(all(torch.equal(g, flat) for g in gathered)), sd=sd), out_path)
    dist.barrier()
    dist.destroy_process_group()


def _dp_reference(O, d_r, Bg, Ln, world):
    """The same two steps in ONE process: the product's forward/backward on each shard separately, gradients combined by hand with
    the mask counts (sum_r n_r g_r / sum_r n_r), ONE fused Adam step on the combination."""
    from model_trainer import ModelTrainer
    from mser.dist import shard_batch
    dev = torch.device("cuda:0")
    tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=d_r, quiet=True, dropout=False)
    load_params(tr.model, O.seeded_params(seed=31, d_r=d_r))
    x, qmask, umask, label = O.seeded_batch(Bg, Ln, d_r=d_r, seed=32, ragged=True)
    shards = [[t.to(dev) for t in shard_batch(x, qmask, umask, label, r, world)] for r in range(world)]
    tr.train()
    tr.scheduler.step(0)
    for _ in range(2):
        acc, cnt = None, 0.0
        for xs, qs, us, ls in shards:
            tr.forward_backward(xs, qs, us, ls)
            n = float(us.sum())
            g = tr.model.flat_store.grad.clone() * n
            acc = g if acc is None else acc + g
            cnt += n
        tr.optim.step(grad=acc, grad_div=torch.tensor([cnt], device=dev))
    torch.cuda.synchronize()
    return tr.model.flat_store.data.detach().cpu()


def _run_two_ranks(O, tmp_path, backend):
    import torch.multiprocessing as mp
    d_r, Bg, Ln, world = 64, 6, 7, 2
    out = str(tmp_path / f"dp_{backend}.pt")
    mp.spawn(_dp_worker, args=(world, _free_port(), backend, out, d_r, Bg, Ln), nprocs=world, join=True)
    res = torch.load(out, weights_only=True)
    assert res["same"], "replicas diverged"
    # every rank raised for rank 1's fault, in training and in evaluation, and both went on afterwards
    for rk in range(world):
        fr = torch.load(out + f".fault{rk}", weights_only=True)
        assert len(fr["raised"]) == 2 and all("device fault" in m and "label" in m for m in fr["raised"]), (rk, fr["raised"])
        assert ("another rank" in fr["raised"][0]) == (rk == 0)
        assert fr["clean"], rk
    # SURVEY 8(e) as the REFERENCE computes it (tests/golden/make_golden.py::dp_case: the reference on each shard separately, gradients
    # combined with the mask counts, one torch.optim.Adam step; twice): the replicas' parameters against that fixture
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dp_two_shards.npz"), allow_pickle=False)
    assert (int(g["d_r"]), int(g["B"]), int(g["L"]), int(g["world"]), int(g["seed_params"]), int(g["seed_batch"])) == (d_r, Bg, Ln, world, 31, 32)
    worst = 0.0
    for k in res["sd"]:
        got = res["sd"][k].reshape(-1).numpy()[g["idx/" + k]]
        worst = max(worst, float(np.abs(got - g["p/" + k]).max()))
    assert worst < 5e-6, worst
    from mser import ops
    shared = torch.cuda.device_count() < world              # the workers then ran one launch per step (see _dp_worker): so does the reference
    if shared:
        ops.set_option(ops.MSER_OPT_PERSISTENT, 0)
    try:
        ref = _dp_reference(O, d_r, Bg, Ln, world)
    finally:
        ops.set_option(ops.MSER_OPT_PERSISTENT, 1)
    # (the weight-gradient GEMMs accumulate split-K partials with float atomics: equal to rounding, not bitwise)
    assert maxabs(res["flat"], ref) < 2e-6, maxabs(res["flat"], ref)


def test_data_parallel_two_ranks_one_gpu_gloo(O, tmp_path):
    """Two processes share cuda:0; the collective itself runs over gloo through a pinned host copy (FlatAllReduce host_staging), every
    device-side piece of the step -- per-shard slot tables and chains, mser_dp_pack, the fused Adam with grad_div / gfault, the initial
    replica broadcast -- is the product's."""
    _run_two_ranks(O, tmp_path, "gloo")


def test_data_parallel_two_ranks_rccl(O, tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier); the one-GPU gloo test above covers the device path")
    _run_two_ranks(O, tmp_path, "nccl")


def test_bench_two_rank_flow_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the one-GPU box: the script starts its two ranks itself (child torch.distributed.run), each
    rank runs its shard (graph-captured forward + backward, eager pack -> all-reduce -> Adam), the timings are MAX-reduced, rank 0 prints
    the one JSON line.  MSER_BENCH_REHEARSE=gloo puts both ranks on device 0 and the collective on gloo through pinned host staging; the
    number is meaningless, the launch / barrier / reduce / report logic is the one the N-GPU run uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSER_BENCH_REHEARSE="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-variants", "--no-roofline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2
    assert line["value"] > 0 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
