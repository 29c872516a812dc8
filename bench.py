#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native speaker-aware LSTHM path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full optimisation step of MARN1_sps on one synthetic batch per GPU (BASELINE.json config 2 shape:
B=32 dialogues x L=128 utterances, d_text=768, d_audio=100, reference width H=128, fp32): zero_grad, forward, MaskedLoss,
backward, (N>1: ONE RCCL all-reduce of the flat gradient buffer), fused Adam.  Inputs are resident in HBM before the timed
region.  Weak scaling: every rank owns its own 32 dialogues, global batch = 32*N.  value = N*B*L / step time.

The single JSON line also carries
  roofline     -- the dominant kernel (the LSTHM backward chain; the forward chain rides along) measured live with HIP events
                  around each of its launches (eager pass after the timed region), against the HBM roofline;
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, eval-mode fwd+bwd, same shapes) timed on this
                  host's cores (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 matrix (v_mfma_f32_32x32x2_f32), dense
B, L, D_R, D_A, H, NCLS = 32, 128, 768, 100, 128, 6


def synth_batch(seed, device, ragged=False, nb=None):
    """x ~ N(0,1), speaker ~ Bernoulli(0.5) one-hot, labels ~ U{0..5}; full-length dialogues for the headline,
    lengths ~ U{L/2..L} (tail zeroed in x, qmask, umask) for the ragged variant that exercises _reverse_seq / padding."""
    rs = np.random.RandomState(seed)
    B = nb or globals()["B"]
    x = rs.standard_normal((L, B, D_R + D_A)).astype(np.float32)
    spk = rs.randint(0, 2, (L, B))
    qmask = np.eye(2, dtype=np.float32)[spk]
    umask = np.ones((B, L), dtype=np.float32)
    if ragged:
        lens = rs.randint(L // 2, L + 1, B)
        lens[0] = L
        for b in range(B):
            x[lens[b]:, b] = 0.0
            qmask[lens[b]:, b] = 0.0
            umask[b, lens[b]:] = 0.0
    label = rs.randint(0, NCLS, (B, L)).astype(np.int64)
    return [torch.tensor(t).to(device) for t in (x, qmask, umask, label)]


def init_attention_weights(model, seed=0):
    """The reference initialises every attention matrix to ones (uniform softmaxes); draw them N(0, s^2) instead so the
    softmaxes are non-trivial.  Cost is identical either way."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "crossatt" in n:
                s = 0.5 if p.shape[0] == 1 else 0.6 / np.sqrt(p.shape[0])
                p.copy_(torch.tensor((rs.standard_normal(tuple(p.shape)) * s).astype(np.float32)))


def lsthm_step_bytes(backward):
    """Algorithmic bytes of ONE LSTHM time step of ONE direction (both streams) -- the streaming model of DESIGN.md 5.
    forward : read U,V [4H,H] x2 + biases, pre-activations [B,4H] x2, h_l|h_a|z [B,3H], c [B,H] x2;
              write gates [B,4H] x2, c x2, h x2 (state) + h,z output rows.
    backward: read U,V x2, saved gates x2, c_t and c_{t-1} x2, z row, dout row [B,4H], dA [4][B,H], dc carry x2;
              write dgates x2 (and read them back in the matvec phase), dA, dHQ, dc carry."""
    f = 4
    if not backward:
        return f * (2 * 2 * 4 * H * H + 2 * 2 * 4 * H + 2 * B * 4 * H + B * 3 * H + 2 * B * H
                    + 2 * B * 4 * H + 2 * B * H + 2 * B * H + 3 * B * H)
    return f * (2 * 2 * 4 * H * H + 2 * B * 4 * H + 4 * B * H + B * H + B * 4 * H + 4 * B * H + 2 * B * H
                + 2 * 2 * B * 4 * H + 4 * B * H + B * H + 2 * B * H)


def lsthm_step_bytes_survey(backward):
    """SURVEY.md 8(d)'s streaming model of one LSTHM + speaker step of one direction (the contract's per-unit figure):
    forward  s * [P_step + B*(2D+2) + B*4H],  P_step = 2*(4H*(D+2H+Hs) + 4*4H) + 2*(4Hs*2Hs + 8Hs) + 2H parameters (every weight of the
    step streamed once, W and S and the speaker LSTMs included); backward = the same weights once more + the saved-gate reads
    B*(2*4H + 2*4Hs + 6H)*s.  H = Hs = 128, D = 100, B = 32, fp32: 3,148,032 / 3,508,480 bytes."""
    D, Hs = 100, H
    p_step = 2 * (4 * H * (D + 2 * H + Hs) + 4 * 4 * H) + 2 * (4 * Hs * 2 * Hs + 8 * Hs) + 2 * H
    by = 4 * (p_step + B * (2 * D + 2) + B * 4 * H)
    if backward:
        by += 4 * B * (2 * 4 * H + 2 * 4 * Hs + 6 * H)
    return by


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE
    rocprofv3 passes over this same command, FETCH_SIZE doubled per the gfx950 correction).  PMC collection serialises kernels,
    so it cannot run inside the timed region; None when no summary names the kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            ks = json.load(open(f))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, v in ks.items():
            if kernel + "<" in name or kernel + "(" in name:
                return v["traffic_bytes_per_launch"]
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(steps=5):
    """The CPU oracle (oracle/ref_cpu.py, kind "port": the restatement pinned against the reference's own outputs) timed on this
    host's cores on the full B x L batch: 1 warm-up + `steps` timed passes per leg, median.  Legs:
      eval_fwd_bwd    -- every Dropout the identity, forward + MaskedLoss + backward   (the parity configuration; this is `value`,
                         the FASTEST of the three, i.e. the most conservative GPU/CPU ratio)
      train_fwd_bwd   -- all 13 dropout sites live, masks drawn inside the timed region with torch's CPU generator, as the
                         reference's nn.Dropout modules do
      trainer_step    -- batch ingest (textf = (r1+r2+r3+r4)/4, cat), train-mode forward + loss + backward, Adam(wd=2e-5) over the
                         100 live tensors: what ModelTrainer.train_network does per batch (model_trainer.py:96-120)"""
    from oracle import ref_cpu as O
    # the GPU box hands a 1-GPU job a 16-core CPU share (cgroup), while os.cpu_count() reports the whole host
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))
    torch.set_num_threads(ncores)
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(seed=0, d_r=D_R).items()}
    x, qmask, umask, label = O.seeded_batch(B, L, d_r=D_R, seed=1)
    r = [x[:, :, :D_R] + 0.01 * i for i in range(4)]          # four RoBERTa layers of the reference's batch tuple
    acouf = x[:, :, D_R:].contiguous()
    shapes = {k: tuple(v.shape) for k, v in O.seeded_drops(1, 1).items()}          # site keys only
    pk = O.DEFAULT_DROPOUT

    def draw_drops():
        full = {}
        for k in shapes:
            if k.startswith("enc"):
                shp = (B, 8, L, L) if k.endswith("attn") else (B, L, 100)
                p_ = pk["enc"]
            elif k.startswith("xattn"):
                shp, p_ = (B, L, L), pk["xattn"]
            elif k == "fc":
                shp, p_ = (L, B, 100), pk["fc"]
            elif k == "out":
                shp, p_ = (L, B, 32), pk["out"]
            elif k.startswith("rec"):
                shp, p_ = (L, B, 4 * H), pk["rec"]
            elif k.endswith("attn"):
                shp, p_ = (L, B, H, H), pk["cell_attn"]
            else:
                shp, p_ = (L, 2, B, H), pk["cell"]
            full[k] = torch.nn.functional.dropout(torch.ones(shp), p_, True)
        return full

    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in P.items()}
    step_no = [0]

    def leg(mode):
        for p_ in P.values():
            p_.grad = None
        t0 = time.perf_counter()
        xx = x
        if mode == "trainer":
            xx = torch.cat([(r[0] + r[1] + r[2] + r[3]) / 4, acouf], dim=-1)
        drops = draw_drops() if mode != "eval" else None
        lp, _, _ = O.marn1_sps_forward(P, xx, qmask, umask, d_r=D_R, drops=drops)
        O.masked_nll(lp, label.view(-1), umask).backward()
        if mode == "trainer":
            step_no[0] += 1
            with torch.no_grad():
                for k, p_ in P.items():
                    if p_.grad is not None:
                        O.adam_step(p_, p_.grad, state[k][0], state[k][1], step_no[0], 1e-3)
        return time.perf_counter() - t0

    legs = {}
    for name, mode in (("eval_fwd_bwd", "eval"), ("train_fwd_bwd", "train"), ("trainer_step", "trainer")):
        ts = [leg(mode) for _ in range(steps + 1)][1:]
        t = float(np.median(ts))
        legs[name] = {"s_per_step": round(t, 4), "utterances_per_s": round(B * L / t, 1)}
        log(f"cpu baseline {name}: {t:.3f} s/step")
    t = legs["eval_fwd_bwd"]["s_per_step"]
    return dict(value=legs["eval_fwd_bwd"]["utterances_per_s"], unit="utterances/s", cores=torch.get_num_threads(), kind="port",
                cpu_model=cpu_model(), legs=legs,
                sample=f"{steps} timed steps (+1 warm-up) per leg of the full B={B},L={L},d_t={D_R} batch, median; value = eval-mode "
                       f"fwd+bwd ({t:.3f} s/step), the fastest leg; torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads")


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`
    as a CHILD process (never exec: the parent has not touched the GPU, but a child is the safe form everywhere), stream its
    output through, return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--graph", action="store_true", help="force hipGraph replay (default: whichever of eager / graph is faster)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary workloads (ragged lengths, ones-initialised attention)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host-side tensor work (the staging copies of the host-batch variant, the CPU baseline) on the box's CPU share: the 1-GPU box hands
    # out 16 cores while torch defaults to one thread per host core (128); oversubscribed, a 50 MB staging copy took 15 ms instead of 0.2
    # (scratch/prefetch_probe.py)
    try:
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    except AttributeError:
        torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # Invoked directly as `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL), BEFORE this
            # process makes any GPU call, relay rank 0's JSON line and leave with the children's exit code.
            raise SystemExit(self_launch(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # MSER_BENCH_REHEARSE=gloo: the N-rank flow of this script on ONE GPU (every rank on device 0, gloo through pinned host staging) -- a
    # rehearsal of the launch / barrier / reduce / report logic where no multi-GPU box is at hand; its number means nothing.
    rehearse = os.environ.get("MSER_BENCH_REHEARSE") == "gloo"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    red_dev = torch.device("cpu") if rehearse else device          # where the script's own small reductions live
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from mser import _lib
    from model_trainer import ModelTrainer
    lib = _lib.load()
    if rehearse and world > 1:
        # ranks share ONE GPU: their persistent launches (160-208 workgroups each) cannot be co-resident and could hold CUs the other
        # waits for; the rehearsal is about this script's launch / reduce / report logic, so it runs one launch per step instead
        lib.mser_set_option(1, 0)          # MSER_OPT_PERSISTENT = 0

    torch.manual_seed(0)
    # headline = parity configuration (every Dropout p = 0, SURVEY.md 7 "Dropout"); the train-mode step with the reference's
    # dropout probabilities is reported under variants
    tr = ModelTrainer(device, lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_sps", loss="NLL", n_classes=NCLS,
                      dataset="IEMOCAP", d_r=D_R, quiet=True, dropout=False)
    init_attention_weights(tr.model)
    tr.train()
    tr.scheduler.step(0)
    x, qmask, umask, label = synth_batch(1000 + rank, device)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up (eager: allocates the flat store, RCCL communicators, caches)
    n_eager_warm = max(1, min(2, args.warmup))
    for _ in range(n_eager_warm):
        tr.train_step(x, qmask, umask, label)
    torch.cuda.synchronize()
    log("eager warm-up done")

    use_graph = not args.no_graph
    graph = None
    if use_graph:
        tr.optim.sync_hyperparams()
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.forward_backward(x, qmask, umask, label)       # settle allocator state on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss_t = tr.forward_backward(x, qmask, umask, label)
            if world == 1:
                tr.optimizer_step(umask, sync_hp=False)

    def step():
        if graph is not None:
            graph.replay()
            if world > 1:
                tr.optimizer_step(umask, sync_hp=False)
        else:
            tr.train_step(x, qmask, umask, label)

    # Eager launches keep the speaker / LSTHM chains as concurrent counter-linked kernels on separate streams; under capture they
    # are ordered (a graph executor may serialise branches).  Pick the faster launch mode unless one was forced.
    if graph is not None and not args.graph:
        def probe(fn, n=6):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t_graph = probe(step)
        t_eager = probe(lambda: tr.train_step(x, qmask, umask, label))
        flag = torch.tensor([1.0 if t_eager < t_graph else 0.0], device=red_dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)        # every rank must take the same path
        if float(flag) > 0:
            graph = None
            use_graph = False
        log(f"probe: graph {t_graph * 1e3:.3f} ms/step, eager {t_eager * 1e3:.3f} ms/step -> {'graph' if use_graph else 'eager'}")
    log("graph captured" if graph is not None else "eager mode")
    for _ in range(max(0, args.warmup - n_eager_warm)):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ms = torch.tensor([elapsed * 1e3 / args.steps], device=red_dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
    ms_per_step = float(ms)
    log(f"timed region done: {ms_per_step:.3f} ms/step")
    # SURVEY 8(d) asks for the median: the contract's `value` is the mean over the one timed region above; the same K steps once more,
    # each bracketed by its own pair of events on the step's stream, give the median beside it (reported, never `value`)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    per = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    ms_median = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    # a persistent chain that gave up at a bounded wait would have made the steps FASTER and wrong: the sticky fault word says so
    # (Adam skipped those updates on the device), and the loss of one more step must be finite
    from mser import fault
    fault.check(device, "bench.py timed region")
    loss_chk, _ = tr.train_step(x, qmask, umask, label)
    if not bool(torch.isfinite(loss_chk)):
        raise SystemExit(f"bench.py: non-finite loss after the timed region ({float(loss_chk)})")
    fault.check(device, "bench.py check step")

    # ---- live roofline measurement: HIP events around every launch of the LSTHM chain kernels (eager pass, same inputs)
    roofline = None
    if rank == 0 and not args.no_roofline:
        def timed(kernel_id, steps_prof=5):
            _lib.check(lib.mser_prof_enable(kernel_id, 4 * L * steps_prof + 64), "prof_enable")
            for _ in range(steps_prof):
                tr.forward_backward(x, qmask, umask, label)
            torch.cuda.synchronize()
            tot, cnt = ctypes.c_float(0), ctypes.c_int32(0)
            _lib.check(lib.mser_prof_collect(ctypes.byref(tot), ctypes.byref(cnt)), "prof_collect")
            lib.mser_prof_enable(0, 0)
            return tot.value * 1e3 / max(cnt.value, 1), cnt.value      # us per launch, launches

        def entry(name, us, launches, backward):
            steps_per_launch = 2 * L if us > 200 else 2                 # persistent launch = T steps x 2 directions
            by = lsthm_step_bytes(backward) * steps_per_launch
            ach = by / (us * 1e-6) / 1e9
            # `achieved` / `frac` follow SURVEY.md 8(d)'s formula (every weight of the step streamed once per step); the narrower model
            # (`*_touched`: only what this kernel actually touches per step -- U, V, states, gates; W, S and the speaker LSTM weights
            # are hoisted or belong to other roles) is printed beside it
            by_t = by
            by = lsthm_step_bytes_survey(backward) * steps_per_launch
            ach_t = by_t / (us * 1e-6) / 1e9
            ach = by / (us * 1e-6) / 1e9
            return dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=pmc_traffic(name), bytes_per_launch=by, avg_launch_us=round(us, 2),
                        achieved_touched=round(ach_t, 1), frac_touched=round(ach_t / HBM_PEAK_GBS, 4), bytes_per_launch_touched=by_t,
                        launches_timed=launches, steps_per_launch=steps_per_launch,
                        note="dependency-latency bound recurrence (2 inter-workgroup hand-offs per time step); weights are "
                             "register-resident, so real HBM traffic is far below this streaming model")
        us_b, n_b = timed(4)     # MSER_PROF_LSTHM_BWD_ROW: brackets lsthm_bwd_persist (or each lsthm_bwd_row launch)
        us_f, n_f = timed(2)     # MSER_PROF_LSTHM_FWD_GATES: brackets cell_fwd_fused (or each lsthm_fwd_gates launch)
        # kernel = the symbol rocprofv3 reports (profiles/*_kernel_stats.csv): the BPTT launch is cell_bwd_fused (LSTHM BPTT with the
        # speaker BPTT riding in the same grid), the forward launch is cell_fwd_fused (LSTHM chain, speaker chain, statistics roles)
        roofline = entry("cell_bwd_fused" if us_b > 200 else "lsthm_bwd_row", us_b, n_b, True)
        roofline["lsthm_forward"] = entry("cell_fwd_fused" if us_f > 200 else "lsthm_fwd_gates", us_f, n_f, False)

        # ---- MFMA-bound kernels: the batched QK^T and attention.V contractions.  Algorithmic FLOPs per launch
        # (SURVEY.md 8(d)): sequence-level cross-modal attention 4*B*L^2*Dk forward (QK^T + PV), 10*B*L^2*Dk backward (recomputed
        # QK^T, dP, dV, dQ, dK); encoder self-attention 4*B*nh*L^2*dk / 10*B*nh*L^2*dk.  Peak = the fp32 matrix rate (exact-fp32
        # MFMA is what the 1e-4 gate needs): 157.3 TFLOP/s.
        def mfma_entry(name, pid, flops):
            us, n = timed(pid, 3)
            ach = flops / (us * 1e-6) / 1e12 if n else 0.0
            return dict(bound="mfma", kernel=name, achieved=round(ach, 2), peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / FP32_MFMA_PEAK_TFLOPS, 4), traffic=None, flops_per_launch=flops, avg_launch_us=round(us, 2),
                        launches_timed=n, note="launches of this kernel run beside the recurrent chains on side streams; a launch is "
                        "32 x 4 (cross-modal) or 32 x 8 (encoder) workgroups, i.e. at most half the chip, and each is latency-bound "
                        "(LDS staging, softmax, two dependent MFMA chains of K <= 128)")
        Dk, nh, dkh = 128, 8, 40
        roofline["attention_mfma"] = [
            mfma_entry("xattn_fwd_kernel", 7, 4 * B * L * L * Dk),
            mfma_entry("xattn_bwd_kernel", 8, 10 * B * L * L * Dk),
            mfma_entry("attn_fwd_kernel", 9, 4 * B * nh * L * L * dkh),
            mfma_entry("attn_bwd_kernel", 10, 10 * B * nh * L * L * dkh),
        ]

    log("roofline pass done")

    # ---- secondary workloads of SURVEY.md 8(d), eager launches, a few steps each (reported, never the headline value)
    variants = None
    if rank == 0 and world == 1 and not args.no_variants:
        variants = {}
        try:
            launch_mode = {}

            def check_variant(batch, tag):
                """What the headline does after its timed region, for EVERY workload this script times: a chain
                that gave up at a bounded wait would have been faster and wrong -- the sticky fault word says so -- and one more step must
                give a finite loss.  A failure is reported in variants[tag]["error"], never as a number."""
                fault.check(device, f"bench.py variant {tag}")
                loss_v, _ = tr.train_step(*batch)
                if not bool(torch.isfinite(loss_v)):
                    raise RuntimeError(f"non-finite loss after the timed loop ({float(loss_v)})")
                fault.check(device, f"bench.py variant {tag} (check step)")

            def time_steps(batch, n=10, tag=None, graph_ok=True):
                """ms per step of the current trainer `tr` on `batch`: eager launches and, where the step captures (MARN1_sps at any
                width, MARN1_nsps / no_en; MARN1_onlysp and DialogueRNN run their linked / host-loop schedules eagerly), a hipGraph replay of it,
                whichever is faster -- the same choice the headline makes.  (MARN1_onlysp's counter-linked launches are eager-only; the nsps
                variants' speaker chains run beside the encoders on plain stream dependencies and capture.)"""
                for _ in range(3):
                    tr.train_step(*batch)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    tr.train_step(*batch)
                torch.cuda.synchronize()
                ms_e = (time.perf_counter() - t) / n * 1e3
                ms_g = None
                if graph_ok and not args.no_graph and type(tr.model).__name__ in ("MARN1_sps", "MARN1_nsps", "MARN1_no_en"):
                    try:
                        tr.optim.sync_hyperparams()
                        side_s = torch.cuda.Stream(device=device)
                        side_s.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side_s):
                            tr.forward_backward(*batch)
                        torch.cuda.current_stream().wait_stream(side_s)
                        torch.cuda.synchronize()
                        g_ = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_):
                            tr.forward_backward(*batch)
                            tr.optimizer_step(batch[2], sync_hp=False)
                        for _ in range(2):
                            g_.replay()
                        torch.cuda.synchronize()
                        t = time.perf_counter()
                        for _ in range(n):
                            g_.replay()
                        torch.cuda.synchronize()
                        ms_g = (time.perf_counter() - t) / n * 1e3
                        del g_
                    except Exception as e:          # a variant that does not capture is reported eager
                        log(f"variant {tag}: graph capture failed ({type(e).__name__}: {e}); eager")
                        torch.cuda.synchronize()
                if tag is not None:
                    launch_mode[tag] = "hipGraph replay" if (ms_g is not None and ms_g < ms_e) else "eager"
                check_variant(batch, tag)
                return ms_g if (ms_g is not None and ms_g < ms_e) else ms_e
            def guarded(tag, fn):
                """One secondary workload: its failure (a fault word, a non-finite loss, an exception) lands in variants[tag]["error"]
                and the others still run."""
                try:
                    fn()
                except Exception as e:
                    import traceback
                    log(f"variant {tag}: " + traceback.format_exc())
                    variants[tag] = {"error": f"{type(e).__name__}: {e}"}
                    torch.cuda.synchronize()
                    fault.clear(device)

            def with_trainer(tr_new, fn):
                """Run fn() with `tr` (what time_steps / check_variant drive) temporarily replaced."""
                nonlocal tr
                tr_main, tr = tr, tr_new
                try:
                    return fn()
                finally:
                    tr = tr_main

            def new_trainer(model="MARN1_sps", dropout=False, **kw):
                t_ = ModelTrainer(device, lr=1e-3, test_step=1, lr_decay=0.98, model=model, loss="NLL", n_classes=NCLS, dataset="IEMOCAP",
                                  quiet=True, dropout=dropout, **kw)
                if model != "DialogueRNN":
                    init_attention_weights(t_.model)
                t_.train()
                t_.scheduler.step(0)
                return t_

            def v_ragged():
                rb = synth_batch(2000, device, ragged=True)
                ms_r = time_steps(rb, tag="ragged_lengths_U(L/2..L)")
                variants["ragged_lengths_U(L/2..L)"] = {"ms_per_step": round(ms_r, 4), "utterances_per_s": round(float(rb[2].sum()) / (ms_r * 1e-3), 1),
                                                        "note": "masked utterances only; padded steps still run, as in the reference"}
            guarded("ragged_lengths_U(L/2..L)", v_ragged)

            def v_ones():
                with torch.no_grad():
                    saved = {n: p.detach().clone() for n, p in tr.model.named_parameters() if "crossatt" in n}
                    for n, p in tr.model.named_parameters():
                        if "crossatt" in n:
                            p.fill_(1.0)                      # the reference's own initialisation (uniform softmaxes)
                try:
                    ms_o = time_steps((x, qmask, umask, label), tag="attention_weights_as_initialised(ones)")
                finally:
                    with torch.no_grad():
                        for n, p in tr.model.named_parameters():
                            if n in saved:
                                p.copy_(saved[n])
                variants["attention_weights_as_initialised(ones)"] = {"ms_per_step": round(ms_o, 4), "utterances_per_s": round(B * L / (ms_o * 1e-3), 1)}
            guarded("attention_weights_as_initialised(ones)", v_ones)

            def v_b64():
                b64 = synth_batch(3000, device, nb=64)
                ms_b = time_steps(b64, tag="batch_64_per_gpu")
                variants["batch_64_per_gpu"] = {"ms_per_step": round(ms_b, 4), "utterances_per_s": round(64 * L / (ms_b * 1e-3), 1),
                                                "note": "two 32-row blocks per role in the persistent chains: the dependent steps are shared by twice the rows"}
            guarded("batch_64_per_gpu", v_b64)

            # SURVEY 8(f) row f3 (reference model_trainer.py:100-105, dataloader.py:45-47): the step as train_network sees it -- the batch
            # starts in HOST memory (four [L, B, d_t] RoBERTa layers + acoustic features + masks + labels: 50 MB at this shape; the unused
            # visual features are not transferred), unpinned as a DataLoader without pin_memory hands it over, and pinned.  With the
            # prefetch the copies of batch i+1 run on a copy stream while batch i computes; without it they are the reference's blocking
            # copies in front of every step.  The headline `value` keeps its inputs resident in HBM, as the contract says.
            def v_host():
                rs_h = np.random.RandomState(6000)
                nb_h = 6

                def host_batch(pin):
                    r = [torch.tensor(rs_h.standard_normal((L, B, D_R)).astype(np.float32)) for _ in range(4)]
                    data = r + [torch.zeros(L, B, 4), torch.tensor(rs_h.standard_normal((L, B, D_A)).astype(np.float32)),
                                torch.tensor(np.eye(2, dtype=np.float32)[rs_h.randint(0, 2, (L, B))]), torch.ones(B, L),
                                torch.tensor(rs_h.randint(0, NCLS, (B, L)).astype(np.int64))]
                    if pin:
                        data = [t.pin_memory() for t in data]
                    return data + [["v"] * B]
                out = {}
                for pin in (False, True):
                    batches = [host_batch(pin) for _ in range(nb_h)]
                    for prefetch in (True, False):
                        trh = new_trainer(d_r=D_R, prefetch=prefetch)

                        def run():
                            trh.train_network(1, batches[:2])
                            torch.cuda.synchronize()
                            t_ = time.perf_counter()
                            for _ in range(2):
                                trh.train_network(1, batches)
                            torch.cuda.synchronize()
                            return (time.perf_counter() - t_) / (2 * nb_h) * 1e3
                        ms_h_ = with_trainer(trh, run)
                        fault.check(device, "bench.py variant trainer_step_from_host_batch")
                        out[("pinned" if pin else "pageable") + ("_prefetch" if prefetch else "_blocking")] = round(ms_h_, 4)
                        del trh
                    del batches
                variants["trainer_step_from_host_batch"] = {
                    "ms_per_step": out["pageable_prefetch"], "ms_per_step_by_mode": out,
                    "utterances_per_s": round(B * L / (out["pageable_prefetch"] * 1e-3), 1),
                    "host_bytes_per_step": 4 * (4 * L * B * D_R + L * B * D_A + 2 * L * B + B * L) + 8 * B * L,
                    "note": "ModelTrainer.train_network from HOST batches (eager launches, epoch-end synchronisation included): *_prefetch = "
                            "staged through reused page-locked buffers and copied on a copy stream one batch ahead; *_blocking = the "
                            "reference's schedule (copies on the compute stream in front of the step); pageable / pinned = what the loader hands over"}
            guarded("trainer_step_from_host_batch", v_host)

            # BASELINE.json configs[1] names hid=256 (in bf16; this build computes in fp32 to hold the 1e-4 logit gate): same batch, a
            # second trainer at the wider cell.
            def v_h256():
                tr256 = new_trainer(d_r=D_R, hidden=256)
                ms_h = with_trainer(tr256, lambda: time_steps((x, qmask, umask, label), tag="hidden_256_f32"))
                del tr256
                variants["hidden_256_f32"] = {"ms_per_step": round(ms_h, 4), "utterances_per_s": round(B * L / (ms_h * 1e-3), 1),
                                              "note": "configs[1] width"}
            guarded("hidden_256_f32", v_h256)

            # train mode as the reference runs it: all 13 dropout sites live (p = 0.1 encoders, 0.2 attention, 0.5 elsewhere)
            def v_drop():
                trd = new_trainer(d_r=D_R, dropout=True)
                ms_d = with_trainer(trd, lambda: time_steps((x, qmask, umask, label), tag="dropout_on"))
                del trd
                variants["dropout_on"] = {"ms_per_step": round(ms_d, 4), "utterances_per_s": round(B * L / (ms_d * 1e-3), 1),
                                          "note": "all 13 sites live; counter-based masks, re-evaluated in the backward instead of stored"}
            guarded("dropout_on", v_drop)

            # MARN1_onlysp, the reference CLI's default model (GRU speaker state per dialogue), same batch
            for tag, mname, dp in (("marn1_onlysp", "MARN1_onlysp", False), ("marn1_onlysp_dropout_on", "MARN1_onlysp", True),
                                   ("marn1_nsps", "MARN1_nsps", False), ("marn1_no_en", "MARN1_no_en", False)):
                def v_f1(tag=tag, mname=mname, dp=dp):
                    tro = new_trainer(model=mname, d_r=D_R, dropout=dp)
                    ms_o2 = with_trainer(tro, lambda: time_steps((x, qmask, umask, label), tag=tag))
                    del tro
                    variants[tag] = {"ms_per_step": round(ms_o2, 4), "utterances_per_s": round(B * L / (ms_o2 * 1e-3), 1),
                                     "note": "SURVEY 8(f) f1; GRU speaker chains" + (" counter-linked to the LSTHM chains (concurrent launches on two streams)"
                                                                                      if mname == "MARN1_onlysp" else " (listener blend), then the LSTHM chains")}
                guarded(tag, v_f1)

            # BASELINE.json configs[3]: DialogueRNN-style global / party / listener / emotion GRUs with attention over the growing history
            # (model/DialogueRNN.py BiModel as model_trainer.py:35-47 builds it), B = 64 dialogues x L = 200 utterances, D_m = 712
            def v_drnn():
                trd = new_trainer(model="DialogueRNN")
                rs = np.random.RandomState(4000)
                Bd, Ld, Dmd = 64, 200, 712
                Ud = torch.tensor(rs.standard_normal((Ld, Bd, Dmd)).astype(np.float32)).to(device)
                qd = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (Ld, Bd))]).to(device)
                ud = torch.ones(Bd, Ld, device=device)
                ld_ = torch.tensor(rs.randint(0, NCLS, (Bd, Ld)).astype(np.int64)).to(device)
                ms_dr = with_trainer(trd, lambda: time_steps((Ud, qd, ud, ld_), n=4, tag="dialoguernn_bimodel_B64_L200"))
                # the same step with the time loop issued launch by launch (MSER_OPT_DRNN_PERSISTENT = 0: 9 + 13 launches per step)
                from mser import ops
                ops.set_option(ops.MSER_OPT_DRNN_PERSISTENT, 0)
                try:
                    ms_dr0 = with_trainer(trd, lambda: time_steps((Ud, qd, ud, ld_), n=3, tag="dialoguernn_bimodel_B64_L200_per_step_launches"))
                finally:
                    ops.set_option(ops.MSER_OPT_DRNN_PERSISTENT, 1)
                del trd
                gflop = 3 * 2 * Ld * 2 * (Bd * (500 * 1500 * 2 + 500 * 1500 + 2 * 500 * 1500 + 500 * 1500 + 2 * 500 * 1500 + 500 * 900 + 300 * 900)
                                          + Bd * 712 * (3 * 1500 + 500)) / 1e9
                variants["dialoguernn_bimodel_B64_L200"] = {
                    "ms_per_step": round(ms_dr, 3), "utterances_per_s": round(Bd * Ld / (ms_dr * 1e-3), 1),
                    "fp32_mfma_frac": round(gflop / (ms_dr * 1e-3) / 1e3 / 157.0, 4),
                    "ms_per_step_per_step_launches": round(ms_dr0, 3),
                    "note": f"configs[3]; ~{gflop:.0f} GFLOP of exact-fp32 GEMM work per training step against the 157 TFLOP/s fp32 matrix peak; "
                            "the time loop of each pass is ONE persistent launch (ms_per_step_per_step_launches: the same step launch by launch)"}
            guarded("dialoguernn_bimodel_B64_L200", v_drnn)

            # BASELINE.json configs[4], one GPU's shard of it: hid = 1024 with the 8-head sequence attention, global batch 256 over 8 GPUs =
            # 32 dialogues x L = 256 per GPU.  Above hid = 256 the weights (134 MB per phase) no longer fit the register files of
            # co-resident workgroups: weights streamed from HBM / the infinity cache.
            def v_h1024():
                tr5 = new_trainer(d_r=D_R, hidden=1024, xattn_heads=8)
                L5 = 256
                rs = np.random.RandomState(5000)
                x5 = torch.tensor(rs.standard_normal((L5, B, D_R + D_A)).astype(np.float32)).to(device)
                q5 = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (L5, B))]).to(device)
                u5 = torch.ones(B, L5, device=device)
                l5 = torch.tensor(rs.randint(0, NCLS, (B, L5)).astype(np.int64)).to(device)

                def run():
                    tr.train_step(x5, q5, u5, l5)
                    torch.cuda.synchronize()
                    t5 = time.perf_counter()
                    for _ in range(3):
                        tr.train_step(x5, q5, u5, l5)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t5) / 3 * 1e3
                    check_variant((x5, q5, u5, l5), "hid1024_8head_B32_L256_shard_of_configs4")
                    return ms
                ms_5 = with_trainer(tr5, run)
                del tr5, x5, q5, u5, l5
                torch.cuda.empty_cache()
                wbytes = 2 * 2 * 4 * 1024 * 1024 * 4 * (3 + 2)       # per time step: 2 directions x 2 streams/cells x [4H, H] fp32 x (U, V, S + W_ih, W_hh)
                variants["hid1024_8head_B32_L256_shard_of_configs4"] = {
                    "ms_per_step": round(ms_5, 3), "utterances_per_s": round(B * L5 / (ms_5 * 1e-3), 1),
                    "weight_stream_GBps": round(3 * wbytes * L5 / (ms_5 * 1e-3) / 1e9, 1),
                    "note": "configs[4] per-GPU shard (global batch 256 / 8 GPUs); eager; the recurrent weights "
                            f"({wbytes / 1e6:.0f} MB) are re-read every step of the forward, the BPTT and (once more, hoisted) the weight gradients: "
                            "weight_stream_GBps is that traffic over the step time, against 8000 GB/s"}
            guarded("hid1024_8head_B32_L256_shard_of_configs4", v_h1024)

            for k_, m_ in launch_mode.items():
                if k_ in variants and "error" not in variants[k_]:
                    variants[k_]["launch"] = m_
            log("variants done")
        except Exception as e:      # a secondary workload must never take the headline line down with it
            import traceback
            log("variants: " + traceback.format_exc())
            variants["error"] = f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        log("cpu baseline done")

    if rank == 0:
        out = {
            "metric": "utterances/sec fwd+bwd (B=32, T=128, d_a=100, d_t=768) @1/2/4/8 GPU",
            "value": round(world * B * L / (ms_per_step * 1e-3), 1),
            "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(ms_median, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"lsthm_sps + cross-modal attn train step (fwd+bwd+Adam), per-GPU batch={B} seq={L} "
                                   f"d_text={D_R} d_audio={D_A} hid={H} (reference width), {NCLS} classes, dropout p=0 (parity configuration)",
                       "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "variants": variants,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
