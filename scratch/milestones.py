"""GPU-side milestones of one eager training step on the main stream (HIP events, no profiler): where the step's wall time goes
between the recurrent chains.  usage: python scratch/milestones.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
import bench
from mser import ops
from model_trainer import ModelTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = ModelTrainer(dev, lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_sps", loss="NLL", n_classes=6, dataset="IEMOCAP", d_r=768, quiet=True, dropout=("--dropout" in sys.argv))
bench.init_attention_weights(tr.model)
tr.train(); tr.scheduler.step(0)
x, qmask, umask, label = bench.synth_batch(1000, dev)
marks = []
orig = ops.marn_cell_run
def patched(desc, phases):
    if phases & (ops.PHASE_LSTHM_FWD | ops.PHASE_LSTHM_BWD):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        orig(desc, phases)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append(("chain_fwd" if phases & ops.PHASE_LSTHM_FWD else "chain_bwd", e0, e1))
    else:
        orig(desc, phases)
ops.marn_cell_run = patched
import mser.model_fn as mf
mf.ops.marn_cell_run = patched
def run_config(tag):
    for _ in range(5):
        tr.train_step(x, qmask, umask, label)
    torch.cuda.synchronize()
    res, allm = [], []
    for it in range(12):          # no host sync inside the loop: the host runs ahead of the GPU exactly as in bench.py
        marks.clear()
        s = torch.cuda.Event(enable_timing=True); s.record()
        tr.train_step(x, qmask, umask, label)
        e = torch.cuda.Event(enable_timing=True); e.record()
        allm.append((s, list(marks), e))
    torch.cuda.synchronize()
    for s, mk, e in allm[2:]:
        (n1, a0, a1), (n2, b0, b1) = mk
        res.append([s.elapsed_time(a0), a0.elapsed_time(a1), a1.elapsed_time(b0), b0.elapsed_time(b1), b1.elapsed_time(e), s.elapsed_time(e)])
    r = np.median(np.array(res), axis=0) * 1e3
    print(tag, "us: pre-chain %.0f | chain fwd %.0f | head+loss+head-bwd %.0f | chain bwd %.0f | post-chain %.0f | total %.0f" % tuple(r))


for rep in range(2):
    run_config("dropout=%s" % ("--dropout" in sys.argv))
