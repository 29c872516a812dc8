"""Step time of the trainer at other batch sizes (scratch).  usage: time_batch.py B [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
import bench
from model_trainer import ModelTrainer
for B in [int(a) for a in sys.argv[1].split(",")]:
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    bench.B, bench.L = B, L
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=768, quiet=True, dropout=False)
    bench.init_attention_weights(tr.model)
    tr.train(); tr.scheduler.step(0)
    batch = bench.synth_batch(1, dev)
    for _ in range(3): tr.train_step(*batch)
    torch.cuda.synchronize()
    t = time.perf_counter(); n = 10
    for _ in range(n): tr.train_step(*batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    print(f"B={B} L={L}: {ms:.2f} ms/step, {B * L / ms * 1e3:.0f} utt/s", flush=True)
