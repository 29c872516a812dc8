"""A few eager training steps of MARN1_sps on the bench batch at a chosen width (profile target).
usage: sps_steps.py [hidden] [steps] [B] [L] [xattn_heads]   (configs[4] shard: 1024 3 32 256 8)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
import bench
from model_trainer import ModelTrainer
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
if len(sys.argv) > 4:
    bench.L = int(sys.argv[4])
heads = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=768, hidden=H, xattn_heads=heads, quiet=True, dropout=False)
bench.init_attention_weights(tr.model)
tr.train(); tr.scheduler.step(0)
batch = bench.synth_batch(1000, dev, nb=nb)
for _ in range(2):
    tr.train_step(*batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    tr.train_step(*batch)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / n * 1e3
from mser import fault
fault.check(dev, "sps_steps")
print(f"MARN1_sps hidden={H} B={nb} L={bench.L} heads={heads}: {ms:.3f} ms/step (eager), {nb * bench.L / ms * 1e3:.0f} utterances/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
