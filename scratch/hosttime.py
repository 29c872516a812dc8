import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
import bench
from model_trainer import ModelTrainer
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_sps", loss="NLL", n_classes=6, dataset="IEMOCAP", d_r=768, quiet=True, dropout=False)
bench.init_attention_weights(tr.model); tr.train()
x, qmask, umask, label = bench.synth_batch(1000, dev)
for _ in range(3): tr.train_step(x, qmask, umask, label)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): tr.train_step(x, qmask, umask, label)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.3f} ms/step ; total {1e3*(t2-t0)/n:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step(x, qmask, umask, label)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
