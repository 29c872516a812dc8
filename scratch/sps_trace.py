"""A few hipGraph-replayed training steps of MARN1_sps on the bench batch (kernel-trace target for scratch/step_timeline.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
import bench
from model_trainer import ModelTrainer
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, 1e-3, 1, 0.98, "MARN1_sps", "NLL", 6, "IEMOCAP", d_r=768, quiet=True, dropout=False)
bench.init_attention_weights(tr.model)
tr.train(); tr.scheduler.step(0)
batch = bench.synth_batch(1000, dev)
for _ in range(3):
    tr.train_step(*batch)
torch.cuda.synchronize()
tr.optim.sync_hyperparams()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    tr.forward_backward(*batch)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.forward_backward(*batch)
    tr.optimizer_step(batch[2], sync_hp=False)
for _ in range(8):
    g.replay()
torch.cuda.synchronize()
