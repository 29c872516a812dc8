"""A few eager training steps of the DialogueRNN BiModel at BASELINE configs[3] (B = 64, L = 200, D_m = 712) -- profile target
(rocprofv3 --kernel-trace --stats) and step timer.  usage: drnn_steps.py [steps] [graph]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np
import torch
from model_trainer import ModelTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, 1e-3, 1, 0.98, "DialogueRNN", "NLL", 6, "IEMOCAP", quiet=True, dropout=False)
tr.train(); tr.scheduler.step(0)
rs = np.random.RandomState(4000)
B, L, Dm = 64, 200, 712
U = torch.tensor(rs.standard_normal((L, B, Dm)).astype(np.float32)).to(dev)
q = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (L, B))]).to(dev)
um = torch.ones(B, L, device=dev)
lab = torch.tensor(rs.randint(0, 6, (B, L)).astype(np.int64)).to(dev)
for _ in range(2):
    tr.train_step(U, q, um, lab)
torch.cuda.synchronize()
step = lambda: tr.train_step(U, q, um, lab)
if use_graph:
    tr.optim.sync_hyperparams()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr.forward_backward(U, q, um, lab)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tr.forward_backward(U, q, um, lab)
        tr.optimizer_step(um, sync_hp=False)
    step = g.replay
    step(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / n * 1e3
print(f"DialogueRNN BiModel B={B} L={L}: {ms:.2f} ms/step ({'graph' if use_graph else 'eager'}), {B * L / ms * 1e3:.0f} utterances/s")
