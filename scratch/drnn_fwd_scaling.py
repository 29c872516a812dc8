"""DialogueRNN BiModel forward only (no_grad) at several L: does drnn_fwd_persist scale with the step count?  usage: drnn_fwd_scaling.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np
import torch
from model_trainer import ModelTrainer
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, 1e-3, 1, 0.98, "DialogueRNN", "NLL", 6, "IEMOCAP", quiet=True, dropout=False)
tr.model.eval()
rs = np.random.RandomState(4000)
B, Dm = 64, 712
for L in (50, 100, 200):
    U = torch.tensor(rs.standard_normal((L, B, Dm)).astype(np.float32)).to(dev)
    q = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (L, B))]).to(dev)
    um = torch.ones(B, L, device=dev)
    with torch.no_grad():
        for _ in range(2):
            tr.model(U, q, um)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            tr.model(U, q, um)
        torch.cuda.synchronize()
    print(f"L={L}: forward {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
