"""Soak: DialogueRNN BiModel training steps at configs[3] size, dropout on, persistent launches: finite loss, fault word clean.
usage: soak_drnn.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np
import torch
from model_trainer import ModelTrainer
from mser import fault
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, 1e-4, 1, 0.98, "DialogueRNN", "NLL", 6, "IEMOCAP", quiet=True, dropout=True)
tr.train(); tr.scheduler.step(0)
rs = np.random.RandomState(4100)
B, L, Dm = 64, 200, 712
losses = []
t0 = time.time()
for i in range(n):
    Li = L if i % 3 else int(rs.randint(L // 2, L + 1))
    U = torch.tensor(rs.standard_normal((Li, B, Dm)).astype(np.float32)).to(dev)
    q = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (Li, B))]).to(dev)
    lens = rs.randint(Li // 2, Li + 1, B); lens[0] = Li
    um = torch.tensor((np.arange(Li)[None, :] < lens[:, None]).astype(np.float32)).to(dev)
    lab = torch.tensor(rs.randint(0, 6, (B, Li)).astype(np.int64)).to(dev)
    out = tr.train_step(U, q, um, lab)
    if i % 50 == 0 or i == n - 1:
        torch.cuda.synchronize()
        fault.check(dev, f"soak step {i}")
        lv = float(out[0]) if isinstance(out, (tuple, list)) else float(out)
        losses.append(lv)
        assert np.isfinite(lv), (i, lv)
        print(f"step {i}: loss {lv:.4f}  ({time.time() - t0:.1f} s)", flush=True)
print("ok", losses)
