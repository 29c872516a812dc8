"""Soak of MARN1_onlysp in train mode (all dropout sites live): n eager steps, then a captured step replayed n times (scratch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
import bench
from model_trainer import ModelTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_onlysp", loss="NLL", n_classes=6, dataset="IEMOCAP", d_r=768, quiet=True)
bench.init_attention_weights(tr.model)
tr.train(); tr.scheduler.step(0)
x, qmask, umask, label = bench.synth_batch(1000, dev, ragged=True)
for i in range(n):
    loss, _ = tr.train_step(x, qmask, umask, label)
    if i % 100 == 0:
        print("eager", i, float(loss), flush=True)
torch.cuda.synchronize()
assert torch.isfinite(loss)
tr.optim.sync_hyperparams()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    tr.forward_backward(x, qmask, umask, label)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lt = tr.forward_backward(x, qmask, umask, label)
    tr.optimizer_step(umask, sync_hp=False)
t1 = time.time()
for i in range(n):
    g.replay()
    if i % 100 == 0:
        torch.cuda.synchronize(); print("graph", i, float(lt), flush=True)
torch.cuda.synchronize()
print("graph ms/step", (time.time() - t1) / n * 1e3)
assert torch.isfinite(lt)
print("soak ok: final loss", float(lt))
