"""A few eager training steps of MARN1_onlysp on the bench batch (scratch: profile with rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
import bench
from model_trainer import ModelTrainer
dev = torch.device("cuda:0")
tr = ModelTrainer(dev, lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_onlysp", loss="NLL", n_classes=6, dataset="IEMOCAP", d_r=768,
                  quiet=True, dropout=("--dropout" in sys.argv))
bench.init_attention_weights(tr.model)
tr.train(); tr.scheduler.step(0)
x, qmask, umask, label = bench.synth_batch(1000, dev)
for i in range(12):
    loss, _ = tr.train_step(x, qmask, umask, label)
torch.cuda.synchronize()
print("loss", float(loss))
import time
t0 = time.perf_counter()
for i in range(20):
    tr.train_step(x, qmask, umask, label)
torch.cuda.synchronize()
print(f"MARN1_onlysp: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step (eager)")
# hipGraph replay of the same step (under capture the GRU and LSTHM chains run one after the other: the counter links are eager-only)
if "--graph" in sys.argv:
    tr.optim.sync_hyperparams()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr.forward_backward(x, qmask, umask, label)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tr.forward_backward(x, qmask, umask, label)
        tr.optimizer_step(umask, sync_hp=False)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    print(f"MARN1_onlysp: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step (hipGraph replay)")
    from mser import fault
    fault.check(dev, "onlysp graph replay")
