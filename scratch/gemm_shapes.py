"""Logs every GEMM of one training step (MSER_GEMM_LOG=1) -> gpurun_out/gemm_shapes.txt"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path[:0] = [%r, os.path.join(%r, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch, bench
from model_trainer import ModelTrainer
tr = ModelTrainer(torch.device("cuda:0"), lr=1e-3, test_step=1, lr_decay=0.98, model="MARN1_sps", loss="NLL", n_classes=6, dataset="IEMOCAP", d_r=768, quiet=True, dropout=False)
tr.train(); tr.scheduler.step(0)
x, q, u, l = bench.synth_batch(1000, torch.device("cuda:0"))
tr.train_step(x, q, u, l); torch.cuda.synchronize()
print("=== STEP", file=sys.stderr, flush=True)
tr.train_step(x, q, u, l); torch.cuda.synchronize()
''' % (ROOT, ROOT)
env = dict(os.environ, MSER_GEMM_LOG="1")
r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
lines = r.stderr.split("=== STEP")[-1].splitlines()
out = [l for l in lines if l.startswith("[gemm]")]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.txt"), "w").write("\n".join(out) + "\n")
print(len(out), "gemm calls logged"); print(r.stderr[-500:] if not out else "")
