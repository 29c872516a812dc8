"""Time the big DialogueRNN / hid1024 GEMM shapes (MSER_GEMM_LOG=1 shows the configuration).  usage: gemm_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
from mser import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(12800, 1500, 712), (12800, 500, 712)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    us = t(lambda: ops.linear(x, W, out))
    print(f"NN  M={M} N={N} K={K}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s")
for (rows, N, K) in [(12800, 1500, 712), (12800, 1500, 500), (25600, 1500, 500), (8192, 4096, 1024)]:
    dy = torch.randn(rows, N, device="cuda"); x = torch.randn(rows, K, device="cuda"); gW = torch.zeros(N, K, device="cuda")
    us = t(lambda: ops.grad_weight(dy, x, gW, splitk=2))
    print(f"TN  rows={rows} N={N} K={K}: {us:.1f} us  {2*rows*N*K/us/1e6:.1f} TFLOP/s")
