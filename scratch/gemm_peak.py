"""Ceiling of the generic fp32 MFMA GEMM (csrc/gemm.hip) at sizes that fill the chip.  usage: gemm_peak.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
from mser import ops
for (M, N, K, what) in ((4096, 4096, 4096, "square"), (4096, 512, 100, "pre-activation product"), (1024, 256, 4096, "hid=256 weight gradient (as matmul)"),
                        (8192, 4096, 1024, "hid=1024 hoisted")):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.matmul(x, w, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        ops.matmul(x, w, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{what}: M={M} N={N} K={K}: {dt * 1e6:.1f} us, {2 * M * N * K / dt / 1e12:.1f} TFLOP/s (fp32 MFMA peak 157)")
