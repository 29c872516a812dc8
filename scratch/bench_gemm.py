import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
from mser import ops

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dev = "cuda"
M = 4096
res = []
def lin(K, N):
    x, W, out = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.linear(x, W, out))
    res.append((f"linear  [4096x{K}]@[{N}x{K}]^T", t, 2 * M * K * N))
def mm(K, N):
    x, W, out = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.matmul(x, W, out))
    res.append((f"matmul  [4096x{K}]@[{K}x{N}]", t, 2 * M * K * N))
def wg(No, Ki, sk):
    dy, x, dW = torch.randn(M, No, device=dev), torch.randn(M, Ki, device=dev), torch.zeros(No, Ki, device=dev)
    t = timeit(lambda: ops.grad_weight(dy, x, dW, splitk=sk))
    res.append((f"wgrad   dW[{No}x{Ki}] K=4096 splitk={sk}", t, 2 * M * No * Ki))
def attn(nb, nh, L, d):
    q, k = torch.randn(L * nb, nh * d, device=dev), torch.randn(L * nb, nh * d, device=dev)
    S = torch.empty(nb, nh, L, L, device=dev)
    t = timeit(lambda: ops.gemm_raw(q, k, S, L, L, d, nb * nh * d, 1, 1, nb * nh * d, L, batch=(nb, nh), sA=(nh * d, d), sB=(nh * d, d), sC=(nh * L * L, L * L)))
    res.append((f"QK^T    {nb}x{nh} [{L}x{d}]@[{d}x{L}]", t, 2 * nb * nh * L * L * d))
    o = torch.empty(L * nb, nh * d, device=dev)
    t = timeit(lambda: ops.gemm_raw(S, k, o, L, d, L, L, 1, nb * nh * d, 1, nb * nh * d, batch=(nb, nh), sA=(nh * L * L, L * L), sB=(nh * d, d), sC=(nh * d, d)))
    res.append((f"P@V     {nb}x{nh} [{L}x{L}]@[{L}x{d}]", t, 2 * nb * nh * L * L * d))
lin(100, 320); lin(768, 100); lin(1280, 100); lin(320, 100); lin(100, 512); lin(128, 512); lin(100, 40); lin(40, 100)
mm(512, 128); mm(512, 100); mm(100, 320); mm(100, 128); mm(128, 128); mm(128, 1280)
for sk in (8, 16, 32, 64): wg(512, 128, sk)
wg(512, 100, 16); wg(320, 100, 16); wg(100, 1280, 16); wg(100, 768, 16)
attn(32, 8, 128, 40); attn(32, 1, 128, 128)
for name, t, fl in res:
    print(f"{name:48s} {t:8.2f} us  {fl / t / 1e6:8.2f} TFLOP/s")
