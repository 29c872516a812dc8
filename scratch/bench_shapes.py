"""Times every distinct GEMM shape of one training step in isolation (50 back-to-back launches, HIP events).
usage: bench_shapes.py [shapes.txt]  -- shapes from scratch/gemm_shapes.py (committed copy: scratch/gemm_shapes.txt)"""
import os, sys, re, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
from mser import _lib
lib = _lib.load()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scratch", "gemm_shapes.txt")
cnt = collections.Counter(l.strip() for l in open(path) if l.startswith("[gemm]"))
dev = torch.device("cuda:0")
pool = torch.randn(64 << 20, device=dev)           # 256 MB of operands
outb = torch.zeros(16 << 20, device=dev)
tot = 0.0
rows = []
for line, n in cnt.items():
    f = dict(re.findall(r"(\w+) (-?\d+)", line.replace("[gemm] ", "")))
    M, N, K, b, sk = int(f["M"]), int(f["N"]), int(f["K"]), int(f["b"]), int(f["splitk"])
    d = _lib.GemmDesc()
    d.A, d.B, d.C = pool.data_ptr(), pool.data_ptr() + (128 << 20), outb.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.sAm, d.sAk, d.sBk, d.sBn, d.ldc = int(f["sAm"]), int(f["sAk"]), int(f["sBk"]), int(f["sBn"]), int(f["ldc"])
    d.batch1, d.batch2 = b, 1
    # batch strides: guess a dense layout that stays inside the pools
    d.sA1 = 128 if d.sAm >= 4096 or d.sAk >= 4096 else M * K
    d.sB1 = 128 if d.sBk >= 4096 or d.sBn >= 4096 else N * K
    d.sC1 = 128 if d.ldc >= 4096 else M * N
    if b > 1:
        d.sA1 = (d.sA1 + 3) // 4 * 4; d.sB1 = (d.sB1 + 3) // 4 * 4
    d.alpha = 1.0; d.splitk = sk; d.flags = int(f["flags"]) & ~2 if sk > 1 else int(f["flags"])
    if int(f["bias"]): d.bias = pool.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    def run(k):
        for _ in range(k):
            rc = lib.mser_gemm(ctypes.byref(d), st)
            assert rc == 0, lib.mser_last_error()
    run(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); run(50); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
    fl = 2.0 * M * N * K * b
    rows.append((best * n, n, best, fl / best / 1e6, M, N, K, b, sk, f["modes"], f["grid"]))
    tot += best * n
rows.sort(reverse=True)
print(f"total GEMM time per step (isolated, serial): {tot:.0f} us over {sum(cnt.values())} calls")
for r in rows:
    print(f"{r[0]:7.1f} us  n={r[1]:2d}  {r[2]:6.2f} us/call {r[3]:6.1f} TF  M {r[4]:5d} N {r[5]:5d} K {r[6]:5d} b {r[7]:3d} sk {r[8]:2d} modes {r[9]} grid {r[10]}")
