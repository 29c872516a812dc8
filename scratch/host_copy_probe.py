"""How fast does a host batch reach page-locked memory / the device?  (numbers quoted in DESIGN 5)"""
import time, torch
from concurrent.futures import ThreadPoolExecutor
n = 128 * 32 * 768
src = [torch.randn(n) for _ in range(4)]
pin = [torch.empty(n).pin_memory() for _ in range(4)]
dev = [torch.empty(n, device="cuda") for _ in range(4)]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
mb = 4 * n * 4 / 1e6
print(f"{mb:.0f} MB per call; torch threads {torch.get_num_threads()}")
print("pageable -> pinned, copy_ :", round(t(lambda: [p.copy_(s) for p, s in zip(pin, src)]), 2), "ms")
pool = ThreadPoolExecutor(8)
def par():
    fs = []
    for p, s in zip(pin, src):
        for c in range(4):
            sl = slice(c * n // 4, (c + 1) * n // 4)
            fs.append(pool.submit(p[sl].copy_, s[sl]))
    for f in fs: f.result()
print("pageable -> pinned, 16 chunks on 8 threads:", round(t(par), 2), "ms")
print("pinned -> device (non_blocking):", round(t(lambda: [d.copy_(p, non_blocking=True) for d, p in zip(dev, pin)]), 2), "ms")
print("pageable -> device (.to):", round(t(lambda: [d.copy_(s) for d, s in zip(dev, src)]), 2), "ms")
pg = [torch.empty(n) for _ in range(4)]
print("pageable -> pageable, copy_:", round(t(lambda: [p.copy_(s) for p, s in zip(pg, src)]), 2), "ms")
