"""Where does the time of the host->device batch pipeline go?  (pageable host batches, bench shape)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
from model_trainer import BatchPrefetcher
L, B, D_R = 128, 32, 768
rs = np.random.RandomState(0)
def host_batch():
    r = [torch.tensor(rs.standard_normal((L, B, D_R)).astype(np.float32)) for _ in range(4)]
    return r + [torch.zeros(L, B, 4), torch.tensor(rs.standard_normal((L, B, 100)).astype(np.float32)),
                torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (L, B))]), torch.ones(B, L),
                torch.tensor(rs.randint(0, 6, (B, L)).astype(np.int64)), ["v"] * B]
batches = [host_batch() for _ in range(6)]
print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
dev = torch.device("cuda:0")
for nthreads in (None, 16, 4):
    if nthreads: torch.set_num_threads(nthreads)
    pf = BatchPrefetcher(dev, True)
    for rep in range(3):
        t0 = time.perf_counter()
        for b in pf(batches):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(batches) * 1e3
    print(f"threads {torch.get_num_threads()}: {dt:.2f} ms per batch through the prefetcher (no compute)")
    t0 = time.perf_counter()
    for b in batches:
        for f in (0, 1, 2, 3, 5):
            pf._pinned(0, f, b[f])
    print(f"   staging copies alone: {(time.perf_counter() - t0) / len(batches) * 1e3:.2f} ms per batch")
