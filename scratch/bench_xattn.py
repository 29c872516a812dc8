"""mser_xattn_seq_fwd / bwd alone at the bench shape (nb = 32, L = 128, one head of 128) and at 8 heads of 16: us per launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
from mser import ops, _lib as L_
for nb, L, nh, dk in ((32, 128, 1, 128), (32, 128, 8, 16), (32, 96, 1, 128)):
    D = nh * dk
    q = torch.randn(L * nb, D, device="cuda"); kv = torch.randn(L * nb, 2 * D, device="cuda"); dO = torch.randn(L * nb, D, device="cuda")
    out = torch.zeros(L * nb, D, device="cuda"); stats = torch.empty(nb, nh, L, 2, device="cuda")
    dq = torch.empty_like(q); dkv = torch.zeros_like(kv)
    d = L_.XAttnDesc()
    d.nb, d.nh, d.Lq, d.Lk, d.dk = nb, nh, L, L, dk
    d.q, d.ldq, d.k, d.ldk, d.v, d.ldv = q.data_ptr(), D, kv.data_ptr(), 2 * D, kv[:, D:].data_ptr(), 2 * D
    d.sbq, d.slq, d.sbk, d.slk = 1, nb, 1, nb
    d.o, d.ldo, d.stats, d.scale = out.data_ptr(), D, stats.data_ptr(), 1.0 / np.sqrt(dk)
    d.dO, d.lddo, d.dq, d.lddq = dO.data_ptr(), D, dq.data_ptr(), D
    d.dk_, d.lddk, d.dv, d.lddv = dkv.data_ptr(), 2 * D, dkv[:, D:].data_ptr(), 2 * D
    for name, fn in (("fwd", ops.xattn_seq_fwd), ("bwd", ops.xattn_seq_bwd)):
        for _ in range(3): fn(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn(d)
        e1.record(); torch.cuda.synchronize()
        print(f"xattn {name} nb={nb} L={L} nh={nh} dk={dk}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
