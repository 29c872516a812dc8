"""Isolated timing of the fused EncoderLayer launches at the bench shape (B=32, L=128, D=100, 8 heads x 40, d_inner 40).
usage: python scratch/bench_encoder.py   (run under `rocprofv3 --kernel-trace --stats` for per-kernel durations)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
from mser import functional as F_, ops
from mser.functional import Layout
nb, L, D, nh, dk, dff = 32, 128, 100, 8, 40, 40
rs = np.random.RandomState(0)
shapes = {"slf_attn.w_qs.weight": (nh * dk, D), "slf_attn.w_ks.weight": (nh * dk, D), "slf_attn.w_vs.weight": (nh * dk, D),
          "slf_attn.fc.weight": (D, nh * dk), "slf_attn.layer_norm.weight": (D,), "slf_attn.layer_norm.bias": (D,),
          "pos_ffn.w_1.weight": (dff, D), "pos_ffn.w_1.bias": (dff,), "pos_ffn.w_2.weight": (D, dff), "pos_ffn.w_2.bias": (D,),
          "pos_ffn.layer_norm.weight": (D,), "pos_ffn.layer_norm.bias": (D,)}
flat = torch.tensor(rs.standard_normal(3 * nh * dk * D).astype(np.float32) * 0.1).cuda()
P = {n: torch.tensor(rs.standard_normal(s).astype(np.float32) * 0.1).cuda() for n, s in shapes.items()}
for i, n in enumerate(("slf_attn.w_qs.weight", "slf_attn.w_ks.weight", "slf_attn.w_vs.weight")):
    P[n] = flat[i * nh * dk * D:(i + 1) * nh * dk * D].view(nh * dk, D)
G = {n: torch.zeros_like(t) for n, t in P.items()}
x = torch.tensor(rs.standard_normal((nb * L, D)).astype(np.float32)).cuda()
do = torch.tensor(rs.standard_normal((nb * L, D)).astype(np.float32)).cuda()
lay = Layout.time_major(L, nb)
def fwd():
    return F_.encoder_layer_fwd(x, None, P.__getitem__, lay, nh, dk, dk)
out, c = fwd()
def bwd():
    return F_.encoder_layer_bwd(c, do, P.__getitem__, G.__getitem__)
for fn, name in ((fwd, "fwd"), (bwd, "bwd (act + grouped wgrad)")):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per call")
