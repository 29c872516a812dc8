"""Isolated timing of the fused recurrent launches (product build): ndir = 1 vs 2, forward fused vs separate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import numpy as np, torch
from mser import ops
from models.lsthm_sps import MARN_cell
torch.manual_seed(0)
T, N = 128, 32
rs = np.random.RandomState(0)
x_l = torch.tensor(rs.standard_normal((T, N, 100)).astype(np.float32)).cuda()
x_a = torch.tensor(rs.standard_normal((T, N, 100)).astype(np.float32)).cuda()
qmask = torch.tensor(np.eye(2, dtype=np.float32)[rs.randint(0, 2, (T, N))]).cuda()
rev = torch.arange(T, device="cuda", dtype=torch.int32).flip(0).repeat_interleave(N).view(T, N).contiguous()

def mk(ndir):
    dirs = []
    keep = []
    for i in range(ndir):
        m = MARN_cell(128, 128, 100, 100).cuda()
        P = dict(m.named_parameters()); G = {k: torch.zeros_like(v) for k, v in P.items()}
        out = torch.zeros(T * N, 512, device="cuda"); dout = torch.randn(T * N, 512, device="cuda")
        keep += [m, P, G, out, dout]
        dirs.append(dict(p=ops.cell_param_struct(lambda n, P=P: P[n].detach()), g=ops.cell_param_struct(lambda n, G=G: G[n]),
                         qmask=qmask, rev=(rev if i else None), out=out, dout=dout))
    ws = torch.zeros(ops.cell_workspace_bytes(T, N, 100, 128, ndir), device="cuda", dtype=torch.uint8)
    dx_l, dx_a = torch.zeros(T * N, 100, device="cuda"), torch.zeros(T * N, 100, device="cuda")
    desc = ops.make_cell_desc(T, N, 100, 128, x_l.view(T * N, 100), x_a.view(T * N, 100), dirs, 512, ws, dx_l=dx_l, dx_a=dx_a)
    return desc, keep, ws

def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)

for ndir in (1, 2):
    desc, keep, ws = mk(ndir)
    ops.marn_cell_fwd(desc); ops.marn_cell_bwd(desc); torch.cuda.synchronize()
    f_all = tm(lambda: ops.marn_cell_run(desc, ops.PHASE_FWD_PREP | ops.PHASE_SPEAKER_FWD | ops.PHASE_LSTHM_FWD))
    f_prep = tm(lambda: ops.marn_cell_run(desc, ops.PHASE_FWD_PREP))
    b_chain = tm(lambda: ops.marn_cell_run(desc, ops.PHASE_BWD_PREP | ops.PHASE_LSTHM_BWD))
    b_prep = tm(lambda: ops.marn_cell_run(desc, ops.PHASE_BWD_PREP))
    print(f"ndir={ndir}: fwd all {f_all:.0f} us (prep {f_prep:.0f}); bwd chain {b_chain:.0f} us (prep {b_prep:.0f}) -> "
          f"fwd {(f_all - f_prep) / T:.2f} us/step, bwd {(b_chain - b_prep) / T:.2f} us/step", flush=True)
    ops.marn_cell_status(desc)
