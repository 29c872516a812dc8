"""Stress of the counter-linked GRU / LSTHM launches of MARN1_onlysp (scratch): for several batch shapes, the linked schedule must
reproduce the sequential schedule's log-probs bit for bit, every time, and never raise the BPTT status flag."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd")]
import torch
from oracle import ref_cpu as O
from models.lsthm_onlysp import MARN1_onlysp
from loss import MaskedLoss
import mser.onlysp_fn as ofn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d_r = 768
net = MARN1_onlysp(6, d_r=d_r).cuda().eval()
sd = net.state_dict()
with torch.no_grad():
    for k, v in O.seeded_params(seed=95, d_r=d_r, variant="onlysp").items():
        sd[k].copy_(v)
bad = 0
for (B, L) in ((32, 128), (48, 40), (17, 64), (64, 32), (32, 16)):
    x, qmask, umask, label = (t.cuda() for t in O.seeded_batch(B, L, d_r=d_r, seed=100 + B, ragged=True))
    def run(linked):
        ofn.LINK_GRU_FWD = ofn.LINK_GRU_BWD = linked
        net.zero_grad(set_to_none=True)
        lp, _, _ = net(x, qmask, umask)
        MaskedLoss(torch.nn.NLLLoss)(lp, label.view(-1), umask).backward()
        torch.cuda.synchronize()
        g = net.marn_cell_b.gru_s.weight_ih.grad.detach().clone()
        return lp.detach().clone(), g
    lp0, g0 = run(False)
    mism = 0
    gmax = 0.0
    for _ in range(reps):
        lp1, g1 = run(True)
        mism += int(not torch.equal(lp1, lp0))
        gmax = max(gmax, float((g1 - g0).abs().max()) / max(1e-9, float(g0.abs().max())))
    net.check_links()          # raises if the sticky fault word is set (mser.fault)
    print(f"B={B} L={L}: {mism} / {reps} forward mismatches, worst relative gradient difference {gmax:.2e}", flush=True)
    bad += mism + int(gmax > 1e-5)
ofn.LINK_GRU_FWD = ofn.LINK_GRU_BWD = True
print("stress ok" if bad == 0 else f"stress FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
