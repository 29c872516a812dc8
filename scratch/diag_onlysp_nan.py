"""Which gradients of MARN1_onlysp are non-finite (B=5, L=7, eval), and the fault word."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import ref_cpu as O
from models.lsthm_onlysp import MARN1_onlysp
from loss import MaskedLoss
from mser import fault, onlysp_fn
if os.environ.get("NOLINK"):
    onlysp_fn.LINK_GRU_BWD = False
    onlysp_fn.LINK_GRU_FWD = False
B, L, d_r = 5, 7, 768
P = O.seeded_params(seed=81, d_r=d_r, variant="onlysp")
net = MARN1_onlysp(6, d_r=d_r).cuda().eval()
sd = net.state_dict()
with torch.no_grad():
    for k, v in P.items():
        sd[k].copy_(v)
x, qmask, umask, label = O.seeded_batch(B, L, d_r=d_r, seed=83 + B, ragged=True)
for rep in range(3):
    net.zero_grad()
    lp, _, _ = net(x.cuda(), qmask.cuda(), umask.cuda())
    loss = MaskedLoss(torch.nn.NLLLoss)(lp, label.cuda().view(-1), umask.cuda())
    loss.backward()
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("rep", rep, "loss", float(loss), "bad grads:", bad[:12], len(bad))
    try:
        fault.check(torch.device("cuda:0"))
        print("fault word clean")
    except Exception as e:
        print("fault:", e)
