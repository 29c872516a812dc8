"""DialogueRNN persistent launches against the per-step launches at the edges of their size range (ad-hoc sweep; the committed cases
are tests/test_gpu_round3.py::test_drnn_persistent_matches_per_step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-framework-for-speaker-emotion-recognition_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import ref_cpu as O
from gpu_util import maxabs
from mser import ops, fault
from test_gpu_round3 import _bimodel_run
from test_gpu_model import _bimodel
cases = [(dict(D_m=64, D_g=512, D_p=512, D_e=512, D_h=32), 128, 4, False), (dict(D_m=20, D_g=8, D_p=8, D_e=8, D_h=4), 1, 3, True),
         (dict(D_m=48, D_g=100, D_p=60, D_e=30, D_h=20), 70, 33, True), (dict(D_m=33, D_g=13, D_p=511, D_e=7, D_h=5), 9, 2, False)]
for dims, B, L, train in cases:
    net = _bimodel(dims, 171, O, train)
    U, qmask, umask, label = O.bimodel_seeded_batch(B, L, D_m=dims["D_m"], seed=172 + B, ragged=True)
    res = {}
    for mode in (0, 1):
        ops.set_option(ops.MSER_OPT_DRNN_PERSISTENT, mode)
        net._rng = None
        res[mode] = _bimodel_run(net, U, qmask, umask, label)
    ops.set_option(ops.MSER_OPT_DRNN_PERSISTENT, 1)
    fault.check(torch.device("cuda:0"), "drnn_sizes")
    e_lp = maxabs(res[1][0], res[0][0])
    e_g = max(float(maxabs(res[1][3][n], res[0][3][n])) / max(1.0, float(res[0][3][n].abs().max())) for n in res[0][3])
    print(dims, "B", B, "L", L, "train", train, "-> max |d logp|", float(e_lp), "max rel grad diff", e_g)
    assert e_lp < 5e-5 and e_g < 5e-5
print("ok")
