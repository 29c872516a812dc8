"""Helpers shared by the GPU parity tests (imported only from tests marked ``gpu``)."""
import numpy as np
import torch

DEV = "cuda:0"


def dev(t):
    return t.to(DEV) if isinstance(t, torch.Tensor) else torch.tensor(np.asarray(t)).to(DEV)


def maxabs(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.tensor(np.asarray(b)).double()
    return float((a - b).abs().max())


def load_params(model, P):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in P.items():
            assert tuple(sd[k].shape) == tuple(v.shape), k
            sd[k].copy_(v)
