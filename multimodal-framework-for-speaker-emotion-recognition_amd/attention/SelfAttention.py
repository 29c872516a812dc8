"""MI355X-native mirror of the reference's ``attention.SelfAttention.ScaledDotProductAttention``
(reference file attention:/SelfAttention.py:8-76): multi-head QK^T-softmax-V with biased projections, optional
multiplicative ``attention_weights`` and boolean ``attention_mask`` (True = masked -> -inf), output projection fc_o.
Linear weights are initialised N(0, 0.001^2) with zero bias like the reference (:35-47).
"""
import numpy as np
import torch
from torch import nn
from torch.nn import init

from mser import functional as F_
from mser import ops
from mser.autograd import ModuleFn, require_gpu
from mser.functional import Layout


class ScaledDotProductAttention(nn.Module):
    '''
    Scaled dot-product attention
    '''

    def __init__(self, d_model, d_k, d_v, h, dropout=.1):
        super(ScaledDotProductAttention, self).__init__()
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.dropout = nn.Dropout(dropout)
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    init.constant_(m.bias, 0)

    def forward(self, queries, keys, values, attention_mask=None, attention_weights=None):
        require_gpu(queries, keys, values)
        names = ["fc_q.weight", "fc_q.bias", "fc_k.weight", "fc_k.bias", "fc_v.weight", "fc_v.bias", "fc_o.weight", "fc_o.bias"]
        params = dict(self.named_parameters())
        h, dk, dv = self.h, self.d_k, self.d_v
        drop = self._last_drop = F_.module_site(self.dropout, queries.device, 5)       # :72 att = self.dropout(att)

        class Impl:
            @staticmethod
            def fwd(queries, keys, values, *pv):
                P = dict(zip(names, pv))
                b, nq, dm = queries.shape
                nk = keys.shape[1]
                xq = queries.contiguous().view(b * nq, dm)
                xk = keys.contiguous().view(b * nk, dm)
                xv = values.contiguous().view(b * nk, dm)
                dev = xq.device
                q, k, v = torch.empty(b * nq, h * dk, device=dev), torch.empty(b * nk, h * dk, device=dev), torch.empty(b * nk, h * dv, device=dev)
                ops.linear(xq, P["fc_q.weight"], q, bias=P["fc_q.bias"])
                ops.linear(xk, P["fc_k.weight"], k, bias=P["fc_k.bias"])
                ops.linear(xv, P["fc_v.weight"], v, bias=P["fc_v.bias"])
                mul = attention_weights.expand(b, h, nq, nk).contiguous().float() if attention_weights is not None else None
                msk = attention_mask.expand(b, h, nq, nk).to(torch.uint8).contiguous() if attention_mask is not None else None
                o = torch.empty(b * nq, h * dv, device=dev)
                lq, lk = Layout.batch_major(b, nq), Layout.batch_major(b, nk)
                Pm = F_.attn_core_fwd(q, k, v, o, lq, lk, h, dk, dv, 1.0 / float(np.sqrt(dk)), mul=mul, mask=msk, mask_on=1,
                                      fill=float("-inf"), drop=drop)
                Pm, Pd = Pm if drop is not None else (Pm, None)
                out = torch.empty(b * nq, dm, device=dev)
                ops.linear(o, P["fc_o.weight"], out, bias=P["fc_o.bias"])
                return out.view(b, nq, dm), (xq, xk, xv, q, k, v, o, Pm, Pd, mul, lq, lk, P)

            @staticmethod
            def bwd(saved, tensors, dout):
                xq, xk, xv, q, k, v, o, Pm, Pd, mul, lq, lk, P = saved
                G = {n: torch.zeros_like(params[n]) for n in names}
                d2 = dout.contiguous().view(xq.shape[0], -1)
                do = torch.empty_like(o)
                ops.matmul(d2, P["fc_o.weight"], do)
                ops.grad_weight(d2, o, G["fc_o.weight"])
                ops.colsum_acc(d2, G["fc_o.bias"])
                dq, dk_, dv_ = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                F_.attn_core_bwd(do, q, k, v, Pm, dq, dk_, dv_, lq, lk, h, dk, dv, 1.0 / float(np.sqrt(dk)), mul=mul, drop=drop, Pd=Pd)
                outs = []
                for g, x, wn, bn in ((dq, xq, "fc_q.weight", "fc_q.bias"), (dk_, xk, "fc_k.weight", "fc_k.bias"),
                                     (dv_, xv, "fc_v.weight", "fc_v.bias")):
                    dx = torch.empty_like(x)
                    ops.matmul(g, P[wn], dx)
                    ops.grad_weight(g, x, G[wn])
                    ops.colsum_acc(g, G[bn])
                    outs.append(dx)
                return (outs[0].view(tensors[0].shape), outs[1].view(tensors[1].shape), outs[2].view(tensors[2].shape),
                        *[G[n] for n in names])

        return ModuleFn.apply(Impl, queries, keys, values, *[params[n] for n in names])
