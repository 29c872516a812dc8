"""MI355X-native mirror of the reference's ``models.encoder`` (reference file model/encoder.py).

Same class names, constructor/forward signatures, parameter names and initialisers; the arithmetic runs in libmser's HIP
kernels (strided MFMA fp32 GEMMs, wave-per-row softmax / LayerNorm) with hand-written backward passes.  In train mode the
Dropout sites draw from libmser's counter-based generator (mser.functional.module_site; parity is defined at p = 0 / eval and,
in train mode, mask for mask: the sites of the last call are kept in ``_last_drop(s)``).  Inputs are batch-major [B, L, D] like the reference's.
"""
import torch
import torch.nn as nn

from mser import functional as F_
from mser.autograd import ModuleFn, require_gpu
from mser.functional import Layout


def _pget(mod, prefix=""):
    sd = dict(mod.named_parameters())
    return lambda n: sd[prefix + n].detach()


class _Grads:
    """zero-initialised gradient accumulators for a standalone module call"""

    def __init__(self, params):
        self.g = {n: torch.zeros_like(p) for n, p in params.items()}

    def __call__(self, n):
        return self.g[n]


def _mask_u8(mask, nb, nh, Lq, Lk):
    if mask is None:
        return None
    m = (mask != 0).to(torch.uint8)
    return m.expand(nb, nh, Lq, Lk).contiguous()


class ScaledDotProductAttention(nn.Module):
    """Reference model/encoder.py:63-86: softmax(q / temperature @ k^T [masked_fill(mask==0,-1e9)]) @ v; q,k,v [B,n,L,d]."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)

    def forward(self, q, k, v, mask=None):
        require_gpu(q, k, v)
        temperature = self.temperature
        drop = self._last_drop = F_.module_site(self.dropout, q.device, 0)

        class Impl:
            @staticmethod
            def fwd(q, k, v):
                B, n, Lq, d = q.shape
                Lk, dv = k.shape[2], v.shape[3]
                # heads become the leading "batch" level: treat [B*n] as nb with one head each
                q2, k2, v2 = (t.contiguous().view(-1, t.shape[-1]) for t in (q, k, v))
                out = torch.empty(B * n * Lq, dv, device=q.device)
                lq, lk = Layout.batch_major(B * n, Lq), Layout.batch_major(B * n, Lk)
                P = F_.attn_core_fwd(q2, k2, v2, out, lq, lk, 1, d, dv, 1.0 / temperature,
                                     mask=_mask_u8(mask, B, n, Lq, Lk), mask_on=0, fill=-1e9, drop=drop)
                P, Pd = P if drop is not None else (P, None)
                # the reference returns the attention AFTER its dropout (:83-86)
                return (out.view(B, n, Lq, dv), (P if Pd is None else Pd).view(B, n, Lq, Lk)), (q2, k2, v2, P, Pd, lq, lk, d, dv)

            @staticmethod
            def bwd(saved, tensors, dout, dattn):
                q2, k2, v2, P, Pd, lq, lk, d, dv = saved
                if dout is None:
                    return (None, None, None)
                dq, dk, dv_ = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
                F_.attn_core_bwd(dout.contiguous().view(-1, dv), q2, k2, v2, P, dq, dk, dv_, lq, lk, 1, d, dv, 1.0 / temperature,
                                 drop=drop, Pd=Pd)
                return tuple(g.view(t.shape) for g, t in zip((dq, dk, dv_), tensors))

        return ModuleFn.apply(Impl, q, k, v)


class MultiHeadAttention(nn.Module):
    """Reference model/encoder.py:7-60."""

    def __init__(self, n_head, d_model, d_model2, d_k, d_v, dropout=0.1):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model2, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model2, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.attention = ScaledDotProductAttention(temperature=d_k ** 0.5)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)

    def forward(self, q, k, v, mask=None):
        require_gpu(q, k, v)
        names = ["w_qs.weight", "w_ks.weight", "w_vs.weight", "fc.weight", "layer_norm.weight", "layer_norm.bias"]
        params = dict(self.named_parameters())
        nh, dk, dv = self.n_head, self.d_k, self.d_v
        same = (q is k) and (k is v)
        drops = self._last_drops = (F_.module_site(self.attention.dropout, q.device, 0), F_.module_site(self.dropout, q.device, 1))

        class Impl:
            @staticmethod
            def fwd(q, k, v, *pv):
                P = dict(zip(names, pv)).__getitem__
                B, Lq, D = q.shape
                Lk = k.shape[1]
                q2 = q.contiguous().view(B * Lq, D)
                k2 = q2 if same else k.contiguous().view(B * Lk, -1)
                v2 = q2 if same else v.contiguous().view(B * Lk, -1)
                m = _mask_u8(mask.unsqueeze(1) if mask is not None else None, B, nh, Lq, Lk)
                out, c = F_.mha_fwd(q2, k2, v2, P, Layout.batch_major(B, Lq), Layout.batch_major(B, Lk), nh, dk, dv, mask=m,
                                    drop_attn=drops[0], drop_fc=drops[1])
                return (out.view(B, Lq, D), c.P if c.Pd is None else c.Pd), (c, P)

            @staticmethod
            def bwd(saved, tensors, dout, dattn):
                c, P = saved
                G = _Grads({n: params[n] for n in names})
                dxq = torch.empty_like(c.xq)
                if same:
                    F_.mha_bwd(c, dout.contiguous().view(c.xq.shape), P, G, dxq, dxq, dxq, init_q=True)
                    dk_ = dv_ = None
                else:
                    dk_, dv_ = torch.zeros_like(c.xk), torch.zeros_like(c.xv)
                    F_.mha_bwd(c, dout.contiguous().view(c.xq.shape), P, G, dxq, dk_, dv_, init_q=True)
                    dk_, dv_ = dk_.view(tensors[1].shape), dv_.view(tensors[2].shape)
                return (dxq.view(tensors[0].shape), dk_, dv_, *[G(n) for n in names])

        if same:
            # one autograd input so the three gradient paths are summed by the kernel, not by autograd
            out, attn = ModuleFn.apply(Impl, q, q.detach(), q.detach(), *[params[n] for n in names])
        else:
            out, attn = ModuleFn.apply(Impl, q, k, v, *[params[n] for n in names])
        return out, attn


class PositionwiseFeedForward(nn.Module):
    """Reference model/encoder.py:89-113 (``fc`` is constructed and never used there; kept for state_dict parity)."""

    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_in, d_hid)
        self.w_2 = nn.Linear(d_hid, d_in)
        self.layer_norm = nn.LayerNorm(d_in, eps=1e-6)
        self.dropout = nn.Dropout(dropout)
        self.fc = nn.Linear(d_in, 100)

    def forward(self, x):
        require_gpu(x)
        names = ["w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias", "layer_norm.weight", "layer_norm.bias"]
        params = dict(self.named_parameters())
        drop = self._last_drop = F_.module_site(self.dropout, x.device, 2)

        class Impl:
            @staticmethod
            def fwd(x, *pv):
                P = dict(zip(names, pv)).__getitem__
                x2 = x.contiguous().view(-1, x.shape[-1])
                out, c = F_.ffn_fwd(x2, P, drop=drop)
                return out.view(x.shape), (c, P)

            @staticmethod
            def bwd(saved, tensors, dout):
                c, P = saved
                G = _Grads({n: params[n] for n in names})
                dx = F_.ffn_bwd(c, dout.contiguous().view(c.x.shape), P, G)
                return (dx.view(tensors[0].shape), *[G(n) for n in names])

        return ModuleFn.apply(Impl, x, *[params[n] for n in names])


class EncoderLayer(nn.Module):
    """Reference model/encoder.py:116-133.  forward(enc_input [B,L,D], slf_attn_mask=None) -> (enc_output, enc_slf_attn)."""

    _NAMES = ["slf_attn.w_qs.weight", "slf_attn.w_ks.weight", "slf_attn.w_vs.weight", "slf_attn.fc.weight",
              "slf_attn.layer_norm.weight", "slf_attn.layer_norm.bias", "pos_ffn.w_1.weight", "pos_ffn.w_1.bias",
              "pos_ffn.w_2.weight", "pos_ffn.w_2.bias", "pos_ffn.layer_norm.weight", "pos_ffn.layer_norm.bias"]

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super(EncoderLayer, self).__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward(self, enc_input, slf_attn_mask=None):
        require_gpu(enc_input)
        names = self._NAMES
        params = dict(self.named_parameters())
        nh, dk, dv = self.slf_attn.n_head, self.slf_attn.d_k, self.slf_attn.d_v
        dev = enc_input.device
        drops = self._last_drops = (F_.module_site(self.slf_attn.attention.dropout, dev, 0), F_.module_site(self.slf_attn.dropout, dev, 1),
                                    F_.module_site(self.pos_ffn.dropout, dev, 2))

        class Impl:
            @staticmethod
            def fwd(x, *pv):
                P = dict(zip(names, pv)).__getitem__
                B, L, D = x.shape
                x2 = x.contiguous().view(B * L, D)
                m = _mask_u8(slf_attn_mask.unsqueeze(1) if slf_attn_mask is not None else None, B, nh, L, L)
                out, c = F_.encoder_layer_fwd(x2, None, P, Layout.batch_major(B, L), nh, dk, dv, mask=m, drops=drops)
                return (out.view(B, L, D), F_.encoder_attention(c)), (c, P)

            @staticmethod
            def bwd(saved, tensors, dout, dattn):
                c, P = saved
                G = _Grads({n: params[n] for n in names})
                dx = F_.encoder_layer_bwd(c, dout.contiguous().view(-1, dout.shape[-1]), P, G)
                return (dx.view(tensors[0].shape), *[G(n) for n in names])

        return ModuleFn.apply(Impl, enc_input, *[params[n] for n in names])
