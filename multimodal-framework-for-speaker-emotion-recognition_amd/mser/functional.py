"""Host-side composition of the hot path out of libmser kernels: forward AND hand-written backward of every block
(encoder layer, sequence-level cross-modal attention, MARN cell, fusion head).  No torch arithmetic on the path:
torch only allocates buffers and provides streams.

Row layout: an activation is a 2-D [rows, D] matrix; ``Layout(nb, nl, sb, sl)`` says where sequence position l of
dialogue b lives: row = b*sb + l*sl (time-major [L,B,D]: sb=1, sl=B; batch-major [B,L,D]: sb=L, sl=1).

``P(name)`` returns a parameter tensor, ``G(name)`` the tensor its gradient is ACCUMULATED into (or None to skip).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch

from . import _lib as L_
from . import ops

Tensor = torch.Tensor
Getter = Callable[[str], Optional[Tensor]]


@dataclass
class Layout:
    nb: int
    nl: int
    sb: int
    sl: int

    @property
    def rows(self) -> int:
        return self.nb * self.nl

    @staticmethod
    def time_major(L: int, B: int) -> "Layout":
        return Layout(B, L, 1, B)

    @staticmethod
    def batch_major(B: int, L: int) -> "Layout":
        return Layout(B, L, L, 1)


def _empty(*shape, like: Tensor, dtype=torch.float32) -> Tensor:
    return torch.empty(*shape, device=like.device, dtype=dtype)


def _zeros(*shape, like: Tensor, dtype=torch.float32) -> Tensor:
    return torch.zeros(*shape, device=like.device, dtype=dtype)


# ====================================================================================================== dropout sites
# Site numbers of the MARN1_sps path (include/mser.h "Dropout"; the reference lines are the nn.Dropout calls they stand for).
SITE_ENC = 0        # + 3*call + {0: attention (encoder.py:83), 1: after fc (:54), 2: after w_2 (:106)}, call = 0..3 (text 1st/2nd, audio 1st/2nd)
SITE_XATTN = 12     # + {0: crossatt_l2a, 1: crossatt_a2l, 2: crossatt_l2a_1, 3: crossatt_a2l_1}   (lsthm_sps.py:98,:126)
SITE_FC = 16        # fc Dropout (:318)
SITE_OUT = 17       # nn_out Dropout (:323)
SITE_REC = 18       # + direction: dropout_rec on the cell outputs (:365,:374)
SITE_CELL = 20      # + 4*direction + {0: h_q0/h_q1 (:183,:188), 1: h_l/h_a (:211,:213), 2: rank-1 attention (:69)}


@dataclass
class DropSite:
    """One dropout site of one step: the device rng words, the site number and p.  ``None`` stands for the identity."""
    rng: Tensor
    site: int
    p: float

    def apply_(self, x: Tensor) -> Tensor:
        ops.dropout_apply_(x, self.rng, self.site, self.p)
        return x

    def scale(self, n: int, draw_bits: int = 32) -> Tensor:
        return ops.dropout_scale(n, self.rng, self.site, self.p, draw_bits=draw_bits)


SITE_MODULE = 64    # module mirrors used on their own (outside MARN1_sps): + a per-class offset; every draw advances the step word
MODULE_DROPOUT_SEED = 0x0D15EA5E
_MODULE_RNG = {}


def module_site(dropout: "torch.nn.Dropout", device, offset: int = 0) -> Optional[DropSite]:
    """Train-mode dropout site of a module mirror used on its own: one generator per device, advanced at every draw.  None when
    the module is in eval mode or p = 0 (``dropout.training`` follows the owning module's .train() / .eval())."""
    if not dropout.training or dropout.p <= 0:
        return None
    rng = _MODULE_RNG.get(device)
    if rng is None:
        rng = _MODULE_RNG[device] = torch.tensor([MODULE_DROPOUT_SEED & 0x7FFFFFFF, 0], dtype=torch.int32, device=device)
    ops.rng_advance_(rng)
    return DropSite(rng.clone(), SITE_MODULE + offset, float(dropout.p))


def zero_dropout(module):
    """Set p = 0 on every nn.Dropout of a module tree: the parity configuration (what tests/golden/make_golden.py does to the
    reference before it records trainer goldens).  Returns the module."""
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return module


# ====================================================================================================== attention core
def attn_core_fwd(q: Tensor, k: Tensor, v: Tensor, out: Tensor, lq: Layout, lk: Layout, nh: int, dk: int, dv: int,
                  scale: float, mul: Optional[Tensor] = None, mask: Optional[Tensor] = None, mask_on: int = 1,
                  fill: float = float("-inf"), drop: Optional[DropSite] = None):
    """P = softmax(scale * Q K^T [* mul, masked]) ; out = dropout(P) V.  q/k/v/out are 2-D row views [rows, nh*d].
    Returns P, or (P, dropout(P)) when a dropout site is given (the softmax backward needs the undropped P)."""
    nb, Lq, Lk = lq.nb, lq.nl, lk.nl
    ldq, ldk, ldv, ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    P = _empty(nb, nh, Lq, Lk, like=q)
    ops.gemm_raw(q, k, P, Lq, Lk, dk, lq.sl * ldq, 1, 1, lk.sl * ldk, Lk, batch=(nb, nh), sA=(lq.sb * ldq, dk),
                 sB=(lk.sb * ldk, dk), sC=(nh * Lq * Lk, Lq * Lk), alpha=scale)
    ops.softmax_rows_(P, nb * nh * Lq, Lk, Lk, mul=mul, mask=mask, mask_on=mask_on, fill=fill)
    Pd = drop.apply_(P.clone()) if drop is not None else P
    ops.gemm_raw(Pd, v, out, Lq, dv, Lk, Lk, 1, lk.sl * ldv, 1, lq.sl * ldo, batch=(nb, nh), sA=(nh * Lq * Lk, Lq * Lk),
                 sB=(lk.sb * ldv, dv), sC=(lq.sb * ldo, dv))
    return P if drop is None else (P, Pd)


def attn_core_bwd(dO: Tensor, q: Tensor, k: Tensor, v: Tensor, P: Tensor, dq: Tensor, dk_: Tensor, dv_: Tensor, lq: Layout,
                  lk: Layout, nh: int, dk: int, dv: int, scale: float, mul: Optional[Tensor] = None,
                  drop: Optional[DropSite] = None, Pd: Optional[Tensor] = None) -> None:
    """Writes dq, dk_, dv_ (2-D row views shaped like q, k, v).  With a dropout site: dV = Pd^T dO, dP = mask o (dO V^T)."""
    nb, Lq, Lk = lq.nb, lq.nl, lk.nl
    ldq, ldk, ldv, lddo = q.stride(0), k.stride(0), v.stride(0), dO.stride(0)
    PP = (nh * Lq * Lk, Lq * Lk)
    dP = torch.empty_like(P)
    ops.gemm_raw(dO, v, dP, Lq, Lk, dv, lq.sl * lddo, 1, 1, lk.sl * ldv, Lk, batch=(nb, nh), sA=(lq.sb * lddo, dv),
                 sB=(lk.sb * ldv, dv), sC=PP)
    ops.gemm_raw(P if drop is None else Pd, dO, dv_, Lk, dv, Lq, 1, Lk, lq.sl * lddo, 1, lk.sl * dv_.stride(0), batch=(nb, nh), sA=PP,
                 sB=(lq.sb * lddo, dv), sC=(lk.sb * dv_.stride(0), dv))
    if drop is not None:
        drop.apply_(dP)
    ops.softmax_bwd_rows_(P, dP, nb * nh * Lq, Lk, Lk, mul=mul)
    ops.gemm_raw(dP, k, dq, Lq, dk, Lk, Lk, 1, lk.sl * ldk, 1, lq.sl * dq.stride(0), batch=(nb, nh), sA=PP,
                 sB=(lk.sb * ldk, dk), sC=(lq.sb * dq.stride(0), dk), alpha=scale)
    ops.gemm_raw(dP, q, dk_, Lk, dk, Lq, 1, Lk, lq.sl * ldq, 1, lk.sl * dk_.stride(0), batch=(nb, nh), sA=PP,
                 sB=(lq.sb * ldq, dk), sC=(lk.sb * dk_.stride(0), dk), alpha=scale)


# ====================================================================================================== encoder layer
@dataclass
class MhaCtx:
    lq: Layout = None
    lk: Layout = None
    xq: Tensor = None
    xk: Tensor = None
    xv: Tensor = None
    q: Tensor = None
    k: Tensor = None
    v: Tensor = None
    qkv: Tensor = None
    P: Tensor = None
    Pd: Tensor = None
    O: Tensor = None
    y1: Tensor = None
    mean: Tensor = None
    rstd: Tensor = None
    drop_attn: Optional[DropSite] = None
    drop_fc: Optional[DropSite] = None
    nh: int = 0
    dk: int = 0
    dv: int = 0


def _fused_qkv(P: Getter, G: Optional[Getter] = None):
    """If w_qs, w_ks, w_vs sit back to back in one storage (the flat parameter buffer lays them out that way), return a
    [2*nq+nv, D] view over the three (and the matching gradient view): one N=960 projection GEMM instead of three."""
    wq, wk, wv = P("w_qs.weight"), P("w_ks.weight"), P("w_vs.weight")
    D = wq.shape[1]
    if not (wq.is_contiguous() and wk.is_contiguous() and wv.is_contiguous()):
        return None, None
    if wk.data_ptr() != wq.data_ptr() + 4 * wq.numel() or wv.data_ptr() != wk.data_ptr() + 4 * wk.numel():
        return None, None
    sq = wq.untyped_storage()
    if sq.data_ptr() != wk.untyped_storage().data_ptr() or sq.data_ptr() != wv.untyped_storage().data_ptr():
        return None, None           # merely adjacent allocations, not one storage
    rows = wq.shape[0] + wk.shape[0] + wv.shape[0]
    W = wq.as_strided((rows, D), (D, 1))
    gW = None
    if G is not None:
        gq, gk, gv = G("w_qs.weight"), G("w_ks.weight"), G("w_vs.weight")
        if gq is None or gk.data_ptr() != gq.data_ptr() + 4 * gq.numel() or gv.data_ptr() != gk.data_ptr() + 4 * gk.numel():
            return None, None
        if gq.untyped_storage().data_ptr() != gk.untyped_storage().data_ptr() or gq.untyped_storage().data_ptr() != gv.untyped_storage().data_ptr():
            return None, None
        gW = gq.as_strided((rows, D), (D, 1))
    return W, gW


def mha_fwd(xq: Tensor, xk: Tensor, xv: Tensor, P: Getter, lq: Layout, lk: Layout, nh: int, dk: int, dv: int,
            mask: Optional[Tensor] = None, out: Optional[Tensor] = None, drop_attn: Optional[DropSite] = None,
            drop_fc: Optional[DropSite] = None):
    """MultiHeadAttention.forward -- reference model/encoder.py:27-60: bias-free projections, softmax(q/sqrt(dk) k^T) v,
    fc, + residual(q input), LayerNorm(eps 1e-6).  Inputs are contiguous 2-D row matrices.  ``mask`` (uint8 [nb,nh,Lq,Lk],
    0 = masked) reproduces masked_fill(mask == 0, -1e9) (:75-77).  Returns (out, ctx); ctx.P is the attention [nb,nh,Lq,Lk]."""
    rows, D = xq.shape
    c = MhaCtx(lq=lq, lk=lk, xq=xq, xk=xk, xv=xv, nh=nh, dk=dk, dv=dv)
    nq, nv = nh * dk, nh * dv
    c.qkv = None
    if xq is xk and xk is xv:                       # self-attention: one [rows, 2nq+nv] buffer
        qkv = _empty(rows, 2 * nq + nv, like=xq)
        c.q, c.k, c.v = qkv[:, :nq], qkv[:, nq:2 * nq], qkv[:, 2 * nq:]
        Wqkv, _ = _fused_qkv(P)
        if Wqkv is not None:
            c.qkv = qkv
            ops.linear(xq, Wqkv, qkv)               # one N = 2nq+nv projection
    else:
        c.q, c.k, c.v = _empty(rows, nq, like=xq), _empty(xk.shape[0], nq, like=xq), _empty(xv.shape[0], nv, like=xq)
    if c.qkv is None:
        ops.linear(xq, P("w_qs.weight"), c.q)
        ops.linear(xk, P("w_ks.weight"), c.k)
        ops.linear(xv, P("w_vs.weight"), c.v)
    c.O = _empty(rows, nv, like=xq)
    c.drop_attn, c.drop_fc = drop_attn, drop_fc
    c.P = attn_core_fwd(c.q, c.k, c.v, c.O, lq, lk, nh, dk, dv, 1.0 / (dk ** 0.5), mask=mask, mask_on=0, fill=-1e9, drop=drop_attn)
    if drop_attn is not None:
        c.P, c.Pd = c.P
    t = _empty(rows, D, like=xq)
    ops.linear(c.O, P("fc.weight"), t)
    if drop_fc is not None:
        drop_fc.apply_(t)                                  # :54 dropout(fc(.)) before the residual
    c.y1 = _empty(rows, D, like=xq)
    if out is None:
        out = _empty(rows, D, like=xq)
    c.mean, c.rstd = _empty(rows, like=xq), _empty(rows, like=xq)
    ops.add_layernorm_fwd(t, xq, P("layer_norm.weight"), P("layer_norm.bias"), out, c.y1, c.mean, c.rstd, 1e-6)
    return out, c


def mha_bwd(c: MhaCtx, dout: Tensor, P: Getter, G: Getter, dxq: Tensor, dxk: Tensor, dxv: Tensor, init_q: bool) -> None:
    """Accumulates into dxq/dxk/dxv (which may alias).  If ``init_q`` the residual gradient INITIALISES dxq (no prior read)."""
    rows, D = dout.shape
    nh, dk, dv = c.nh, c.dk, c.dv
    dy1 = _empty(rows, D, like=dout)          # never modified afterwards: the deferred fc weight gradient reads it
    ops.layernorm_bwd(dout, c.y1, c.mean, c.rstd, P("layer_norm.weight"), dy1, G("layer_norm.weight"), G("layer_norm.bias"))
    dO = _empty(rows, nh * dv, like=dout)
    dt = dy1 if c.drop_fc is None else c.drop_fc.apply_(dy1.clone())      # gradient at the fc output (dy1 itself: the residual's)
    ops.matmul(dt, P("fc.weight"), dO)
    ops.grad_weight(dt, c.O, G("fc.weight"))
    dk_kw = dict(drop=c.drop_attn, Pd=c.Pd)
    Wqkv, gWqkv = _fused_qkv(P, G) if (c.qkv is not None and dxq is dxk and dxk is dxv and init_q) else (None, None)
    if Wqkv is not None:
        nq = nh * dk
        dqkv = torch.empty_like(c.qkv)
        attn_core_bwd(dO, c.q, c.k, c.v, c.P, dqkv[:, :nq], dqkv[:, nq:2 * nq], dqkv[:, 2 * nq:], c.lq, c.lk, nh, dk, dv, 1.0 / (dk ** 0.5),
                      **dk_kw)
        ops.matmul(dqkv, Wqkv, dxq, R1=dy1)            # residual gradient + [dq|dk|dv] @ [Wq;Wk;Wv]  (one K = 2nq+nv GEMM)
        ops.grad_weight(dqkv, c.xq, gWqkv)
        return
    dq, dk_, dv_ = torch.empty_like(c.q), torch.empty_like(c.k), torch.empty_like(c.v)
    attn_core_bwd(dO, c.q, c.k, c.v, c.P, dq, dk_, dv_, c.lq, c.lk, nh, dk, dv, 1.0 / (dk ** 0.5), **dk_kw)
    first = True
    for nm, dg, xin, dx in (("w_qs.weight", dq, c.xq, dxq), ("w_ks.weight", dk_, c.xk, dxk), ("w_vs.weight", dv_, c.xv, dxv)):
        if dx is not None:
            if first and init_q:
                ops.matmul(dg, P(nm), dx, R1=dy1)          # dxq = residual gradient + dq Wq (written, not accumulated)
            else:
                if first:
                    ops.add_rows(dx, dx, dy1)
                ops.matmul(dg, P(nm), dx, accum=True)
            first = False
        ops.grad_weight(dg, xin, G(nm))


@dataclass
class FfnCtx:
    x: Tensor = None
    hdn: Tensor = None
    y2: Tensor = None
    mean: Tensor = None
    rstd: Tensor = None
    drop: Optional[DropSite] = None


def ffn_fwd(x: Tensor, P: Getter, out: Optional[Tensor] = None, drop: Optional[DropSite] = None):
    """PositionwiseFeedForward.forward -- reference model/encoder.py:101-113 (w_2(relu(w_1 x)) + x, LayerNorm; ``fc`` unused)."""
    rows, D = x.shape
    c = FfnCtx(x=x, drop=drop)
    c.hdn = _empty(rows, P("w_1.weight").shape[0], like=x)
    ops.linear(x, P("w_1.weight"), c.hdn, bias=P("w_1.bias"), relu=True)
    t = _empty(rows, D, like=x)
    ops.linear(c.hdn, P("w_2.weight"), t, bias=P("w_2.bias"))
    if drop is not None:
        drop.apply_(t)                                     # :106 dropout(w_2(.)) before the residual
    c.y2 = _empty(rows, D, like=x)
    if out is None:
        out = _empty(rows, D, like=x)
    c.mean, c.rstd = _empty(rows, like=x), _empty(rows, like=x)
    ops.add_layernorm_fwd(t, x, P("layer_norm.weight"), P("layer_norm.bias"), out, c.y2, c.mean, c.rstd, 1e-6)
    return out, c


def ffn_bwd(c: FfnCtx, dout: Tensor, P: Getter, G: Getter) -> Tensor:
    rows, D = dout.shape
    dy2 = _empty(rows, D, like=dout)
    ops.layernorm_bwd(dout, c.y2, c.mean, c.rstd, P("layer_norm.weight"), dy2, G("layer_norm.weight"), G("layer_norm.bias"))
    dh = _empty(rows, c.hdn.shape[1], like=dout)
    dt = dy2 if c.drop is None else c.drop.apply_(dy2.clone())           # gradient at the w_2 output (dy2 itself: the residual's)
    ops.matmul(dt, P("w_2.weight"), dh)
    ops.grad_weight(dt, c.hdn, G("w_2.weight"))
    ops.colsum_acc(dt, G("w_2.bias"))
    ops.relu_bwd_(dh, c.hdn)
    dx = _empty(rows, D, like=dout)
    ops.matmul(dh, P("w_1.weight"), dx, R1=dy2)                       # residual + FFN input path (dy2 stays intact for its wgrads)
    ops.grad_weight(dh, c.x, G("w_1.weight"))
    ops.colsum_acc(dh, G("w_1.bias"))
    return dx


def _sub(P: Getter, prefix: str) -> Getter:
    return lambda n: P(prefix + n)


# The fused EncoderLayer (csrc/encoder.hip: 3 launches forward, 3 on the backward's activation chain) is used whenever the
# shape fits its LDS tiles; FUSED_ENCODER = False forces the composed path (generic GEMM + row kernels), which stays as the
# general fallback (L > 128, other head widths) and as the cross-check in tests/test_gpu_ops.py.
FUSED_ENCODER = True

_ENC_PARAMS = (("w_qs", "slf_attn.w_qs.weight"), ("w_ks", "slf_attn.w_ks.weight"), ("w_vs", "slf_attn.w_vs.weight"),
               ("fc", "slf_attn.fc.weight"), ("ln1_g", "slf_attn.layer_norm.weight"), ("ln1_b", "slf_attn.layer_norm.bias"),
               ("w1", "pos_ffn.w_1.weight"), ("b1", "pos_ffn.w_1.bias"), ("w2", "pos_ffn.w_2.weight"), ("b2", "pos_ffn.w_2.bias"),
               ("ln2_g", "pos_ffn.layer_norm.weight"), ("ln2_b", "pos_ffn.layer_norm.bias"))


@dataclass
class EncFusedCtx:
    desc: object = None
    keep: tuple = ()
    P: Tensor = None          # attention [nb, nh, L, L] (the module's second return value)
    rows: int = 0
    D: int = 0


def _encoder_desc(e0: Tensor, P: Getter, lay: Layout, nh: int, dk: int, dv: int, mask: Optional[Tensor]):
    from . import _lib as L_
    d = L_.EncoderDesc()
    d.nb, d.nl, d.sb, d.sl = lay.nb, lay.nl, lay.sb, lay.sl
    d.D, d.nh, d.dk, d.dv = e0.shape[1], nh, dk, dv
    d.dff = P("pos_ffn.w_1.weight").shape[0]
    d.eps = 1e-6
    d.x = e0.data_ptr()
    d.mask = mask.data_ptr() if mask is not None else None
    for f, n in _ENC_PARAMS:
        t = P(n)
        if not t.is_contiguous():
            return None
        setattr(d, f, t.data_ptr())
    return d


def encoder_attention(c) -> Tensor:
    """The attention tensor [nb, nh, L, L] of an encoder_layer_fwd context (either path); with dropout the dropped one, which is
    what the reference returns (model/encoder.py:83-86)."""
    if isinstance(c, EncFusedCtx):
        return c.P
    return c[0].Pd if c[0].Pd is not None else c[0].P


def encoder_layer_fwd(x: Tensor, x2: Optional[Tensor], P: Getter, lay: Layout, nh: int, dk: int, dv: int,
                      mask: Optional[Tensor] = None, out: Optional[Tensor] = None, drops=None, need_attn: bool = True):
    """EncoderLayer.forward on input (x + x2) -- reference model/encoder.py:130-133; x2 carries the residual of the model's
    second pass (model/lsthm_sps.py:357-358).  Returns (out [rows,D], ctx).  ``drops`` = (attention, fc, ffn) DropSites (each may
    be None) with consecutive site numbers and one rng.  The fused kernels draw them in place but keep only the plain softmax, so a
    caller that returns the (dropped) attention tensor like the reference (``need_attn``) gets the composed path instead."""
    rows, D = x.shape
    if x2 is None and x.stride(0) == D:
        e0 = x
    else:
        e0 = _empty(rows, D, like=x)
        ops.add_rows(e0, x, x2)
    drops = drops if (drops is not None and any(s_ is not None for s_ in drops)) else None
    fused_drop = None
    if drops is not None and not need_attn:
        live = [s_ for s_ in drops if s_ is not None]
        base = live[0].site - drops.index(live[0])
        if all(s_ is None or (s_.rng is live[0].rng and s_.site == base + i) for i, s_ in enumerate(drops)):
            fused_drop = (live[0].rng, base, tuple(s_.p if s_ is not None else 0.0 for s_ in drops))
    if FUSED_ENCODER and (drops is None or fused_drop is not None):
        d = _encoder_desc(e0, P, lay, nh, dk, dv, mask)
        if d is not None and fused_drop is not None:
            d.rng, d.drop_site = fused_drop[0].data_ptr(), fused_drop[1]
            d.p_attn, d.p_fc, d.p_ffn = fused_drop[2]
        if d is not None and ops.encoder_layer_supported(d):
            nq, F = nh * dk, d.dff
            if out is None:
                out = _empty(rows, D, like=x)
            qkv = _empty(rows, 2 * nq + nh * dv, like=x)
            Pm = _empty(lay.nb, nh, lay.nl, lay.nl, like=x)
            O = _empty(rows, nh * dv, like=x)
            y1, e1, y2 = _empty(rows, D, like=x), _empty(rows, D, like=x), _empty(rows, D, like=x)
            hdn = _empty(rows, F, like=x)
            st = _empty(4, rows, like=x)
            d.qkv, d.P, d.O, d.y1, d.e1, d.hdn, d.y2, d.out = (t.data_ptr() for t in (qkv, Pm, O, y1, e1, hdn, y2, out))
            d.mean1, d.rstd1, d.mean2, d.rstd2 = (st[i].data_ptr() for i in range(4))
            ops.encoder_layer_fwd(d)
            return out, EncFusedCtx(desc=d, keep=(e0, mask, qkv, Pm, O, y1, e1, hdn, y2, st, out), P=Pm, rows=rows, D=D)
    da, dfc, dffn = drops if drops is not None else (None, None, None)
    e1, cm = mha_fwd(e0, e0, e0, _sub(P, "slf_attn."), lay, lay, nh, dk, dv, mask=mask, drop_attn=da, drop_fc=dfc)
    out, cf = ffn_fwd(e1, _sub(P, "pos_ffn."), out=out, drop=dffn)
    return out, (cm, cf)


def encoder_layer_bwd(c, dout: Tensor, P: Getter, G: Getter) -> Tensor:
    """Returns d(x + x2) [rows, D]; parameter gradients are accumulated into G(name)."""
    if isinstance(c, EncFusedCtx):
        d, rows, D = c.desc, c.rows, c.D
        nq3, F = 3 * d.nh * d.dk, d.dff
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dx = _empty(rows, D, like=dout)
        dy2, dy1 = _empty(rows, D, like=dout), _empty(rows, D, like=dout)
        dh, dO, dqkv = _empty(rows, F, like=dout), _empty(rows, d.nh * d.dv, like=dout), _empty(rows, nq3, like=dout)
        d.dout, d.dx, d.dy2, d.dy1, d.dh, d.dO, d.dqkv = (t.data_ptr() for t in (dout, dx, dy2, dy1, dh, dO, dqkv))
        dt1 = None
        if d.rng and (d.p_fc > 0 or d.p_ffn > 0):
            dt1 = _empty(rows, D, like=dout)
            d.dt1 = dt1.data_ptr()
        for f, n in _ENC_PARAMS:
            g = G(n)
            if g is None or not g.is_contiguous():
                raise RuntimeError(f"encoder_layer_bwd: gradient buffer for {n} missing or not contiguous")
            setattr(d, "g_" + f, g.data_ptr())
        ops.encoder_layer_bwd(d, ops.ENC_BWD_ACT)
        ops.encoder_layer_bwd(d, ops.ENC_BWD_WGRAD, deferred=(dout, dx, dy2, dy1, dh, dO, dqkv, dt1) + tuple(c.keep))
        return dx
    cm, cf = c
    de1 = ffn_bwd(cf, dout, _sub(P, "pos_ffn."), _sub(G, "pos_ffn."))
    de0 = torch.empty_like(de1)
    mha_bwd(cm, de1, _sub(P, "slf_attn."), _sub(G, "slf_attn."), de0, de0, de0, init_q=True)
    return de0


# ====================================================================================================== CrossAttention2/3
@dataclass
class XAttnCtx:
    x1: Tensor = None
    x2: Tensor = None
    a1: Optional[Tensor] = None
    a2: Optional[Tensor] = None
    Q: Tensor = None
    KV: Tensor = None
    P: Tensor = None
    drop: Optional[DropSite] = None
    Pd: Tensor = None
    l1: Layout = None
    l2: Layout = None
    heads: int = 1
    dk: int = 0
    dv: int = 0
    fused: object = None       # mser_xattn_desc of the fused core launch (None: composed path)
    stats: Tensor = None       # [nb, heads, Lq, 2] softmax row statistics saved by the fused forward
    out: Tensor = None


FUSED_XATTN = True          # the fused QK^T-softmax-V launch (csrc/xattn.hip) where the shape allows; False: GEMM + softmax rows + GEMM


def _xattn_desc(c: "XAttnCtx", out: Tensor, Dk: int):
    d = L_.XAttnDesc()
    d.nb, d.nh, d.Lq, d.Lk, d.dk = c.l1.nb, c.heads, c.l1.nl, c.l2.nl, c.dk
    K_, V_ = c.KV[:, :Dk], c.KV[:, Dk:]
    d.q, d.ldq, d.k, d.ldk, d.v, d.ldv = c.Q.data_ptr(), c.Q.stride(0), K_.data_ptr(), K_.stride(0), V_.data_ptr(), V_.stride(0)
    d.sbq, d.slq, d.sbk, d.slk = c.l1.sb, c.l1.sl, c.l2.sb, c.l2.sl
    d.o, d.ldo = out.data_ptr(), out.stride(0)
    d.scale = 1.0 / (c.dk ** 0.5)
    if c.drop is not None:
        d.rng, d.site, d.p = c.drop.rng.data_ptr(), c.drop.site, float(c.drop.p)
    return d


def xattn_fwd(x1: Tensor, a1: Optional[Tensor], x2: Tensor, a2: Optional[Tensor], Wq: Tensor, Wk: Tensor, Wv: Tensor,
              l1: Layout, l2: Layout, out: Tensor, heads: int = 1, drop: Optional[DropSite] = None):
    """CrossAttention2/3.forward(a1*x1, a2*x2) -- reference model/lsthm_sps.py:88-101 / :116-129 with the learnable scalars of
    :377-383 folded into the projection GEMMs.  x1 [rows1,D1], x2 [rows2,D2]; out [rows1,Dv] (any leading dimension).  The core
    softmax(Q K^T / sqrt(dk)) V is ONE fused launch (mser_xattn_seq_fwd: no [B, L, L] tensor in HBM) when the shape fits it."""
    c = XAttnCtx(x1=x1, x2=x2, a1=a1, a2=a2, l1=l1, l2=l2, heads=heads, drop=drop)
    Dk, Dv = Wq.shape[1], Wv.shape[1]
    c.dk, c.dv = Dk // heads, Dv // heads
    c.Q = _empty(x1.shape[0], Dk, like=x1)
    c.KV = _empty(x2.shape[0], Dk + Dv, like=x1)
    # the three projections are independent: one grouped launch (they sit in front of the recurrent chains' launch; three nodes of
    # ~12 us each kept the branch running into the chains' first steps, where its traffic slows their hand-offs)
    ops.matmul_group([(x1, Wq, c.Q, a1), (x2, Wk, c.KV[:, :Dk], a2), (x2, Wv, c.KV[:, Dk:], a2)])
    if FUSED_XATTN and c.dk == c.dv:
        d = _xattn_desc(c, out, Dk)
        if ops.xattn_seq_supported(d):
            c.stats = _empty(l1.nb, heads, l1.nl, 2, like=x1)
            d.stats = c.stats.data_ptr()
            ops.xattn_seq_fwd(d)
            c.fused, c.out = d, out
            return c
    c.P = attn_core_fwd(c.Q, c.KV[:, :Dk], c.KV[:, Dk:], out, l1, l2, heads, c.dk, c.dv, 1.0 / (c.dk ** 0.5), drop=drop)
    if drop is not None:
        c.P, c.Pd = c.P                                     # :98 / :126 dropout(softmax(.)) before .V
    return c


def xattn_bwd(c: XAttnCtx, dout: Tensor, Wq: Tensor, Wk: Tensor, Wv: Tensor, gWq: Tensor, gWk: Tensor, gWv: Tensor,
              dx1: Tensor, dx2: Tensor, ga1: Optional[Tensor], ga2: Optional[Tensor]) -> None:
    """Accumulates into dx1, dx2 (grads of the UNSCALED inputs), the weight grads and the scalar grads."""
    Dk = Wq.shape[1]
    dQ = torch.empty_like(c.Q)
    if c.fused is not None:
        # dK / dV meet contributions from every 32-query tile: each tile stores into its own slab, the launch's second kernel adds them
        # in a fixed order (deterministic; round 2 used float atomics into a zeroed buffer)
        nslab = (c.l1.nl + 31) // 32
        slabs = torch.empty((nslab,) + tuple(c.KV.shape), device=c.KV.device, dtype=c.KV.dtype)
        dKV = slabs[0]
        dO = dout if (dout.stride(1) == 1 and dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0) else dout.contiguous()
        d = c.fused
        d.dO, d.lddo = dO.data_ptr(), dO.stride(0)
        d.dq, d.lddq = dQ.data_ptr(), dQ.stride(0)
        d.dk_, d.lddk = dKV.data_ptr(), dKV.stride(0)
        d.dv, d.lddv = dKV[:, Dk:].data_ptr(), dKV.stride(0)
        d.part_stride = slabs.stride(0) if nslab > 1 else 0
        if nslab == 1:
            dKV.zero_()
        ops.xattn_seq_bwd(d)
        c._keep_bwd = dO
    else:
        dKV = torch.empty_like(c.KV)
        attn_core_bwd(dout, c.Q, c.KV[:, :Dk], c.KV[:, Dk:], c.P, dQ, dKV[:, :Dk], dKV[:, Dk:], c.l1, c.l2, c.heads, c.dk, c.dv,
                      1.0 / (c.dk ** 0.5), drop=c.drop, Pd=c.Pd)
    ops.grad_weight(dQ, c.x1, gWq, transposed=True, alpha_dev=c.a1)
    ops.grad_weight(dKV[:, :Dk], c.x2, gWk, transposed=True, alpha_dev=c.a2)
    ops.grad_weight(dKV[:, Dk:], c.x2, gWv, transposed=True, alpha_dev=c.a2)
    t1 = _empty(c.x1.shape[0], c.x1.shape[1], like=dout)
    ops.matmul_nt(dQ, Wq, t1)                                         # grad wrt (a1 * x1)
    ops.scale_acc_dot(dx1, t1, c.x1, c.a1, ga1)
    t2 = _empty(c.x2.shape[0], c.x2.shape[1], like=dout)
    ops.matmul_nt(dKV[:, :Dk], Wk, t2)
    ops.matmul_nt(dKV[:, Dk:], Wv, t2, accum=True)
    ops.scale_acc_dot(dx2, t2, c.x2, c.a2, ga2)
