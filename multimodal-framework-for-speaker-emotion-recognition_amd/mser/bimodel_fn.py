"""Forward / backward of the DialogueRNN ``BiModel`` (reference model/DialogueRNN.py:201-277 with att2=True, as
model_trainer.py:35-47 / model_trainer_d.py:23-33 construct it: listener_state=True, context_attention='general'; SURVEY.md 8(f)
row f2, BASELINE configs[3]) as explicit kernel sequences:

* both DialogueRNNs (forward and reversed direction) run in ``mser_drnn_fwd/bwd`` (csrc/dialogue.hip): hoisted input GEMMs, a host
  loop of direction-batched fp32-MFMA GEMMs + gate epilogues + the history-attention kernel per step, weight gradients as large
  GEMMs after the BPTT;
* the head -- 'general2' MatchingAttention of every position over all positions of cat[e_f, e_b] (:256-262, :61-68,:75), linear +
  ReLU (:264), smax_fc + log_softmax (:269) -- is composed from the generic GEMM (one batch entry per dialogue for the score and
  pooling products) and two row kernels.

Dropout sites (train mode; identity when ``drop`` is None): SITE_DRNN + 4*direction + {0: g, 1: qs, 2: ql, 3: e} inside the cells,
SITE_DRNN_REC + direction on the emotion states (:245,:252), SITE_DRNN_HID on the hidden layer (:268).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import functional as F_
from . import ops
from .model_fn import Getter, _sub

Tensor = torch.Tensor

SITE_DRNN = 40
SITE_DRNN_REC = 48
SITE_DRNN_HID = 50


@dataclass
class BiDims:
    D_m: int = 712
    D_g: int = 500
    D_p: int = 500
    D_e: int = 300
    D_h: int = 300
    n_classes: int = 6


@dataclass
class BiDrop:
    rng: Tensor = None
    p_cell: float = 0.1        # DialogueRNNCell.dropout (:98) = BiModel's dropout_rec ctor argument (:216-219)
    p_rec: float = 0.25        # BiModel.dropout_rec = Dropout(dropout + 0.15) (:215)
    p_hid: float = 0.1         # BiModel.dropout (:214)

    def site(self, sid: int, p: float):
        return F_.DropSite(self.rng, sid, float(p)) if p > 0 else None

    def any(self) -> bool:
        return self.p_cell > 0 or self.p_rec > 0 or self.p_hid > 0


@dataclass
class BiCtx:
    dims: BiDims = None
    L: int = 0
    B: int = 0
    U2: Tensor = None
    qmask: Tensor = None
    umask: Tensor = None
    rev: Tensor = None
    ws: Tensor = None
    params: tuple = None
    E: Tensor = None
    X_: Tensor = None
    S0: Tensor = None
    alpha: Tensor = None
    att: Tensor = None
    hid: Tensor = None
    lp: Tensor = None
    desc: object = None
    drop: BiDrop = None
    cell_drop: tuple = None


def _scores(Xq: Tensor, E: Tensor, out: Tensor, Ln: int, B: int, W: int) -> None:
    """out[b, t, s] = <Xq[t*B+b], E[s*B+b]>   (one batch entry per dialogue)"""
    ops.gemm_raw(Xq, E, out, Ln, Ln, W, B * W, 1, 1, B * W, Ln, (B, 1), (W, 0), (W, 0), (Ln * Ln, 0))


def _pool(A: Tensor, E: Tensor, out: Tensor, Ln: int, B: int, W: int, accum: bool = False, transposed: bool = False) -> None:
    """out[t*B+b] (+)= sum_s A[b, t, s] E[s*B+b]  (transposed: sum_s A[b, s, t] E[s*B+b])"""
    sAm, sAk = (1, Ln) if transposed else (Ln, 1)
    ops.gemm_raw(A, E, out, Ln, W, Ln, sAm, sAk, B * W, 1, B * W, (B, 1), (Ln * Ln, 0), (W, 0), (W, 0), accum=accum)


def bimodel_forward(P: Getter, U: Tensor, qmask: Tensor, umask: Tensor, dims: BiDims, drop: Optional[BiDrop] = None,
                    need_alpha: bool = True):
    """U [L,B,D_m], qmask [L,B,2], umask [B,L] -> (log_prob [L,B,C], alpha [B,L,L] (row (b, t) = alpha_t[b, :]), ctx)."""
    if drop is not None and not drop.any():
        drop = None
    Ln, B, Dm = U.shape
    d = dims
    if Dm != d.D_m:
        raise RuntimeError(f"BiModel: U has {Dm} features, the model was built for D_m = {d.D_m}")
    for t, nm in ((U, "U"), (qmask, "qmask"), (umask, "umask")):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError(f"{nm} must be a float32 GPU tensor (got {t.dtype} on {t.device})")
    U, qmask, umask = U.contiguous(), qmask.contiguous(), umask.contiguous()
    dev = U.device
    N, W = Ln * B, 2 * d.D_e
    c = BiCtx(dims=d, L=Ln, B=B, qmask=qmask, umask=umask, drop=drop)
    c.U2 = U.view(N, Dm)
    lens = torch.empty(B, device=dev, dtype=torch.int32)
    c.rev = torch.empty(Ln, B, device=dev, dtype=torch.int32)
    ops.build_reverse_index(umask, lens, c.rev)
    c.E = torch.zeros(N, W, device=dev)                         # rows beyond a dialogue's length stay zero in the reversed half
    c.ws = torch.empty(ops.drnn_workspace_bytes(Ln, B, d.D_m, d.D_g, d.D_p, d.D_e), device=dev, dtype=torch.uint8)
    c.params = (ops.drnn_param_struct(_sub(P, "dialog_rnn_f.dialogue_cell.")), ops.drnn_param_struct(_sub(P, "dialog_rnn_r.dialogue_cell.")))
    if drop is not None and drop.p_cell > 0:
        c.cell_drop = (drop.rng, [SITE_DRNN, SITE_DRNN + 4], drop.p_cell)
    c.desc = ops.make_drnn_desc(Ln, B, (d.D_m, d.D_g, d.D_p, d.D_e), c.U2, qmask, c.rev, c.params, c.E, c.ws, drop=c.cell_drop)
    ops.drnn_fwd(c.desc)
    if drop is not None and drop.p_rec > 0:
        for i in range(2):
            drop.site(SITE_DRNN_REC + i, drop.p_rec).apply_(c.E[:, d.D_e * i:d.D_e * (i + 1)])
    # ---- 'general2' matching attention of every position over all positions (:256-262)
    c.X_ = torch.empty(N, W, device=dev)
    ops.linear(c.E, P("matchatt.transform.weight"), c.X_, bias=P("matchatt.transform.bias"))
    c.S0 = torch.empty(B, Ln, Ln, device=dev)
    _scores(c.X_, c.E, c.S0, Ln, B, W)
    c.alpha = torch.empty(B, Ln, Ln, device=dev)
    ops.general2_rows_fwd(c.S0, c.alpha, umask, B * Ln, Ln, Ln)
    c.att = torch.empty(N, W, device=dev)
    _pool(c.alpha, c.E, c.att, Ln, B, W)
    # ---- linear + ReLU (+ dropout), smax_fc, log_softmax (:264-269)
    c.hid = torch.empty(N, 2 * d.D_h, device=dev)
    ops.linear(c.att, P("linear.weight"), c.hid, bias=P("linear.bias"), relu=True)
    if drop is not None and drop.p_hid > 0:
        drop.site(SITE_DRNN_HID, drop.p_hid).apply_(c.hid)
    y = torch.empty(N, d.n_classes, device=dev)
    ops.linear(c.hid, P("smax_fc.weight"), y, bias=P("smax_fc.bias"))
    c.lp = torch.empty(N, d.n_classes, device=dev)
    ops.logsoftmax_tb_fwd(y, c.lp, 1, N)                        # (L = 1: rows keep their time-major order)
    return c.lp.view(Ln, B, d.n_classes), c.alpha, c


def bimodel_alpha_dir(c: BiCtx, direction: int) -> Tensor:
    """alpha_f / alpha_b as one [T, B, T] tensor (row (t, b) valid in its first t entries), a copy of the workspace array."""
    Ln, B = c.L, c.B
    off = ops.drnn_alpha_ptr(c.desc, direction) - c.ws.data_ptr()
    return c.ws[off:off + Ln * B * Ln * 4].view(torch.float32).view(Ln, B, Ln).clone()


def bimodel_backward(c: BiCtx, P: Getter, G: Getter, dlp: Tensor) -> None:
    """Accumulates every parameter gradient into G(name).  dlp [L,B,C]."""
    d = c.dims
    Ln, B, N, W = c.L, c.B, c.L * c.B, 2 * c.dims.D_e
    dev = dlp.device
    drop = c.drop
    with ops.wgrad_scope(None, batch=True):
        dy = torch.empty(N, d.n_classes, device=dev)
        ops.logsoftmax_tb_bwd(dlp.contiguous().view(N, d.n_classes), c.lp, dy, 1, N)
        ops.grad_weight(dy, c.hid, G("smax_fc.weight"))
        ops.colsum_acc(dy, G("smax_fc.bias"))
        dhid = torch.empty_like(c.hid)
        ops.matmul(dy, P("smax_fc.weight"), dhid)
        ops.relu_bwd_(dhid, c.hid)
        if drop is not None and drop.p_hid > 0:
            drop.site(SITE_DRNN_HID, drop.p_hid).apply_(dhid)
        ops.grad_weight(dhid, c.att, G("linear.weight"))
        ops.colsum_acc(dhid, G("linear.bias"))
        datt = torch.empty(N, W, device=dev)
        ops.matmul(dhid, P("linear.weight"), datt)
        # pooling: att[t,b] = sum_s alpha[b,t,s] E[s,b]
        dA = torch.empty(B, Ln, Ln, device=dev)
        _scores(datt, c.E, dA, Ln, B, W)                                    # d alpha[b,t,s] = <datt[t,b], E[s,b]>
        dE = torch.empty(N, W, device=dev)
        _pool(c.alpha, datt, dE, Ln, B, W, transposed=True)                 # dE[s,b]  = sum_t alpha[b,t,s] datt[t,b]
        ops.general2_rows_bwd(c.S0, c.umask, dA, B * Ln, Ln, Ln)            # dA <- d S0
        dX = torch.empty(N, W, device=dev)
        _pool(dA, c.E, dX, Ln, B, W)                                        # dX_[t,b] = sum_s dS0[b,t,s] E[s,b]
        _pool(dA, c.X_, dE, Ln, B, W, accum=True, transposed=True)          # dE[s,b] += sum_t dS0[b,t,s] X_[t,b]
        ops.grad_weight(dX, c.E, G("matchatt.transform.weight"))
        ops.colsum_acc(dX, G("matchatt.transform.bias"))
        ops.matmul(dX, P("matchatt.transform.weight"), dE, accum=True)
        if drop is not None and drop.p_rec > 0:
            for i in range(2):
                drop.site(SITE_DRNN_REC + i, drop.p_rec).apply_(dE[:, d.D_e * i:d.D_e * (i + 1)])
    grads = (ops.drnn_param_struct(_sub(G, "dialog_rnn_f.dialogue_cell.")), ops.drnn_param_struct(_sub(G, "dialog_rnn_r.dialogue_cell.")))
    desc = ops.make_drnn_desc(Ln, B, (d.D_m, d.D_g, d.D_p, d.D_e), c.U2, c.qmask, c.rev, c.params, c.E, c.ws, grads=grads, dout=dE,
                              drop=c.cell_drop)
    ops.drnn_bwd(desc)
