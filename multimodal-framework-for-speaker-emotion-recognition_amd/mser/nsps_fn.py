"""Forward / backward of ``MARN1_nsps`` and ``MARN1_no_en`` (reference model/lsthm_nsps.py:242-360, model/lsthm_no_en.py:242-360;
SURVEY.md 8(f) row f1) as explicit kernel sequences, built from the pieces of ``mser.model_fn`` / ``mser.onlysp_fn``:

* encoders as in MARN1_sps (second pass on x + first pass, :307-310); ``no_en``: the text stream skips its encoder
  (lsthm_no_en.py:306,:309) and feeds ``linear_in``'s output to the cells and the attention modules directly;
* the speaker state is ONE GRU per dialogue fed ``x[t] = [linear_in(text)[t] | audio[t]]`` -- the PRE-encoder features (:305,:177,:182)
  -- with the listener blend of :188-191 (``mser_gru_speaker_desc::listener_blend``); the LSTHM streams and the rank-1 attention
  run in the cell's persistent launches with ``ext_hq`` = that state;
* only ``h_l`` and ``h_a`` of the cells reach the head (``h`` and ``h_sp`` are computed by the reference and never used, :335-339);
* ``CrossAttention2`` is the LayerNorm'd form on the unscaled encoder outputs: LN(softmax(Q K^T / sqrt(dk)) V + x_1), dh = dk = dv =
  100 (:75-108, :287-288, :341-342);
* head (:347-355): w = softmax(p); out = nn_out(cat[w1 [hf_l | hb_l | attn2], w2 [hf_a | hb_a | attn1]] + relu(fc(x_l)));
  ``fc2(x_a)`` is computed by the reference and never used (its parameters stay gradient-less).

The recurrent chains run one after the other (GRU chains, then the cell's launches); the counter links of ``mser.onlysp_fn`` are
not used here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import functional as F_
from . import ops
from .functional import Layout
from .model_fn import DropCfg, Getter, ModelDims, _Streams, _sub
from .onlysp_fn import gru_speaker_dir_bwd, gru_speaker_dir_fwd

Tensor = torch.Tensor

SITE_NSPS_REC = 28      # + {0: hf_l, 1: hf_a, 2: hb_l, 3: hb_a}: dropout_rec on the four [L,B,H] outputs that reach the head (:317-318,:330-331)
LN_EPS = 1e-6           # CrossAttention2.layer_norm (:88)


@dataclass
class NspsCtx:
    dims: ModelDims = None
    no_en: bool = False
    L: int = 0
    B: int = 0
    x2d: Tensor = None
    xl0: Tensor = None
    enc: list = None
    x_l: Tensor = None
    x_a: Tensor = None
    rev: Tensor = None
    Hc: list = None            # per direction [N, 4H] cell output rows (h_l | h_a | z | -)
    cell_ws: Tensor = None
    cell_dirs: list = None
    gru: list = None
    cell_drop: tuple = None
    xa: list = None            # attention contexts
    ln: list = None            # per attention module: (sum [N,D], mean, rstd)
    Lb: Tensor = None          # [N, 2H + D] = hf_l | hb_l | attn2
    Ab: Tensor = None          # [N, 2H + D] = hf_a | hb_a | attn1
    wsm: Tensor = None         # softmax(p) [2]
    R: Tensor = None           # relu(fc(x_l)) (dropped) [N, 2(2H + D)]
    Z: Tensor = None
    y2: Tensor = None
    y3: Tensor = None
    lp: Tensor = None
    drop: DropCfg = None


def _xattn_ln_fwd(c: NspsCtx, i: int, P: Getter, name: str, x1: Tensor, x2: Tensor, lay: Layout, out: Tensor, drop) -> None:
    """out[N, D] (a column block of Lb / Ab) = LayerNorm(attention(x1, x2) + x1)."""
    N, D = x1.shape
    att = torch.empty(N, D, device=x1.device)
    c.xa[i] = F_.xattn_fwd(x1, None, x2, None, P(name + ".Wq"), P(name + ".Wk"), P(name + ".Wv"), lay, lay, att, 1, drop=drop)
    y = torch.empty(N, D, device=x1.device)
    ssum = torch.empty(N, D, device=x1.device)
    st = torch.empty(2, N, device=x1.device)
    ops.add_layernorm_fwd(att, x1, P(name + ".layer_norm.weight"), P(name + ".layer_norm.bias"), y, ssum, st[0], st[1], LN_EPS)
    ops.add_rows(out, y)
    c.ln[i] = (ssum, st)


def _xattn_ln_bwd(c: NspsCtx, i: int, P: Getter, G: Getter, name: str, dout: Tensor, dx1: Tensor, dx2: Tensor) -> None:
    """dout [N, D] (a column block, any leading dimension): accumulates into dx1 (query side + residual) and dx2."""
    ssum, st = c.ln[i]
    N, D = ssum.shape
    dy = torch.empty(N, D, device=ssum.device)
    ops.add_rows(dy, dout)
    dsum = torch.empty(N, D, device=ssum.device)
    ops.layernorm_bwd(dy, ssum, st[0], st[1], P(name + ".layer_norm.weight"), dsum, G(name + ".layer_norm.weight"),
                      G(name + ".layer_norm.bias"))
    ops.add_rows(dx1, dx1, dsum)                                        # the residual x_1 (:92,:105)
    F_.xattn_bwd(c.xa[i], dsum, P(name + ".Wq"), P(name + ".Wk"), P(name + ".Wv"), G(name + ".Wq"), G(name + ".Wk"), G(name + ".Wv"),
                 dx1, dx2, None, None)


def nsps_forward(P: Getter, x: Tensor, qmask: Tensor, umask: Tensor, dims: ModelDims, no_en: bool = False,
                 drop: Optional[DropCfg] = None):
    """x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L] -> (log_probs [B*L,C], x_l [L,B,D], x_a [L,B,D], ctx)."""
    if drop is not None and not drop.any():
        drop = None
    Ln, B, Fin = x.shape
    d = dims
    for t, nm in ((x, "x"), (qmask, "qmask"), (umask, "umask")):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError(f"{nm} must be a float32 GPU tensor (got {t.dtype} on {t.device})")
    if Fin < d.d_r + d.d_a:
        raise RuntimeError(f"x has {Fin} features, model expects d_r+d_a = {d.d_r + d.d_a}")
    x, qmask, umask = x.contiguous(), qmask.contiguous(), umask.contiguous()
    N, D, H = Ln * B, d.D, d.H
    dev = x.device
    lay = Layout.time_major(Ln, B)
    c = NspsCtx(dims=d, no_en=no_en, L=Ln, B=B, drop=drop)
    c.x2d = x.view(N, Fin)

    def enc_drops(call):
        if drop is None:
            return None
        ps = drop.p_enc_l if call < 2 else drop.p_enc_a
        return tuple(drop.site(F_.SITE_ENC + 3 * call + i, ps[i]) for i in range(3))

    # ---- linear_in, encoders (audio branch on a side stream)
    c.xl0 = torch.empty(N, D, device=dev)
    xa0 = c.x2d[:, d.d_r:d.d_r + d.d_a]
    c.x_a = torch.empty(N, D, device=dev)
    c.enc = [None] * 4
    Pl, Pa = _sub(P, "encoder_l."), _sub(P, "encoder_a.")
    cur = torch.cuda.current_stream()
    s_audio, s_x = _Streams.get(dev)[:2]
    s_audio.wait_stream(cur)
    with torch.cuda.stream(s_audio):
        e1a, c.enc[2] = F_.encoder_layer_fwd(xa0, None, Pa, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(2), need_attn=False)
        _, c.enc[3] = F_.encoder_layer_fwd(xa0, e1a, Pa, lay, d.n_head, d.d_k, d.d_v, out=c.x_a, drops=enc_drops(3), need_attn=False)
    ops.linear(c.x2d[:, :d.d_r], P("linear_in.weight"), c.xl0, bias=P("linear_in.bias"))
    # The speaker chains of this variant read the PRE-encoder features (linear_in's output and the raw audio) and qmask only: they run
    # on a stream of their own beside the encoders (two 16-dialogue blocks per direction: 4 CUs), complete before the LSTHM chains start.
    lens = torch.empty(B, device=dev, dtype=torch.int32)
    c.rev = torch.empty(Ln, B, device=dev, dtype=torch.int32)
    ops.build_reverse_index(umask, lens, c.rev)
    s_g = _Streams.get(dev)[2]
    s_g.wait_stream(cur)
    c.gru = []
    with torch.cuda.stream(s_g):
        for i, (pre, rev) in enumerate((("marn_cell_f.", None), ("marn_cell_b.", c.rev))):
            site = drop.site(F_.SITE_CELL + 4 * i, drop.p_cell[i]) if drop is not None else None
            c.gru.append(gru_speaker_dir_fwd(_sub(P, pre), c.xl0, xa0, qmask, rev, None, Ln, B, H, site, launch=False, lblend=True))
        ops.gru_speaker_fwd([g.desc for g in c.gru])
    if no_en:
        c.x_l = c.xl0                                                   # lsthm_no_en.py:306,:309: no text encoder
    else:
        c.x_l = torch.empty(N, D, device=dev)
        e1, c.enc[0] = F_.encoder_layer_fwd(c.xl0, None, Pl, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(0), need_attn=False)
        _, c.enc[1] = F_.encoder_layer_fwd(c.xl0, e1, Pl, lay, d.n_head, d.d_k, d.d_v, out=c.x_l, drops=enc_drops(1), need_attn=False)
    cur.wait_stream(s_audio)

    W = 2 * H + D
    c.Lb, c.Ab = torch.empty(N, W, device=dev), torch.empty(N, W, device=dev)
    # ---- LayerNorm'd sequence attention (:341-342) on a side stream beside the recurrent chains
    c.xa, c.ln = [None, None], [None, None]

    def xa_drop(i):
        return drop.site(F_.SITE_XATTN + i, drop.p_xattn[i]) if drop is not None else None

    s_x.wait_stream(cur)
    with torch.cuda.stream(s_x):
        _xattn_ln_fwd(c, 0, P, "crossatt_l2a", c.x_l, c.x_a, lay, c.Ab[:, 2 * H:], xa_drop(0))      # attn1 -> a (:353)
        _xattn_ln_fwd(c, 1, P, "crossatt_a2l", c.x_a, c.x_l, lay, c.Lb[:, 2 * H:], xa_drop(1))      # attn2 -> l (:352)

    # ---- the two cells: GRU speaker chains on the pre-encoder features, then the LSTHM chains of both directions in one launch
    c.cell_ws = torch.empty(ops.cell_workspace_bytes(Ln, B, D, H, 2), device=dev, dtype=torch.uint8)
    c.Hc = [torch.empty(N, 4 * H, device=dev), torch.empty(N, 4 * H, device=dev)]
    c.cell_dirs = [
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_f.")), qmask=qmask, rev=None, out=c.Hc[0]),
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_b.")), qmask=qmask, rev=c.rev, out=c.Hc[1]),
    ]
    if drop is not None and (any(p_ > 0 for p_ in drop.p_cell) or any(p_ > 0 for p_ in drop.p_cell_attn)):
        c.cell_drop = (drop.rng, [F_.SITE_CELL, F_.SITE_CELL + 4], drop.p_cell, drop.p_cell_attn)
    desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 4 * H, c.cell_ws, drop=c.cell_drop,
                              ext_hq=[g.hs for g in c.gru])
    ops.marn_cell_run(desc, ops.PHASE_FWD_PREP)
    cur.wait_stream(s_g)                                             # the speaker rows h_s are complete
    ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)
    # h_l / h_a of both directions -> their column blocks of l and a; dropout_rec on each of the four (:317-318,:330-331)
    for i in range(2):
        ops.add_rows(c.Lb[:, H * i:H * (i + 1)], c.Hc[i][:, 0:H])
        ops.add_rows(c.Ab[:, H * i:H * (i + 1)], c.Hc[i][:, H:2 * H])
        if drop is not None and drop.p_rec > 0:
            drop.site(SITE_NSPS_REC + 2 * i, drop.p_rec).apply_(c.Lb[:, H * i:H * (i + 1)])
            drop.site(SITE_NSPS_REC + 2 * i + 1, drop.p_rec).apply_(c.Ab[:, H * i:H * (i + 1)])
    cur.wait_stream(s_x)

    # ---- fusion (:347-355): softmax(p), fc residual, weighted concatenation
    c.wsm = P("p").detach().clone().view(1, 2)
    ops.softmax_rows_(c.wsm, 1, 2, 2)
    c.R = torch.empty(N, 2 * W, device=dev)
    ops.linear(c.x_l, P("fc.0.weight"), c.R, bias=P("fc.0.bias"), relu=True)
    if drop is not None and drop.p_fc > 0:
        drop.site(F_.SITE_FC, drop.p_fc).apply_(c.R)
    c.Z = c.R.clone()
    ops.scale_acc_dot(c.Z[:, :W], c.Lb, None, c.wsm[0, 0:1], None)
    ops.scale_acc_dot(c.Z[:, W:], c.Ab, None, c.wsm[0, 1:2], None)
    # ---- nn_out + log_softmax
    h_out = P("nn_out.0.weight").shape[0]
    c.y2 = torch.empty(N, h_out, device=dev)
    ops.linear(c.Z, P("nn_out.0.weight"), c.y2, bias=P("nn_out.0.bias"), relu=True)
    if drop is not None and drop.p_out > 0:
        drop.site(F_.SITE_OUT, drop.p_out).apply_(c.y2)
    c.y3 = torch.empty(N, d.n_classes, device=dev)
    ops.linear(c.y2, P("nn_out.3.weight"), c.y3, bias=P("nn_out.3.bias"))
    c.lp = torch.empty(B * Ln, d.n_classes, device=dev)
    ops.logsoftmax_tb_fwd(c.y3, c.lp, Ln, B)
    return c.lp, c.x_l.view(Ln, B, D), c.x_a.view(Ln, B, D), c


def nsps_backward(c: NspsCtx, P: Getter, G: Getter, dlp: Tensor, dx_l_out: Optional[Tensor] = None,
                  dx_a_out: Optional[Tensor] = None) -> None:
    """Accumulates every parameter gradient into G(name)."""
    d = c.dims
    Ln, B, N, D, H = c.L, c.B, c.L * c.B, c.dims.D, c.dims.H
    W = 2 * H + D
    dev = dlp.device
    drop = c.drop
    with ops.wgrad_scope(None, batch=True):
        # ---- nn_out
        dy3 = torch.empty(N, d.n_classes, device=dev)
        ops.logsoftmax_tb_bwd(dlp.contiguous(), c.lp, dy3, Ln, B)
        ops.grad_weight(dy3, c.y2, G("nn_out.3.weight"))
        ops.colsum_acc(dy3, G("nn_out.3.bias"))
        dy2 = torch.empty_like(c.y2)
        ops.matmul(dy3, P("nn_out.3.weight"), dy2)
        ops.relu_bwd_(dy2, c.y2)                       # the saved y2 is the dropped one: a dropped unit reads 0 and fails the ReLU test
        if drop is not None and drop.p_out > 0:
            drop.site(F_.SITE_OUT, drop.p_out).apply_(dy2)
        ops.grad_weight(dy2, c.Z, G("nn_out.0.weight"))
        ops.colsum_acc(dy2, G("nn_out.0.bias"))
        dZ = torch.empty(N, 2 * W, device=dev)
        ops.matmul(dy2, P("nn_out.0.weight"), dZ)
        # ---- fusion: d(softmax(p)), the two weighted concatenations, the fc residual
        dx_l = torch.zeros(N, D, device=dev) if dx_l_out is None else dx_l_out.reshape(N, D).clone()
        dx_a = torch.zeros(N, D, device=dev) if dx_a_out is None else dx_a_out.reshape(N, D).clone()
        dLb, dAb = torch.zeros(N, W, device=dev), torch.zeros(N, W, device=dev)
        dw = torch.zeros(1, 2, device=dev)
        ops.scale_acc_dot(dLb, dZ[:, :W], c.Lb, c.wsm[0, 0:1], dw[0, 0:1])
        ops.scale_acc_dot(dAb, dZ[:, W:], c.Ab, c.wsm[0, 1:2], dw[0, 1:2])
        ops.softmax_bwd_rows_(c.wsm, dw, 1, 2, 2)
        ops.add_rows(G("p").view(1, 2), G("p").view(1, 2), dw)
        dR = dZ                                          # (dZ is not needed beyond this point: reuse it)
        ops.relu_bwd_(dR, c.R)
        if drop is not None and drop.p_fc > 0:
            drop.site(F_.SITE_FC, drop.p_fc).apply_(dR)
        ops.grad_weight(dR, c.x_l, G("fc.0.weight"))
        ops.colsum_acc(dR, G("fc.0.bias"))
        ops.matmul(dR, P("fc.0.weight"), dx_l, accum=True)
        # ---- sequence attention modules (side stream, their own accumulators)
        cur = torch.cuda.current_stream()
        s_audio, s_x = _Streams.get(dev)[:2]
        dxl_x, dxa_x = torch.zeros(N, D, device=dev), torch.zeros(N, D, device=dev)
        s_x.wait_stream(cur)
        with torch.cuda.stream(s_x):
            _xattn_ln_bwd(c, 0, P, G, "crossatt_l2a", dAb[:, 2 * H:], dxl_x, dxa_x)
            _xattn_ln_bwd(c, 1, P, G, "crossatt_a2l", dLb[:, 2 * H:], dxa_x, dxl_x)
        # ---- the cells: gradients arrive at h_l / h_a only
        dHc = [torch.zeros(N, 4 * H, device=dev), torch.zeros(N, 4 * H, device=dev)]
        for i in range(2):
            if drop is not None and drop.p_rec > 0:
                drop.site(SITE_NSPS_REC + 2 * i, drop.p_rec).apply_(dLb[:, H * i:H * (i + 1)])
                drop.site(SITE_NSPS_REC + 2 * i + 1, drop.p_rec).apply_(dAb[:, H * i:H * (i + 1)])
            ops.add_rows(dHc[i][:, 0:H], dLb[:, H * i:H * (i + 1)])
            ops.add_rows(dHc[i][:, H:2 * H], dAb[:, H * i:H * (i + 1)])
        dhq = [torch.empty(N, H, device=dev) for _ in range(2)]
        for r, pre, i in ((c.cell_dirs[0], "marn_cell_f.", 0), (c.cell_dirs[1], "marn_cell_b.", 1)):
            r["g"] = ops.cell_param_struct(_sub(G, pre))
            r["dout"] = dHc[i]
        desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 4 * H, c.cell_ws, dx_l=dx_l, dx_a=dx_a, drop=c.cell_drop,
                                  ext_hq=[g.hs for g in c.gru], ext_dhq=dhq)
        dgs = [(torch.empty(N, 3 * H, device=dev), torch.empty(N, 3 * H, device=dev)) for _ in range(2)]
        ops.marn_cell_run(desc, ops.PHASE_BWD_PREP | ops.PHASE_LSTHM_BWD)
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX | ops.PHASE_LSTHM_WGRAD | ops.PHASE_SPEAKER_BWD)
        # the GRU BPTT feeds only linear_in (through dxl0) and its own parameters: on its own stream beside the encoders' backward
        dxl0 = torch.zeros(N, D, device=dev)              # gradient at linear_in's output through the GRU inputs (x = cat[x_l0 | audio])
        s_g = _Streams.get(dev)[2]
        s_g.wait_stream(cur)
        with torch.cuda.stream(s_g):
            for i in range(2):
                ops.gru_speaker_set_grads(c.gru[i].desc, dhq[i], dgs[i][0], dgs[i][1])
            ops.gru_speaker_bwd([g.desc for g in c.gru])
            for i, pre in enumerate(("marn_cell_f.", "marn_cell_b.")):
                gru_speaker_dir_bwd(c.gru[i], _sub(P, pre), _sub(G, pre), dgs[i][0], dgs[i][1], dxl0, None, Ln, B, H)
        # ---- encoders and linear_in
        cur.wait_stream(s_x)
        ops.add_rows(dx_l, dx_l, dxl_x)
        ops.add_rows(dx_a, dx_a, dxa_x)
        Pl, Pa, Gl, Ga = _sub(P, "encoder_l."), _sub(P, "encoder_a."), _sub(G, "encoder_l."), _sub(G, "encoder_a.")
        s_audio.wait_stream(cur)
        with torch.cuda.stream(s_audio):
            d2a = F_.encoder_layer_bwd(c.enc[3], dx_a, Pa, Ga)         # grad of (audio + first pass): flows to both
            F_.encoder_layer_bwd(c.enc[2], d2a, Pa, Ga)                # the raw audio features need no gradient
        if c.no_en:
            cur.wait_stream(s_g)                                       # dxl0 and the GRU gradient products' operands are complete
            ops.add_rows(dxl0, dxl0, dx_l)
        else:
            d2 = F_.encoder_layer_bwd(c.enc[1], dx_l, Pl, Gl)
            d1 = F_.encoder_layer_bwd(c.enc[0], d2, Pl, Gl)
            cur.wait_stream(s_g)
            ops.add_rows(dxl0, dxl0, d1)
            ops.add_rows(dxl0, dxl0, d2)
        ops.grad_weight(dxl0, c.x2d[:, :d.d_r], G("linear_in.weight"))
        ops.colsum_acc(dxl0, G("linear_in.bias"))
        cur.wait_stream(s_audio)
