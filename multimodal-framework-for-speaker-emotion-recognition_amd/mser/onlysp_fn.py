"""Forward / backward of ``MARN1_onlysp`` (reference model/lsthm_onlysp.py:209-300, the reference CLI's default model; SURVEY.md 8(f)
row f1) as explicit kernel sequences, built from the same pieces as ``mser.model_fn``:

* the speaker state is ONE GRU per dialogue (``mser_gru_speaker_fwd/bwd``, csrc/gru_speaker.hip) fed ``[x_l[t] | x_a[t]]``; its
  input product is a GEMM over all steps, its weight gradients are GEMMs after the chain;
* the LSTHM streams and the rank-1 attention run in the cell's persistent launches with ``ext_hq`` = that state
  (``mser_cell_desc``), the cell returns the total gradient at it (``ext_dhq``);
* the second encoder pass takes the first pass's output without the residual add (:262-266);
* the head is ``nn_out`` = Linear(10H -> 32) + ReLU + Dropout + Linear(32 -> C) on cat[h_f, h_b, attn1, attn2] (:287).

Schedule: in eager mode the GRU chains run CONCURRENTLY with the LSTHM chains, linked through the cell's step counters
(LINK_GRU_FWD / LINK_GRU_BWD below); under stream capture they run before / after them.  The audio encoder branch and the
sequence-level attention modules run on side streams beside the text branch and the recurrent chains.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import functional as F_
from . import fault, ops
from .functional import Layout
from .model_fn import DropCfg, Getter, ModelDims, _Streams, _sub

Tensor = torch.Tensor


@dataclass
class GruDirCtx:
    desc: object = None
    xl: Tensor = None          # inputs in the direction's time order [T*B, D]
    xa: Tensor = None
    qm: Tensor = None          # qmask in the direction's time order [T, B, 2]
    rev: Optional[Tensor] = None
    hs: Tensor = None
    save: Tensor = None
    gi: Tensor = None
    drop: object = None
    progress: Tensor = None


# The GRU forward chains run CONCURRENTLY with the LSTHM forward chains (eager mode, persistent launches): they write the speaker
# rows straight into the cell's workspace and advance the step counter the LSTHM chain already waits on (mser_cell_desc::ext_linked).
# False, stream capture, or sizes without the persistent launch: the GRU chains finish first and the rows are copied.
LINK_GRU_FWD = True
# Likewise the GRU BPTT runs concurrently with the LSTHM BPTT as its consumer: step t starts when the cell's BPTT counter shows that
# the gradient rows of h_s[t] are complete (bounded wait), and reads them from the workspace with device-coherent loads.
LINK_GRU_BWD = True


def gru_speaker_dir_fwd(P: Getter, x_l: Tensor, x_a: Tensor, qmask: Tensor, rev: Optional[Tensor], out_q: Optional[Tensor], T: int,
                        B: int, H: int, drop, launch: bool = True, hs: Optional[Tensor] = None, lblend: bool = False,
                        hli: Optional[Tensor] = None) -> GruDirCtx:
    """One direction's speaker chain.  x_l / x_a [T*B, D] natural order; ``rev`` (int32 [T, B]) selects the reversed direction;
    ``out_q`` is the h_s quarter of the cell output rows (natural order).  P: names relative to the MARN_cell.  ``launch=False``
    prepares everything (input product included) and leaves the chain launch (ops.gru_speaker_fwd(ctx.desc)) to the caller."""
    D = x_l.shape[1]
    c = GruDirCtx(rev=rev, drop=drop)
    if rev is not None:
        c.xl, c.xa = torch.empty(T * B, D, device=x_l.device), torch.empty(T * B, D, device=x_l.device)
        ops.reverse_by_length(x_l, rev, c.xl, T, B)
        ops.reverse_by_length(x_a, rev, c.xa, T, B)
        c.qm = torch.empty(T, B, 2, device=x_l.device)
        ops.reverse_by_length(qmask.view(T * B, 2), rev, c.qm.view(T * B, 2), T, B)
    else:
        c.xl, c.xa, c.qm = x_l, x_a, qmask
    Wih = P("gru_s.weight_ih")
    c.gi = torch.empty(T * B, 3 * H, device=x_l.device)
    ops.linear(c.xl, Wih[:, :D], c.gi, bias=P("gru_s.bias_ih"))                 # gi = [x_l | x_a] W_ih^T + b_ih  (:172,:177)
    ops.linear(c.xa, Wih[:, D:], c.gi, accum=True)
    c.hs = hs if hs is not None else torch.empty(T * B, H, device=x_l.device)
    c.save = torch.empty(T * B, 5 * H, device=x_l.device)
    c.desc = ops.gru_speaker_desc(T, B, H, c.gi, P("gru_s.weight_hh"), P("gru_s.bias_hh"), c.qm, c.hs, c.save, out=out_q, rev=rev,
                                  drop=drop, lblend=lblend, hli=hli)
    if launch:
        ops.gru_speaker_fwd(c.desc)
    return c


def gru_speaker_dir_bwd(c: GruDirCtx, P: Getter, G: Getter, dgi: Tensor, dgh: Tensor, dx_l: Optional[Tensor], dx_a: Optional[Tensor],
                        T: int, B: int, H: int) -> None:
    """After the chain's BPTT (ops.gru_speaker_bwd) has produced dgi / dgh: accumulates the GRU parameter gradients into G and the
    input gradients into dx_l / dx_a (natural order; None: that input needs no gradient)."""
    D = c.xl.shape[1]
    dev = dgi.device
    gWih = G("gru_s.weight_ih")
    ops.grad_weight(dgi, c.xl, gWih[:, :D])
    ops.grad_weight(dgi, c.xa, gWih[:, D:])
    ops.colsum_acc(dgi, G("gru_s.bias_ih"))
    ops.grad_weight(dgh, c.save[:, :H], G("gru_s.weight_hh"))                  # qs0 rows
    ops.colsum_acc(dgh, G("gru_s.bias_hh"))
    Wih = P("gru_s.weight_ih")
    for X, Wp, dx in ((c.xl, Wih[:, :D], dx_l), (c.xa, Wih[:, D:], dx_a)):
        if dx is None:
            continue
        if c.rev is None:
            ops.matmul(dgi, Wp, dx, accum=True)
        else:
            t = torch.empty(T * B, D, device=dev)
            ops.matmul(dgi, Wp, t)
            u = torch.empty(T * B, D, device=dev)
            ops.reverse_by_length(t, c.rev, u, T, B)                             # the reversal is an involution on the valid rows
            ops.add_rows(dx, dx, u)


@dataclass
class OnlyspCtx:
    dims: ModelDims = None
    L: int = 0
    B: int = 0
    x2d: Tensor = None
    xl0: Tensor = None
    enc: list = None
    x_l: Tensor = None
    x_a: Tensor = None
    rev: Tensor = None
    Hcat: Tensor = None
    cell_ws: Tensor = None
    cell_dirs: list = None
    gru: list = None
    cell_drop: tuple = None
    xa: list = None
    A1: Tensor = None
    A2: Tensor = None
    y2: Tensor = None
    y3: Tensor = None
    lp: Tensor = None
    drop: DropCfg = None
    qmask: Tensor = None
    gru_status: Tensor = None


def onlysp_forward(P: Getter, x: Tensor, qmask: Tensor, umask: Tensor, dims: ModelDims, drop: Optional[DropCfg] = None):
    """x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L] -> (log_probs [B*L,C], x_l [L,B,D], x_a [L,B,D], ctx)."""
    if drop is not None and not drop.any():
        drop = None
    Ln, B, Fin = x.shape
    d = dims
    for t, nm in ((x, "x"), (qmask, "qmask"), (umask, "umask")):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError(f"{nm} must be a float32 GPU tensor (got {t.dtype} on {t.device})")
    x, qmask, umask = x.contiguous(), qmask.contiguous(), umask.contiguous()
    N, D, H = Ln * B, d.D, d.H
    dev = x.device
    lay = Layout.time_major(Ln, B)
    c = OnlyspCtx(dims=d, L=Ln, B=B, drop=drop, qmask=qmask)
    c.x2d = x.view(N, Fin)

    def enc_drops(call):
        if drop is None:
            return None
        ps = drop.p_enc_l if call < 2 else drop.p_enc_a
        return tuple(drop.site(F_.SITE_ENC + 3 * call + i, ps[i]) for i in range(3))

    # ---- linear_in + two encoder passes per modality; the second pass takes the first pass's output (:262-266)
    c.xl0 = torch.empty(N, D, device=dev)
    xa0 = c.x2d[:, d.d_r:d.d_r + d.d_a]
    c.x_l, c.x_a = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev)
    c.enc = [None] * 4
    Pl, Pa = _sub(P, "encoder_l."), _sub(P, "encoder_a.")
    cur = torch.cuda.current_stream()
    s_audio, s_x = _Streams.get(dev)[:2]
    s_audio.wait_stream(cur)
    with torch.cuda.stream(s_audio):
        e1a, c.enc[2] = F_.encoder_layer_fwd(xa0, None, Pa, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(2), need_attn=False)
        _, c.enc[3] = F_.encoder_layer_fwd(e1a, None, Pa, lay, d.n_head, d.d_k, d.d_v, out=c.x_a, drops=enc_drops(3), need_attn=False)
    ops.linear(c.x2d[:, :d.d_r], P("linear_in.weight"), c.xl0, bias=P("linear_in.bias"))
    e1, c.enc[0] = F_.encoder_layer_fwd(c.xl0, None, Pl, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(0), need_attn=False)
    _, c.enc[1] = F_.encoder_layer_fwd(e1, None, Pl, lay, d.n_head, d.d_k, d.d_v, out=c.x_l, drops=enc_drops(1), need_attn=False)
    cur.wait_stream(s_audio)                     # x_l and x_a are final

    c.Hcat = torch.empty(N, 10 * H, device=dev)
    # ---- sequence-level cross-modal attention (:277-283): needs only the encoder outputs -- issued on a side stream, it runs beside
    # the recurrent chains (which leave most of the chip idle); it writes the last two H-wide column blocks of Hcat
    w, v, v1, v2 = P("w"), P("v"), P("v1"), P("v2")
    c.A1, c.A2 = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
    c.xa = [None] * 4

    def xa_drop(i):
        return drop.site(F_.SITE_XATTN + i, drop.p_xattn[i]) if drop is not None else None

    # (eager mode: the HOST needs ~8 us per launch, so the recurrent chains -- the critical path -- are issued first and the 8 launches
    # of the attention modules behind them; the side stream only waits for the encoders' outputs, through this event)
    ev_x = torch.cuda.Event()
    ev_x.record(cur)

    def xattn_branch():
        s_x.wait_event(ev_x)
        with torch.cuda.stream(s_x):
            c.xa[0] = F_.xattn_fwd(c.x_l, w, c.x_a, v, P("crossatt_l2a.Wq"), P("crossatt_l2a.Wk"), P("crossatt_l2a.Wv"), lay, lay, c.A1, 1,
                                   drop=xa_drop(0))
            c.xa[2] = F_.xattn_fwd(c.x_a, v, c.A1, v1, P("crossatt_l2a_1.Wq"), P("crossatt_l2a_1.Wk"), P("crossatt_l2a_1.Wv"), lay, lay,
                                   c.Hcat[:, 8 * H:9 * H], 1, drop=xa_drop(2))
            c.xa[1] = F_.xattn_fwd(c.x_a, v, c.x_l, w, P("crossatt_a2l.Wq"), P("crossatt_a2l.Wk"), P("crossatt_a2l.Wv"), lay, lay, c.A2, 1,
                                   drop=xa_drop(1))
            c.xa[3] = F_.xattn_fwd(c.x_l, w, c.A2, v2, P("crossatt_a2l_1.Wq"), P("crossatt_a2l_1.Wk"), P("crossatt_a2l_1.Wv"), lay, lay,
                                   c.Hcat[:, 9 * H:10 * H], 1, drop=xa_drop(3))


    # ---- the two cells: GRU speaker chains, then the LSTHM chains of both directions in one launch
    lens = torch.empty(B, device=dev, dtype=torch.int32)
    c.rev = torch.empty(Ln, B, device=dev, dtype=torch.int32)
    ops.build_reverse_index(umask, lens, c.rev)
    c.cell_ws = torch.empty(ops.cell_workspace_bytes(Ln, B, D, H, 2), device=dev, dtype=torch.uint8)
    c.cell_dirs = [
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_f.")), qmask=qmask, rev=None, out=c.Hcat[:, 0:4 * H]),
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_b.")), qmask=qmask, rev=c.rev, out=c.Hcat[:, 4 * H:8 * H]),
    ]
    if drop is not None and (any(p_ > 0 for p_ in drop.p_cell) or any(p_ > 0 for p_ in drop.p_cell_attn)):
        c.cell_drop = (drop.rng, [F_.SITE_CELL, F_.SITE_CELL + 4], drop.p_cell, drop.p_cell_attn)
    # linked mode: where the producer publishes (workspace rows + step counter), asked of a descriptor without ext fields
    links = None
    if LINK_GRU_FWD and not torch.cuda.is_current_stream_capturing():
        probe = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 10 * H, c.cell_ws)
        gru_wgs = 2 * ((B + 15) // 16)            # the producer launch: both directions' chains, one 16-dialogue block per workgroup
        links = [ops.cell_ext_link(probe, i, gru_wgs) for i in range(2)]
        if not all(lk[5] for lk in links):
            links = None
    c.gru = []
    for i, (pre, rev) in enumerate((("marn_cell_f.", None), ("marn_cell_b.", c.rev))):
        site = drop.site(F_.SITE_CELL + 4 * i, drop.p_cell[i]) if drop is not None else None
        hs = None
        if links is not None:      # the chain's output rows ARE the workspace's h_q rows
            off = links[i][0] - c.cell_ws.data_ptr()
            hs = c.cell_ws[off:off + N * H * 4].view(torch.float32).view(N, H)
        g = gru_speaker_dir_fwd(_sub(P, pre), c.x_l, c.x_a, qmask, rev, c.Hcat[:, 4 * H * i + 3 * H:4 * H * (i + 1)], Ln, B, H, site,
                                launch=False, hs=hs)
        if links is not None:
            g.desc.pub_counter, g.desc.pub_replicas, g.desc.pub_replica_stride, g.desc.pub_per_step = links[i][1:5]
            g.progress = torch.zeros((B + 15) // 16, device=dev, dtype=torch.int32)      # per-block progress words of the chain
            g.desc.pub_progress = g.progress.data_ptr()
        c.gru.append(g)
    desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 10 * H, c.cell_ws, drop=c.cell_drop,
                              ext_hq=[g.hs for g in c.gru], ext_linked=links is not None)
    # FWD_PREP zeroes the reversed direction's output rows (h_s quarter included: rows at and beyond len_b stay zero) and the step
    # counters, so the speaker chains, which write that quarter and advance a counter, go after it
    ops.marn_cell_run(desc, ops.PHASE_FWD_PREP)
    if links is not None:
        s_g = _Streams.get(dev)[2]
        s_g.wait_stream(cur)
        with torch.cuda.stream(s_g):
            ops.gru_speaker_fwd([g.desc for g in c.gru])             # producer: both directions' chains, one launch, its own stream
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)                 # consumer: follows the producer step by step
        xattn_branch()
        cur.wait_stream(s_g)
    else:
        ops.gru_speaker_fwd([g.desc for g in c.gru])                 # both directions' chains in one launch
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)
        xattn_branch()
    cur.wait_stream(s_x)
    if drop is not None and drop.p_rec > 0:
        for i in range(2):
            drop.site(F_.SITE_REC + i, drop.p_rec).apply_(c.Hcat[:, 4 * H * i:4 * H * (i + 1)])

    # ---- head: nn_out on the concatenation (:287)
    h_out = P("nn_out.0.weight").shape[0]
    c.y2 = torch.empty(N, h_out, device=dev)
    ops.linear(c.Hcat, P("nn_out.0.weight"), c.y2, bias=P("nn_out.0.bias"), relu=True)
    if drop is not None and drop.p_out > 0:
        drop.site(F_.SITE_OUT, drop.p_out).apply_(c.y2)
    c.y3 = torch.empty(N, d.n_classes, device=dev)
    ops.linear(c.y2, P("nn_out.3.weight"), c.y3, bias=P("nn_out.3.bias"))
    c.lp = torch.empty(B * Ln, d.n_classes, device=dev)
    ops.logsoftmax_tb_fwd(c.y3, c.lp, Ln, B)
    return c.lp, c.x_l.view(Ln, B, D), c.x_a.view(Ln, B, D), c


def onlysp_backward(c: OnlyspCtx, P: Getter, G: Getter, dlp: Tensor, dx_l_out: Optional[Tensor] = None,
                    dx_a_out: Optional[Tensor] = None, status: Optional[Tensor] = None) -> None:
    """Accumulates every parameter gradient into G(name).  ``status`` (int32 [1] on the device, optional): set to 1 by a linked GRU
    BPTT that gave up waiting for the cell's BPTT (never cleared here: a caller may keep one across steps and look at it rarely)."""
    d = c.dims
    Ln, B, N, D, H = c.L, c.B, c.L * c.B, c.dims.D, c.dims.H
    dev = dlp.device
    drop = c.drop
    with ops.wgrad_scope(None, batch=True):
        # ---- head
        dy3 = torch.empty(N, d.n_classes, device=dev)
        ops.logsoftmax_tb_bwd(dlp.contiguous(), c.lp, dy3, Ln, B)
        ops.grad_weight(dy3, c.y2, G("nn_out.3.weight"))
        ops.colsum_acc(dy3, G("nn_out.3.bias"))
        dy2 = torch.empty_like(c.y2)
        ops.matmul(dy3, P("nn_out.3.weight"), dy2)
        ops.relu_bwd_(dy2, c.y2)                       # the saved y2 is the dropped one: a dropped unit reads 0 and fails the ReLU test
        if drop is not None and drop.p_out > 0:
            drop.site(F_.SITE_OUT, drop.p_out).apply_(dy2)
        ops.grad_weight(dy2, c.Hcat, G("nn_out.0.weight"))
        ops.colsum_acc(dy2, G("nn_out.0.bias"))
        dH = torch.empty(N, 10 * H, device=dev)
        ops.matmul(dy2, P("nn_out.0.weight"), dH)
        if drop is not None and drop.p_rec > 0:
            for i in range(2):
                drop.site(F_.SITE_REC + i, drop.p_rec).apply_(dH[:, 4 * H * i:4 * H * (i + 1)])
        # ---- sequence-level attention modules: on a side stream beside the recurrent chains, into accumulators of their own
        cur = torch.cuda.current_stream()
        s_audio, s_x = _Streams.get(dev)[:2]
        zbuf = torch.zeros(4 * N * D + 2 * N * H, device=dev)          # ONE fill for the six zero-initialised accumulators
        zd = [zbuf[i * N * D:(i + 1) * N * D].view(N, D) for i in range(4)]
        dx_l = zd[0] if dx_l_out is None else dx_l_out.reshape(N, D).clone()
        dx_a = zd[1] if dx_a_out is None else dx_a_out.reshape(N, D).clone()
        dxl_x, dxa_x = zd[2], zd[3]
        dA1, dA2 = zbuf[4 * N * D:4 * N * D + N * H].view(N, H), zbuf[4 * N * D + N * H:].view(N, H)

        def xb(i, name, dout, dx1, dx2, ga1, ga2):
            F_.xattn_bwd(c.xa[i], dout, P(name + ".Wq"), P(name + ".Wk"), P(name + ".Wv"), G(name + ".Wq"), G(name + ".Wk"),
                         G(name + ".Wv"), dx1, dx2, ga1, ga2)

        ev_h = torch.cuda.Event()
        ev_h.record(cur)                               # dH and the zeroed accumulators are ready

        def xattn_branch():                            # issued BEHIND the BPTT launches (host issue order, as in the forward)
            s_x.wait_event(ev_h)
            with torch.cuda.stream(s_x):
                xb(2, "crossatt_l2a_1", dH[:, 8 * H:9 * H], dxa_x, dA1, G("v"), G("v1"))
                xb(0, "crossatt_l2a", dA1, dxl_x, dxa_x, G("w"), G("v"))
                xb(3, "crossatt_a2l_1", dH[:, 9 * H:10 * H], dxl_x, dA2, G("w"), G("v2"))
                xb(1, "crossatt_a2l", dA2, dxa_x, dxl_x, G("v"), G("w"))
        # ---- LSTHM chains of both directions (ext speaker state), then the GRU chains
        dhq = [torch.empty(N, H, device=dev) for _ in range(2)]
        for r, pre, sl in ((c.cell_dirs[0], "marn_cell_f.", slice(0, 4 * H)), (c.cell_dirs[1], "marn_cell_b.", slice(4 * H, 8 * H))):
            r["g"] = ops.cell_param_struct(_sub(G, pre))
            r["dout"] = dH[:, sl]
        desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 10 * H, c.cell_ws, dx_l=dx_l, dx_a=dx_a, drop=c.cell_drop,
                                  ext_hq=[g.hs for g in c.gru], ext_dhq=dhq)
        dgs = [(torch.empty(N, 3 * H, device=dev), torch.empty(N, 3 * H, device=dev)) for _ in range(2)]
        blinks = None
        if LINK_GRU_BWD and not torch.cuda.is_current_stream_capturing():
            blinks = [ops.cell_ext_link_bwd(desc, i, 2 * ((B + 15) // 16)) for i in range(2)]
            if not all(lk[6] for lk in blinks):
                blinks = None
        ops.marn_cell_run(desc, ops.PHASE_BWD_PREP)                # zeroes the BPTT step counters: before producer AND consumer
        if blinks is not None:
            c.gru_status = status if status is not None else fault.word(dev)
            for i in range(2):
                ops.gru_speaker_link_bwd(c.gru[i].desc, blinks[i], dgs[i][0], dgs[i][1], c.gru_status)
            s_g = _Streams.get(dev)[2]
            s_g.wait_stream(cur)
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD)            # producer (main stream)
            with torch.cuda.stream(s_g):
                ops.gru_speaker_bwd([g.desc for g in c.gru])        # consumer: both directions' BPTT, one launch, its own stream
            xattn_branch()
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX | ops.PHASE_LSTHM_WGRAD)
            cur.wait_stream(s_g)
        else:
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD)
            xattn_branch()
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX | ops.PHASE_LSTHM_WGRAD | ops.PHASE_SPEAKER_BWD)
            for i in range(2):
                ops.gru_speaker_set_grads(c.gru[i].desc, dhq[i], dgs[i][0], dgs[i][1])
            ops.gru_speaker_bwd([g.desc for g in c.gru])              # both directions' BPTT in one launch
        for i, pre in enumerate(("marn_cell_f.", "marn_cell_b.")):
            gru_speaker_dir_bwd(c.gru[i], _sub(P, pre), _sub(G, pre), dgs[i][0], dgs[i][1], dx_l, dx_a, Ln, B, H)
        # ---- encoders (audio branch on a side stream) and linear_in
        cur.wait_stream(s_x)
        ops.add_rows(dx_l, dx_l, dxl_x)
        ops.add_rows(dx_a, dx_a, dxa_x)
        Pl, Pa, Gl, Ga = _sub(P, "encoder_l."), _sub(P, "encoder_a."), _sub(G, "encoder_l."), _sub(G, "encoder_a.")
        s_audio.wait_stream(cur)
        with torch.cuda.stream(s_audio):
            F_.encoder_layer_bwd(c.enc[2], F_.encoder_layer_bwd(c.enc[3], dx_a, Pa, Ga), Pa, Ga)
        d1 = F_.encoder_layer_bwd(c.enc[0], F_.encoder_layer_bwd(c.enc[1], dx_l, Pl, Gl), Pl, Gl)
        ops.grad_weight(d1, c.x2d[:, :d.d_r], G("linear_in.weight"))
        ops.colsum_acc(d1, G("linear_in.bias"))
        cur.wait_stream(s_audio)
