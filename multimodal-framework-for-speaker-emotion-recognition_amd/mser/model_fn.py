"""Forward / backward of the whole MARN1_sps path (reference model/lsthm_sps.py:349-394) as explicit kernel sequences.

``marn1_forward`` returns the outputs and a context object; ``marn1_backward`` consumes it.  Dropout: ``drop=None`` (eval mode,
or every p = 0) is the identity and the parity configuration (SURVEY.md 7 "Dropout"); a ``DropCfg`` draws every site from the
counter-based generator of include/mser.h, the same mask in the forward and the backward.
The two encoder branches (text / audio) and the speaker chain are independent until the LSTHM chain, so they are
enqueued on side streams (fork/join with events; capturable into one hipGraph).
"""
from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import functional as F_
from . import ops
from .functional import Layout

Tensor = torch.Tensor
Getter = Callable[[str], Optional[Tensor]]


@dataclass
class ModelDims:
    d_r: int = 1024
    d_a: int = 100
    D: int = 100
    H: int = 128
    n_head: int = 8
    d_k: int = 40
    d_v: int = 40
    n_classes: int = 6
    xattn_heads: int = 1


@dataclass
class DropCfg:
    """Dropout probabilities of one training step, site by site (defaults = the reference's constructor defaults), and the
    step's generator words {seed, step} (int32 [2] on the device)."""
    rng: Tensor = None
    p_enc_l: tuple = (0.1, 0.1, 0.1)        # encoder_l: attention (encoder.py:83), after fc (:54), after w_2 (:106)
    p_enc_a: tuple = (0.1, 0.1, 0.1)
    p_xattn: tuple = (0.2, 0.2, 0.2, 0.2)   # crossatt_l2a, crossatt_a2l, crossatt_l2a_1, crossatt_a2l_1 (lsthm_sps.py:98,:126)
    p_fc: float = 0.5                       # :318
    p_out: float = 0.5                      # :323
    p_rec: float = 0.5                      # :365, :374
    p_cell: tuple = (0.5, 0.5)              # per direction: h_q0/h_q1, h_l/h_a (:183,:188,:211,:213)
    p_cell_attn: tuple = (0.2, 0.2)         # per direction: rank-1 attention (:69)

    def site(self, sid: int, p: float):
        return F_.DropSite(self.rng, sid, float(p)) if p > 0 else None

    def any(self) -> bool:
        return any(p > 0 for p in (*self.p_enc_l, *self.p_enc_a, *self.p_xattn, self.p_fc, self.p_out, self.p_rec, *self.p_cell,
                                   *self.p_cell_attn))


@dataclass
class ModelCtx:
    dims: ModelDims = None
    L: int = 0
    B: int = 0
    x2d: Tensor = None
    xl0: Tensor = None
    enc: list = None
    x_l: Tensor = None
    x_a: Tensor = None
    rev: Tensor = None
    lens: Tensor = None
    Hcat: Tensor = None
    cell_ws: Tensor = None
    cell_dirs: list = None
    cell_desc: object = None
    pipelined: bool = False
    A1: Tensor = None
    A2: Tensor = None
    xa: list = None
    y1: Tensor = None
    y1r: Tensor = None
    y2: Tensor = None
    lp: Tensor = None
    tail: object = None
    qmask: Tensor = None
    drop: DropCfg = None
    cell_drop: tuple = None
    zbuf: Tensor = None        # zeroed accumulators of the backward's attention branches, prepared by the forward (training)


def _sub(P: Getter, prefix: str) -> Getter:
    return lambda n: P(prefix + n)


class _Streams:
    """Lazily created side streams for the independent branches."""
    _s = None

    @classmethod
    def get(cls, dev):
        if cls._s is None or cls._s[0].device != dev:
            cls._s = [torch.cuda.Stream(device=dev) for _ in range(5)]
        return cls._s


def marn1_forward(P: Getter, x: Tensor, qmask: Tensor, umask: Tensor, dims: ModelDims, use_streams: bool = True,
                  drop: Optional[DropCfg] = None, prep_backward: bool = False):
    """x [L,B,d_r+d_a] f32, qmask [L,B,2] f32, umask [B,L] f32 -> (log_probs [B*L,C], x_l [L,B,D], x_a [L,B,D], ctx).
    prep_backward: a backward over this forward will follow (the forward then zeroes that backward's accumulators on a side stream)."""
    if drop is not None and not drop.any():
        drop = None
    Ln, B, Fin = x.shape
    d = dims
    if Fin < d.d_r + d.d_a:
        raise RuntimeError(f"x has {Fin} features, model expects d_r+d_a = {d.d_r + d.d_a}")
    for t, nm in ((x, "x"), (qmask, "qmask"), (umask, "umask")):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError(f"{nm} must be a float32 GPU tensor (got {t.dtype} on {t.device})")
    x = x.contiguous()
    qmask = qmask.contiguous()
    umask = umask.contiguous()
    N, D, H = Ln * B, d.D, d.H
    lay = Layout.time_major(Ln, B)
    c = ModelCtx(dims=d, L=Ln, B=B, qmask=qmask, drop=drop)

    def enc_drops(call):            # call: 0/1 = text first/second pass, 2/3 = audio
        if drop is None:
            return None
        ps = drop.p_enc_l if call < 2 else drop.p_enc_a
        return tuple(drop.site(F_.SITE_ENC + 3 * call + i, ps[i]) for i in range(3))

    def xa_drop(i):
        return drop.site(F_.SITE_XATTN + i, drop.p_xattn[i]) if drop is not None else None
    c.x2d = x.view(N, Fin)
    cur = torch.cuda.current_stream()
    side = _Streams.get(x.device) if use_streams else None

    # ---- text branch (current stream) and audio branch (side stream 0): linear_in + 2 x EncoderLayer each
    c.xl0 = torch.empty(N, D, device=x.device)
    xa0 = c.x2d[:, d.d_r:d.d_r + d.d_a]
    c.x_l = torch.empty(N, D, device=x.device)
    c.x_a = torch.empty(N, D, device=x.device)
    c.enc = [None] * 4
    Pl, Pa = _sub(P, "encoder_l."), _sub(P, "encoder_a.")

    e1 = [None, None]

    def text_branch(stage=None):         # stage 0 / 1: first / second encoder pass alone (None: both)
        if stage in (None, 0):
            ops.linear(c.x2d[:, :d.d_r], P("linear_in.weight"), c.xl0, bias=P("linear_in.bias"))
            e1[0], c.enc[0] = F_.encoder_layer_fwd(c.xl0, None, Pl, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(0), need_attn=False)
        if stage in (None, 1):
            _, c.enc[1] = F_.encoder_layer_fwd(c.xl0, e1[0], Pl, lay, d.n_head, d.d_k, d.d_v, out=c.x_l, drops=enc_drops(1), need_attn=False)

    def audio_branch(stage=None):
        if stage in (None, 0):
            e1[1], c.enc[2] = F_.encoder_layer_fwd(xa0, None, Pa, lay, d.n_head, d.d_k, d.d_v, drops=enc_drops(2), need_attn=False)
        if stage in (None, 1):
            _, c.enc[3] = F_.encoder_layer_fwd(xa0, e1[1], Pa, lay, d.n_head, d.d_k, d.d_v, out=c.x_a, drops=enc_drops(3), need_attn=False)

    c.lens = torch.empty(B, device=x.device, dtype=torch.int32)
    c.rev = torch.empty(Ln, B, device=x.device, dtype=torch.int32)
    ops.build_reverse_index(umask, c.lens, c.rev)
    # the bidirectional MARN cell: both directions share every launch; its speaker chain depends on qmask only
    c.Hcat = torch.empty(N, 10 * H, device=x.device)
    nbytes = ops.cell_workspace_bytes(Ln, B, D, H, 2)
    c.cell_ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    c.cell_dirs = [
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_f.")), qmask=qmask, rev=None, out=c.Hcat[:, 0:4 * H]),
        dict(p=ops.cell_param_struct(_sub(P, "marn_cell_b.")), qmask=qmask, rev=c.rev, out=c.Hcat[:, 4 * H:8 * H]),
    ]
    c.cell_drop = None
    if drop is not None and (any(p_ > 0 for p_ in drop.p_cell) or any(p_ > 0 for p_ in drop.p_cell_attn)):
        c.cell_drop = (drop.rng, [F_.SITE_CELL, F_.SITE_CELL + 4], drop.p_cell, drop.p_cell_attn)
    desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 10 * H, c.cell_ws, drop=c.cell_drop)
    c.cell_desc = desc
    w, v, v1, v2 = P("w"), P("v"), P("v1"), P("v2")
    c.A1 = torch.empty(N, H, device=x.device)
    c.A2 = torch.empty(N, H, device=x.device)
    hh = d.xattn_heads
    c.xa = [None] * 4

    # sequence-level cross-modal attention (:377-383): needs only the encoder outputs, so it runs beside the LSTHM chain
    def xattn_a():      # attn1 path
        c.xa[0] = F_.xattn_fwd(c.x_l, w, c.x_a, v, P("crossatt_l2a.Wq"), P("crossatt_l2a.Wk"), P("crossatt_l2a.Wv"), lay, lay, c.A1, hh,
                               drop=xa_drop(0))
        c.xa[2] = F_.xattn_fwd(c.x_a, v, c.A1, v1, P("crossatt_l2a_1.Wq"), P("crossatt_l2a_1.Wk"), P("crossatt_l2a_1.Wv"), lay, lay,
                               c.Hcat[:, 8 * H:9 * H], hh, drop=xa_drop(2))

    def xattn_b():      # attn2 path
        c.xa[1] = F_.xattn_fwd(c.x_a, v, c.x_l, w, P("crossatt_a2l.Wq"), P("crossatt_a2l.Wk"), P("crossatt_a2l.Wv"), lay, lay, c.A2, hh,
                               drop=xa_drop(1))
        c.xa[3] = F_.xattn_fwd(c.x_l, w, c.A2, v2, P("crossatt_a2l_1.Wq"), P("crossatt_a2l_1.Wk"), P("crossatt_a2l_1.Wv"), lay, lay,
                               c.Hcat[:, 9 * H:10 * H], hh, drop=xa_drop(3))

    # Counter-linked concurrent kernels need REAL concurrency: a hipGraph executor may serialise parallel branches in an order that
    # starts the consumer first (it would spin until its bounded time-out), so under stream capture the phases are ordered instead.
    # (Two separate counter-linked launches -- PHASE_SEPARATE_SPEAKER -- let the speaker chain start under the encoders in eager
    # mode, but the fused launch measures faster even there, 3.43 against 3.52 ms per step, and carries the statistics roles.)
    pipelined = False
    sep = ops.PHASE_SEPARATE_SPEAKER if pipelined else 0
    c.pipelined = False          # backward: both BPTT chains share one fused launch, nothing to overlap by hand
    if side is not None:
        s_audio, s_spk, s_xa, s_xb = side[:4]
        for st in side[:4]:
            st.wait_stream(cur)
        ev_prep = torch.cuda.Event()
        # training: everything of the backward that only needs ZEROING (the cell's carries / accumulators / BPTT counters, the six
        # accumulators of the attention branches) is done here, beside the encoders, instead of between the head's backward and the BPTT
        c.zbuf = torch.empty(2 * N * H + 4 * N * D, device=x.device) if prep_backward else None

        LATE_PREP = True           # see below, at the chains' launch (False: the preparation beside the encoders, as in round 1)

        def prep_branch():
            with torch.cuda.stream(s_spk):
                if c.zbuf is not None:
                    c.zbuf.zero_()
                if not LATE_PREP:
                    ops.marn_cell_run(desc, ops.PHASE_FWD_PREP | (ops.PHASE_PREP_BOTH if c.zbuf is not None else 0))   # tables, initial states, counters: off the encoders' stream
                ev_prep.record(s_spk)
                if not LATE_PREP:
                    ops.marn_cell_run(desc, ops.PHASE_SPEAKER_FWD | sep)   # qmask-only chain: overlaps the encoders AND the LSTHM chain
        # (the preparation is issued FIRST although the encoders are the critical path: a hipGraph replay gave a branch forked behind
        # them a queue only after the attention branches -- 3.29 against 3.21 ms per step; it is 6 nodes, ~35 us, since its fills share one launch)
        prep_branch()
        # the two branches are issued layer by layer in alternation: the replay dispatches nodes in issue order (~6 us each), a branch
        # issued whole behind the other started 70 us late
        for stage in (0, 1):
            text_branch(stage)                                  # the longer branch (linear_in in front) first
            with torch.cuda.stream(s_audio):
                audio_branch(stage)
        if LATE_PREP:
            # the audio stream's hoisted input products follow the audio branch at once (it finishes ~50 us before the text branch, which
            # has linear_in in front): the critical path in front of the chains then holds the text stream's half of them only
            with torch.cuda.stream(s_audio):
                ops.marn_cell_run(desc, ops.PHASE_LSTHM_PRE_A)
        cur.wait_stream(s_audio)                                # x_l and x_a are final
        ev_x = torch.cuda.Event()
        ev_x.record(cur)
        if pipelined:
            cur.wait_event(ev_prep)
        else:
            cur.wait_stream(s_spk)
        # The critical chain is ISSUED first (the host needs ~10 us per launch): pipelined, the LSTHM kernel follows the speaker
        # kernel step by step through a device-side counter; both are persistent (64 + 64 workgroups) and the attention GEMMs,
        # issued afterwards on two side streams, fill the other CUs.
        # (issuing the attention branches first and the chains behind them measured the same step time: what the chains lose to the
        # branches' traffic in their first ~80 us equals what waiting for the branches would cost)
        if LATE_PREP:
            # the cell's preparation (47 MB of fills: sentinel words of the hand-off arrays, zeroed carries) runs RIGHT IN FRONT of the chains
            # although it costs ~20 us of the critical path there: done early, beside the encoders, those arrays have left the
            # infinity cache by the time the chains hand data through them, and every hand-off pays for it (cell-only measurement,
            # scratch/diag_stamps.py trash: 775 us per forward launch right behind the preparation, 917 us with 1 GB of traffic between;
            # in the step: 836 against 870 us per forward launch, the step time itself unchanged within the noise)
            # -- on the side stream, beside the hoisted input products of the LSTHM streams (50 us on this stream)
            # (round 3, tried: the text stream's products issued right behind its own encoder branch, without waiting for the audio
            #  stream's -- 53 us idle in the kernel trace --: the hipGraph executor then put the whole audio branch on the text branch's
            #  queue, BEHIND it: 2.94 against 2.84 ms per step.  Which branch gets which queue follows the fork / join shape; reverted.)
            s_spk.wait_event(ev_x)
            with torch.cuda.stream(s_spk):
                ops.marn_cell_run(desc, ops.PHASE_FWD_PREP | (ops.PHASE_PREP_BOTH if c.zbuf is not None else 0) | ops.PHASE_SPEAKER_FWD | sep)
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_PRE_L)
            cur.wait_stream(s_spk)
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD | ops.PHASE_PRE_DONE | sep)
        else:
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD | sep)
        # ONE branch for all four modules: with two branches (attn1 path / attn2 path) a hipGraph replay ran the first-level modules of
        # both on one queue and put the second-level ones on the MAIN queue, behind the chain launch (kernel trace, round 3:
        # 70 us of attention on the critical path between the chain and the head); a single chain of nodes stays on its side queue and
        # is done 150 us into the 760 us chain
        s_xa.wait_event(ev_x)
        with torch.cuda.stream(s_xa):
            xattn_a()
            xattn_b()
        cur.wait_stream(s_spk)
        cur.wait_stream(s_xa)
    else:
        ops.marn_cell_run(desc, ops.PHASE_FWD_PREP | ops.PHASE_SPEAKER_FWD)
        text_branch()
        audio_branch()
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_FWD)
        xattn_a()
        xattn_b()

    # ---- dropout_rec on the two directions' outputs (:365, :374; the backward direction is already back in time order)
    if drop is not None and drop.p_rec > 0:
        for i in range(2):
            drop.site(F_.SITE_REC + i, drop.p_rec).apply_(c.Hcat[:, 4 * H * i:4 * H * (i + 1)])
    # ---- fusion head (:390-393)
    c.y1 = torch.empty(N, D, device=x.device)
    ops.linear(c.Hcat, P("fc.0.weight"), c.y1, bias=P("fc.0.bias"), relu=True)
    if drop is not None and drop.p_fc > 0:
        drop.site(F_.SITE_FC, drop.p_fc).apply_(c.y1)
    # residual + nn_out + log_softmax: one row-tiled launch (csrc/encoder.hip tail_fwd_kernel)
    c.y1r = torch.empty(N, D, device=x.device)
    h_out = P("nn_out.0.weight").shape[0]
    c.y2 = torch.empty(N, h_out, device=x.device)
    c.lp = torch.empty(B * Ln, d.n_classes, device=x.device)
    c.tail = ops.head_tail_desc(Ln, B, c.y1, c.x_l, c.x_a, P("nn_out.0.weight"), P("nn_out.0.bias"), P("nn_out.3.weight"),
                                P("nn_out.3.bias"), c.y1r, c.y2, c.lp)
    if drop is not None and (drop.p_fc > 0 or drop.p_out > 0):
        c.tail.rng, c.tail.site_out, c.tail.p_out, c.tail.p_fc = drop.rng.data_ptr(), F_.SITE_OUT, drop.p_out, drop.p_fc
    ops.head_tail_fwd(c.tail)
    return c.lp, c.x_l.view(Ln, B, D), c.x_a.view(Ln, B, D), c


def marn1_backward(c: ModelCtx, P: Getter, G: Getter, dlp: Tensor, dx_l_out: Optional[Tensor] = None,
                   dx_a_out: Optional[Tensor] = None, use_streams: bool = True) -> None:
    """Accumulates every parameter gradient into G(name).  dlp [B*L,C]; optional grads of the returned x_l / x_a."""
    dev = dlp.device
    # The side stream costs the host an event record + wait per call (~25 us): worth it only when launches are captured into a
    # hipGraph (no host cost at replay); in eager mode the host would become the bottleneck.
    wstream = _Streams.get(dev)[4] if (use_streams and torch.cuda.is_current_stream_capturing()) else None
    # parameter gradients: bias / LayerNorm sums on their own stream (under capture), the weight-gradient GEMMs of the head, the
    # sequence-level attention modules and linear_in as ONE grouped launch at the end of the backward
    with ops.wgrad_scope(wstream, batch=True):
        _marn1_backward(c, P, G, dlp, dx_l_out, dx_a_out, use_streams)


def _marn1_backward(c, P, G, dlp, dx_l_out, dx_a_out, use_streams):
    d = c.dims
    Ln, B, N, D, H = c.L, c.B, c.L * c.B, c.dims.D, c.dims.H
    dev = dlp.device
    lay = Layout.time_major(Ln, B)
    dlp = dlp.contiguous()
    cur = torch.cuda.current_stream()
    side = _Streams.get(dev) if use_streams else None
    # ---- buffers of the backward; everything that only needs ZEROING (the attention branches' accumulators, the cell's carries
    # and step counters: ~12 tiny launches) is issued on a side stream now, so that it is not queued between the head's backward
    # and the BPTT launch on the critical path
    dx_l = torch.empty(N, D, device=dev)                      # = d(y1r), then accumulates every x_l gradient
    dx_a = torch.empty(N, D, device=dev)
    dH = torch.empty(N, 10 * H, device=dev)
    prepped = getattr(c, "zbuf", None) is not None            # the forward zeroed everything already (first backward over it only)
    zbuf = c.zbuf if prepped else torch.empty(2 * N * H + 4 * N * D, device=dev)     # ONE zero fill for the six accumulators of the attention branches
    c.zbuf = None
    dA1, dA2 = zbuf[:N * H].view(N, H), zbuf[N * H:2 * N * H].view(N, H)
    dxl_a, dxa_a, dxl_b, dxa_b = (zbuf[2 * N * H + i * N * D:2 * N * H + (i + 1) * N * D].view(N, D) for i in range(4))
    for r, pre, sl in ((c.cell_dirs[0], "marn_cell_f.", slice(0, 4 * H)), (c.cell_dirs[1], "marn_cell_b.", slice(4 * H, 8 * H))):
        r["g"] = ops.cell_param_struct(_sub(G, pre))
        r["dout"] = dH[:, sl]
    # the attention branches' partial input gradients are folded into dx_l / dx_a by the cell's BWD_DX phase (one launch)
    desc = ops.make_cell_desc(Ln, B, D, H, c.x_l, c.x_a, c.cell_dirs, 10 * H, c.cell_ws, dx_l=dx_l, dx_a=dx_a,
                              dx_l_add=(dxl_a, dxl_b), dx_a_add=(dxa_a, dxa_b), drop=c.cell_drop)
    c.cell_desc = desc
    ev_prep = None
    # (under stream capture the extra branch makes the graph executor order the attention branches behind the BPTT node: measured
    # +290 us per replay, so the zeroing stays inline there)
    if prepped:
        pass
    elif side is not None and not torch.cuda.is_current_stream_capturing():
        s_prep = side[1]
        s_prep.wait_stream(cur)
        with torch.cuda.stream(s_prep):
            zbuf.zero_()
            ops.marn_cell_run(desc, ops.PHASE_BWD_PREP)
            ev_prep = torch.cuda.Event()
            ev_prep.record(s_prep)
    else:
        zbuf.zero_()
    # ---- head: log_softmax, nn_out and the residual fan-out backward in one launch (bias gradients included)
    dy3 = torch.empty(N, d.n_classes, device=dev)
    dy2 = torch.empty_like(c.y2)
    dy1 = torch.empty(N, D, device=dev)
    t = c.tail
    t.dlp, t.dy3, t.dy2, t.dy1, t.dx_l, t.dx_a = (x_.data_ptr() for x_ in (dlp, dy3, dy2, dy1, dx_l, dx_a))
    dxl_in = dx_l_out.reshape(N, D).contiguous() if dx_l_out is not None else None
    dxa_in = dx_a_out.reshape(N, D).contiguous() if dx_a_out is not None else None
    t.dx_l_in = dxl_in.data_ptr() if dxl_in is not None else None
    t.dx_a_in = dxa_in.data_ptr() if dxa_in is not None else None
    t.g_b0, t.g_b3, t.g_bfc = G("nn_out.0.bias").data_ptr(), G("nn_out.3.bias").data_ptr(), G("fc.0.bias").data_ptr()
    ops.head_tail_bwd(t)
    ops.grad_weight(dy3, c.y2, G("nn_out.3.weight"))
    ops.grad_weight(dy2, c.y1r, G("nn_out.0.weight"))
    ops.matmul(dy1, P("fc.0.weight"), dH)
    ops.grad_weight(dy1, c.Hcat, G("fc.0.weight"))
    if c.drop is not None and c.drop.p_rec > 0:
        for i in range(2):
            c.drop.site(F_.SITE_REC + i, c.drop.p_rec).apply_(dH[:, 4 * H * i:4 * H * (i + 1)])
    # ---- the four sequence-level attention modules (two independent chains, side streams) run beside the LSTHM BPTT chain.
    # Each chain accumulates its x_l / x_a gradients into its own buffers (no cross-stream read-modify-write).
    w, v, v1, v2 = P("w"), P("v"), P("v1"), P("v2")
    # the learnable scalars w, v receive contributions from both chains: float atomics on one word each, order-insensitive

    def xb(i, name, dout, dx1, dx2, ga1, ga2):
        F_.xattn_bwd(c.xa[i], dout, P(name + ".Wq"), P(name + ".Wk"), P(name + ".Wv"), G(name + ".Wq"), G(name + ".Wk"),
                     G(name + ".Wv"), dx1, dx2, ga1, ga2)

    def xattn_a_bwd():     # attn1 path: crossatt_l2a_1(v x_a, v1 A1) <- crossatt_l2a(w x_l, v x_a)
        xb(2, "crossatt_l2a_1", dH[:, 8 * H:9 * H], dxa_a, dA1, G("v"), G("v1"))
        xb(0, "crossatt_l2a", dA1, dxl_a, dxa_a, G("w"), G("v"))

    def xattn_b_bwd():     # attn2 path: crossatt_a2l_1(w x_l, v2 A2) <- crossatt_a2l(v x_a, w x_l)
        xb(3, "crossatt_a2l_1", dH[:, 9 * H:10 * H], dxl_b, dA2, G("w"), G("v2"))
        xb(1, "crossatt_a2l", dA2, dxa_b, dxl_b, G("v"), G("w"))

    Pl, Pa, Gl, Ga = _sub(P, "encoder_l."), _sub(P, "encoder_a."), _sub(G, "encoder_l."), _sub(G, "encoder_a.")

    def text_branch(flush_stream=None, audio_stream=None):
        d2 = F_.encoder_layer_bwd(c.enc[1], dx_l, Pl, Gl)          # grad of (xl0 + e1): flows to both
        if flush_stream is not None:
            # weight gradients collected so far (the audio branch's two layers, already issued, and this layer): one grouped launch
            # on a side stream, so that it overlaps the last encoder layer's backward instead of trailing the step
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            flush_stream.wait_event(ev)
            flush_stream.wait_stream(audio_stream)
            with torch.cuda.stream(flush_stream):
                ops.wgrad_scope.flush()
        d1 = F_.encoder_layer_bwd(c.enc[0], d2, Pl, Gl)
        ops.add_rows(d1, d1, d2)                                   # d(xl0)
        ops.grad_weight(d1, c.x2d[:, :d.d_r], G("linear_in.weight"))
        ops.colsum_acc(d1, G("linear_in.bias"))

    def audio_branch():
        d2 = F_.encoder_layer_bwd(c.enc[3], dx_a, Pa, Ga)
        F_.encoder_layer_bwd(c.enc[2], d2, Pa, Ga)                 # input features need no gradient

    if side is not None:
        s_audio, s_spk, s_xa, s_xb = side[:4]
        ev_h = torch.cuda.Event()
        if ev_prep is not None:
            cur.wait_event(ev_prep)                                # zeroed accumulators, carries and step counters
            ev_h.record(cur)                                       # dH and the initial dx_l / dx_a are ready
        else:
            ev_h.record(cur)                                       # (capture: the attention branches fork BEFORE the prep nodes)
            if not prepped:
                ops.marn_cell_run(desc, ops.PHASE_BWD_PREP)
        # critical chain first (host issue order matters: ~10 us per launch)
        if c.pipelined:
            s_spk.wait_stream(cur)
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD)               # BPTT chain (persistent kernel, 64 CUs)
        if c.pipelined:
            with torch.cuda.stream(s_spk):                         # speaker BPTT follows the LSTHM BPTT step by step (counter-linked)
                ops.marn_cell_run(desc, ops.PHASE_SPEAKER_BWD)
        s_xa.wait_event(ev_h)
        s_xb.wait_event(ev_h)
        with torch.cuda.stream(s_xa):
            xattn_a_bwd()
        with torch.cuda.stream(s_xb):
            xattn_b_bwd()
        # the weight gradients of the head and of the four attention modules are complete long before the BPTT chain is: their
        # grouped launch goes out now, on a side stream, and runs on the CUs the chain leaves idle
        wst = ops.wgrad_stream()
        if wst is not None:
            # (under capture) on the scope's own stream, which is joined only at the end of the backward: the join in front of the
            # input-gradient phase below then waits for the attention chains alone, not for this 150 us grouped launch behind them
            # (same box, alternating: median step 2.845 -> 2.820 ms)
            wst.wait_stream(s_xa)
            wst.wait_stream(s_xb)
            with torch.cuda.stream(wst):
                ops.wgrad_scope.flush()
        else:
            with torch.cuda.stream(s_xa):
                s_xa.wait_stream(s_xb)
                ops.wgrad_scope.flush()
        cur.wait_stream(s_xa)
        cur.wait_stream(s_xb)
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX)            # dx_l += dg W_l + attention branches, likewise dx_a
        ev_dx = torch.cuda.Event()
        ev_dx.record(cur)
        s_audio.wait_stream(cur)
        # the encoders' backward is the critical path from here: issued FIRST (a hipGraph replay dispatches nodes in issue order, ~6 us
        # each: the four bias column sums of the cell, issued in front, delayed the first post_bwd by 17 us; kernel trace, round 3)
        # (Tried in round 3: the two branches issued layer by layer, interleaved, with the second layers' weight gradients flushed as soon
        # as both second layers are done instead of after the whole audio branch -- on paper 40-50 us less tail; as a hipGraph replay
        # 2.99 ms against 2.82: the executor's queue assignment follows the issue order and serialised the branches.)
        with torch.cuda.stream(s_audio):
            audio_branch()
        text_branch(flush_stream=s_xb, audio_stream=s_audio)
        s_xa.wait_event(ev_dx)
        if not c.pipelined:
            s_spk.wait_event(ev_dx)
            with torch.cuda.stream(s_spk):
                ops.marn_cell_run(desc, ops.PHASE_SPEAKER_BWD)     # speaker BPTT: touches only speaker-cell gradients
        with torch.cuda.stream(s_xa):
            ops.marn_cell_run(desc, ops.PHASE_LSTHM_WGRAD)         # LSTHM parameter gradients: nothing downstream reads them
        # what is left in the batch (the first text layer's and linear_in's weight gradients) has all its operands on this stream: it goes
        # out now, beside the other flush, instead of behind the joins
        ops.wgrad_scope.flush()
        for st in (s_audio, s_spk, s_xa, s_xb):
            cur.wait_stream(st)
    else:
        xattn_a_bwd()
        xattn_b_bwd()
        ops.marn_cell_run(desc, (0 if prepped else ops.PHASE_BWD_PREP) | ops.PHASE_LSTHM_BWD)
        ops.marn_cell_run(desc, ops.PHASE_LSTHM_BWD_DX | ops.PHASE_LSTHM_WGRAD | ops.PHASE_SPEAKER_BWD)
        text_branch()
        audio_branch()
