"""CPU oracle for the speaker-aware LSTHM hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch, fp32 or fp64, autograd for the
backward) of the reference algorithm on the path BASELINE.json names.  It is
the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package never
does; the product fails loudly when its HIP library is missing.

Pinning: every function here is checked against golden vectors produced by
importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).  Parity is therefore *pinned* for the
reference-native dims (D=100, H=128, d_r in {768,1024}); for widths the
reference cannot run (H != 128, multi-head cross-modal attention) this oracle is
the definition.

All parameters are passed as a flat ``dict[str, Tensor]`` whose keys are the
reference's ``state_dict`` names (``/root/reference/model/lsthm_sps.py:298-346``),
so golden weights load without translation.

Reference citations are ``path:line`` relative to the reference checkout.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# L2 ops
# --------------------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x.matmul(w.t())
    return y if b is None else y + b


def lsthm1(P: Params, pre: str, x: Tensor, ctm: Tensor, htm: Tensor, ztm: Tensor, s: Tensor):
    """LSTHM1.forward -- model/lsthm_sps.py:28-44.  Gate order f, i, o, c~."""
    g = (linear(x, P[pre + "W.weight"], P[pre + "W.bias"])
         + linear(htm, P[pre + "U.weight"], P[pre + "U.bias"])
         + linear(ztm, P[pre + "V.weight"], P[pre + "V.bias"])
         + linear(s, P[pre + "S.weight"], P[pre + "S.bias"]))
    H = ctm.shape[1]
    f = torch.sigmoid(g[:, :H])
    i = torch.sigmoid(g[:, H:2 * H])
    o = torch.sigmoid(g[:, 2 * H:3 * H])
    ch = torch.tanh(g[:, 3 * H:])
    c = f * ctm + i * ch
    h = torch.tanh(c) * o
    return c, h


def lstm_cell(P: Params, pre: str, x: Tensor, h: Tensor, c: Tensor):
    """torch.nn.LSTMCell semantics (gate order i, f, g, o) -- used at model/lsthm_sps.py:182,187."""
    g = (linear(x, P[pre + "weight_ih"], P[pre + "bias_ih"])
         + linear(h, P[pre + "weight_hh"], P[pre + "bias_hh"]))
    H = h.shape[1]
    i = torch.sigmoid(g[:, :H])
    f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    h2 = o * torch.tanh(c2)
    return h2, c2


def cross_attention(P: Params, pre: str, x1: Tensor, x2: Tensor, drop: Optional[Tensor] = None) -> Tensor:
    """CrossAttention.forward (per-step, over the FEATURE axis) -- model/lsthm_sps.py:59-72.

    As written: Q = x1 (x) Wq, K = x2 (x) Wk are [B,H,H]; softmax(Q/sqrt(dh) K, -1) x2.
    ``dh`` is the hard-coded 128 of :50 when H == 128; for other widths this oracle
    uses the feature width (the reference cannot run those).
    """
    H = x1.shape[1]
    Q = x1.unsqueeze(-1).matmul(P[pre + "Wq"])          # [B,H,H]
    K = x2.unsqueeze(-1).matmul(P[pre + "Wk"])          # [B,H,H]
    attn = F.softmax((Q / (H ** 0.5)).matmul(K), dim=-1)
    if drop is not None:                                 # :69 self.dropout(attn); ``drop`` holds the factors 0 | 1/(1-p), [B,H,H]
        attn = attn * drop
    return attn.matmul(x2.unsqueeze(-1)).squeeze(-1)


def cross_attention_seq(P: Params, pre: str, x1: Tensor, x2: Tensor, heads: int = 1, drop: Optional[Tensor] = None) -> Tensor:
    """CrossAttention2 / CrossAttention3 (over the UTTERANCE axis) -- model/lsthm_sps.py:88-101, :116-129.

    x1 [L1,B,D1], x2 [L2,B,D2] time-major -> [L1,B,Dv].  ``heads`` > 1 is the build's
    extension for BASELINE config 5 (split Dk/Dv into heads, scale 1/sqrt(Dk/heads));
    heads == 1 is the reference.
    """
    Wq, Wk, Wv = P[pre + "Wq"], P[pre + "Wk"], P[pre + "Wv"]
    a = x1.permute(1, 0, 2)
    b = x2.permute(1, 0, 2)
    Q, K, V = a.matmul(Wq), b.matmul(Wk), b.matmul(Wv)
    dk = Wq.shape[1]
    if heads == 1:
        attn = F.softmax((Q / (dk ** 0.5)).matmul(K.transpose(1, 2)), dim=-1)
        if drop is not None:                             # :98 / :126; factors [B,L1,L2]
            attn = attn * drop.reshape(attn.shape)
        return attn.matmul(V).permute(1, 0, 2)
    Bn, L1, _ = Q.shape
    L2 = K.shape[1]
    hd = dk // heads
    Qh = Q.view(Bn, L1, heads, hd).transpose(1, 2)
    Kh = K.view(Bn, L2, heads, hd).transpose(1, 2)
    Vh = V.view(Bn, L2, heads, -1).transpose(1, 2)
    attn = F.softmax((Qh / (hd ** 0.5)).matmul(Kh.transpose(2, 3)), dim=-1)
    if drop is not None:
        attn = attn * drop.reshape(attn.shape)
    return attn.matmul(Vh).transpose(1, 2).reshape(Bn, L1, -1).permute(1, 0, 2)


# --------------------------------------------------------------------------------------
# encoder (model/encoder.py)
# --------------------------------------------------------------------------------------
def sdpa(q: Tensor, k: Tensor, v: Tensor, temperature: float, mask: Optional[Tensor] = None, drop: Optional[Tensor] = None):
    """ScaledDotProductAttention.forward -- model/encoder.py:71-86."""
    attn = (q / temperature).matmul(k.transpose(2, 3))
    if mask is not None:
        attn = attn.masked_fill(mask == 0, -1e9)
    attn = F.softmax(attn, dim=-1)
    if drop is not None:                                 # :83 self.dropout(softmax); factors [B,nh,Lq,Lk]
        attn = attn * drop
    return attn.matmul(v), attn


def mha(P: Params, pre: str, q: Tensor, k: Tensor, v: Tensor, n_head: int, d_k: int, d_v: int,
        mask: Optional[Tensor] = None, drops=(None, None)):
    """MultiHeadAttention.forward -- model/encoder.py:27-60 (bias-free projections, post-LN eps 1e-6)."""
    B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    residual = q
    qh = linear(q, P[pre + "w_qs.weight"]).view(B, Lq, n_head, d_k).transpose(1, 2)
    kh = linear(k, P[pre + "w_ks.weight"]).view(B, Lk, n_head, d_k).transpose(1, 2)
    vh = linear(v, P[pre + "w_vs.weight"]).view(B, Lk, n_head, d_v).transpose(1, 2)
    if mask is not None:
        mask = mask.unsqueeze(1)
    o, attn = sdpa(qh, kh, vh, d_k ** 0.5, mask, drops[0])
    o = o.transpose(1, 2).contiguous().view(B, Lq, -1)
    o = linear(o, P[pre + "fc.weight"])
    if drops[1] is not None:                             # :54 self.dropout(self.fc(q)); factors [B,Lq,D]
        o = o * drops[1]
    o = o + residual
    o = F.layer_norm(o, (o.shape[-1],), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)
    return o, attn


def ffn(P: Params, pre: str, x: Tensor, drop: Optional[Tensor] = None) -> Tensor:
    """PositionwiseFeedForward.forward -- model/encoder.py:101-113 (``fc`` is dead)."""
    y = linear(F.relu(linear(x, P[pre + "w_1.weight"], P[pre + "w_1.bias"])),
               P[pre + "w_2.weight"], P[pre + "w_2.bias"])
    if drop is not None:                                 # :106
        y = y * drop
    y = y + x
    return F.layer_norm(y, (y.shape[-1],), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)


def encoder_layer(P: Params, pre: str, x: Tensor, n_head: int = 8, d_k: int = 40, d_v: int = 40,
                  mask: Optional[Tensor] = None, drops=(None, None, None)):
    """EncoderLayer.forward -- model/encoder.py:130-133.  ``drops``: dropout factors of (attention, fc, ffn), each optional."""
    o, attn = mha(P, pre + "slf_attn.", x, x, x, n_head, d_k, d_v, mask, drops[:2])
    return ffn(P, pre + "pos_ffn.", o, drops[2]), attn


def self_attention_lib(P: Params, pre: str, queries: Tensor, keys: Tensor, values: Tensor, h: int,
                       d_k: int, d_v: int, attention_mask: Optional[Tensor] = None,
                       attention_weights: Optional[Tensor] = None, drop: Optional[Tensor] = None) -> Tensor:
    """attention:/SelfAttention.py::ScaledDotProductAttention.forward -- :49-76 (biased projections)."""
    b, nq, nk = queries.shape[0], queries.shape[1], keys.shape[1]
    q = linear(queries, P[pre + "fc_q.weight"], P[pre + "fc_q.bias"]).view(b, nq, h, d_k).permute(0, 2, 1, 3)
    k = linear(keys, P[pre + "fc_k.weight"], P[pre + "fc_k.bias"]).view(b, nk, h, d_k).permute(0, 2, 3, 1)
    v = linear(values, P[pre + "fc_v.weight"], P[pre + "fc_v.bias"]).view(b, nk, h, d_v).permute(0, 2, 1, 3)
    att = q.matmul(k) / math.sqrt(d_k)
    if attention_weights is not None:
        att = att * attention_weights
    if attention_mask is not None:
        att = att.masked_fill(attention_mask, float("-inf"))
    att = torch.softmax(att, -1)
    if drop is not None:                                 # :72 att = self.dropout(att); factors [b,h,nq,nk]
        att = att * drop
    out = att.matmul(v).permute(0, 2, 1, 3).contiguous().view(b, nq, h * d_v)
    return linear(out, P[pre + "fc_o.weight"], P[pre + "fc_o.bias"])


# --------------------------------------------------------------------------------------
# MARN_cell: slot-table restatement of model/lsthm_sps.py:156-221 + :238-259
# --------------------------------------------------------------------------------------
def slot_tables(qmask: Tensor):
    """Per-step speaker slot tables from qmask [T,B,2] alone.

    party[t,b]  = argmax(qmask[t,b]) (ties / all-zero -> 0)            (:177)
    perm[t,r]   = dialogue whose state lands in row r of the (P0 || P1) ordering  (:242-257, :191-193)
    n0[t]       = number of party-0 dialogues at step t.
    """
    party = torch.argmax(qmask, dim=2)                       # [T,B]
    perm = torch.argsort(party, dim=1, stable=True)          # P0 rows (dialogue order), then P1 rows
    n0 = (party == 0).sum(dim=1)
    return party, perm, n0


def speaker_recurrence(P: Params, pre: str, qmask: Tensor, Hs: int, drop: Optional[Tensor] = None):
    """The qmask-only recurrence inside MARN_cell.forward (:172-207): returns h_q[t] for every step.

    Rows of the two LSTMCell states are indexed by compaction slot, not dialogue; padded slots
    are fed zeros and still advance (:249-257, :182, :187).  The blended ``q`` is applied row-wise
    in the permuted order (:204-207).
    """
    T, B, _ = qmask.shape
    dt, dev = qmask.dtype, qmask.device
    party, perm, n0 = slot_tables(qmask)
    q = torch.zeros(B, 2, Hs, dtype=dt, device=dev)
    hq0 = torch.zeros(B, Hs, dtype=dt, device=dev)
    cq0 = torch.zeros_like(hq0)
    hq1 = torch.zeros_like(hq0)
    cq1 = torch.zeros_like(hq0)
    out = []
    for t in range(T):
        N0 = int(n0[t])
        N1 = B - N0
        src = perm[t]
        h0 = q[src, party[t][src]]                                   # cat[q0_sel[:N0], q1_sel[:N1]]
        zeros = torch.zeros(B, Hs, dtype=dt, device=dev)
        if N0:
            q0_sel = torch.cat([h0[:N0], zeros[:B - N0]], 0)
            hq0, cq0 = lstm_cell(P, pre + "lstm_q0.", q0_sel, hq0, cq0)
            if drop is not None:                         # :183 dropout on the carried state; factors [T,2,B,Hs] by slot row
                hq0 = hq0 * drop[t, 0]
        if N1:
            q1_sel = torch.cat([h0[N0:], zeros[:B - N1]], 0)
            hq1, cq1 = lstm_cell(P, pre + "lstm_q1.", q1_sel, hq1, cq1)
            if drop is not None:                         # :188
                hq1 = hq1 * drop[t, 1]
        hq = torch.cat([hq0[:N0], hq1[:N1]], 0)
        m = qmask[t].unsqueeze(2)
        q = h0.unsqueeze(1) * (1 - m) + hq.unsqueeze(1) * m
        out.append(hq)
    return torch.stack(out, 0)                                          # [T,B,Hs]


def marn_cell(P: Params, pre: str, x_l: Tensor, x_a: Tensor, qmask: Tensor, H: int = 128, Hs: int = 128,
              drops: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MARN_cell.forward -- model/lsthm_sps.py:156-221.  Returns h [T,B,3H+Hs] = cat(h_l,h_a,z_l,h_q).
    ``drops`` (train mode): dropout factors "hq" [T,2,B,Hs] (:183,:188), "h" [T,2,B,H] (:211,:213), "attn" [T,B,H,H] (:69)."""
    drops = drops or {}
    T, B, _ = x_l.shape
    dt, dev = x_l.dtype, x_l.device
    hq_all = speaker_recurrence(P, pre, qmask.to(dt), Hs, drops.get("hq"))
    h_l = torch.zeros(B, H, dtype=dt, device=dev)
    h_a, c_l, c_a, z = (torch.zeros_like(h_l) for _ in range(4))
    outs = []
    for t in range(T):
        hq = hq_all[t]
        c_l, h_l = lsthm1(P, pre + "lsthm_l.", x_l[t], c_l, h_l, z, hq)     # :210
        c_a, h_a = lsthm1(P, pre + "lsthm_a.", x_a[t], c_a, h_a, z, hq)     # :212 (also z_l)
        if "h" in drops:                                                    # :211, :213 (the dropped h is the carried state)
            h_l, h_a = h_l * drops["h"][t, 0], h_a * drops["h"][t, 1]
        z = cross_attention(P, pre + "crossatt_l2a.", c_l, c_a, drops["attn"][t] if "attn" in drops else None)   # :215
        outs.append(torch.cat([h_l, h_a, z, hq], 1))
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------------------
# GRU-speaker variant: model/lsthm_onlysp.py -- the reference CLI's default model (train.py:126)
# --------------------------------------------------------------------------------------
def gru_cell(P: Params, pre: str, x: Tensor, h: Tensor) -> Tensor:
    """torch.nn.GRUCell semantics (gates r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn))) -- model/lsthm_onlysp.py:177."""
    gi = linear(x, P[pre + "weight_ih"], P[pre + "bias_ih"])
    gh = linear(h, P[pre + "weight_hh"], P[pre + "bias_hh"])
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def marn_cell_onlysp(P: Params, pre: str, x_l: Tensor, x_a: Tensor, qmask: Tensor, H: int = 128,
                     drops: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MARN_cell.forward of model/lsthm_onlysp.py:158-197: ONE speaker GRU per dialogue, fed [x_l[t] | x_a[t]] and the state of the
    party speaking at t (no slot compaction: batch-independent); its dropped output h_s is the LSTHM streams' speaker input and the
    new state of that party.  Returns [T,B,4H] = cat(h_l, h_a, z_l, h_s).
    ``drops``: "hs" [T,B,H] (:177), "h" [T,2,B,H] (:184,:186), "attn" [T,B,H,H] (:69)."""
    drops = drops or {}
    T, B, _ = x_l.shape
    dt, dev = x_l.dtype, x_l.device
    qm = qmask.to(dt)
    q = torch.zeros(B, 2, H, dtype=dt, device=dev)
    h_l = torch.zeros(B, H, dtype=dt, device=dev)
    h_a, c_l, c_a, z = (torch.zeros_like(h_l) for _ in range(4))
    rows = torch.arange(B, device=dev)
    outs = []
    for t in range(T):
        U = torch.cat([x_l[t], x_a[t]], 1)                                  # :172
        idx = torch.argmax(qm[t], 1)                                        # :175
        h_s = gru_cell(P, pre + "gru_s.", U, q[rows, idx])                  # :176-177
        if "hs" in drops:
            h_s = h_s * drops["hs"][t]
        m = qm[t].unsqueeze(2)
        q = q * (1 - m) + h_s.unsqueeze(1) * m                              # :179-181
        c_l, h_l = lsthm1(P, pre + "lsthm_l.", x_l[t], c_l, h_l, z, h_s)     # :183
        c_a, h_a = lsthm1(P, pre + "lsthm_a.", x_a[t], c_a, h_a, z, h_s)     # :185
        if "h" in drops:
            h_l, h_a = h_l * drops["h"][t, 0], h_a * drops["h"][t, 1]
        z = cross_attention(P, pre + "crossatt_l2a.", c_l, c_a, drops["attn"][t] if "attn" in drops else None)   # :188
        outs.append(torch.cat([h_l, h_a, z, h_s], 1))
    return torch.stack(outs, 0)


def marn1_onlysp_forward(P: Params, x: Tensor, qmask: Tensor, umask: Tensor, d_r: int = 1024, d_a: int = 100, H: int = 128,
                         n_head: int = 8, d_k: int = 40, d_v: int = 40, drops: Optional[Dict[str, Tensor]] = None):
    """MARN1_onlysp.forward -- model/lsthm_onlysp.py:260-300.  Against MARN1_sps: the second encoder pass takes the first pass's output
    (no residual add, :264-267), the cell is marn_cell_onlysp, and the head is nn_out (10H -> 32 -> C) directly on
    cat[h, attn1, attn2] (:287).  ``drops`` as in marn1_sps_forward, with "cell{k}.hs" instead of "cell{k}.hq" and no "fc"."""
    dr = drops or {}

    def enc_dr(k):
        return tuple(dr.get(f"enc{k}.{n}") for n in ("attn", "fc", "ffn"))

    def cell_dr(k):
        return {n: dr[f"cell{k}.{n}"] for n in ("hs", "h", "attn") if f"cell{k}.{n}" in dr}

    x_l = x[:, :, :d_r].permute(1, 0, 2)
    x_a = x[:, :, d_r:d_r + d_a].permute(1, 0, 2)
    x_l = linear(x_l, P["linear_in.weight"], P["linear_in.bias"])
    x_l, _ = encoder_layer(P, "encoder_l.", x_l, n_head, d_k, d_v, drops=enc_dr(0))
    x_a, _ = encoder_layer(P, "encoder_a.", x_a, n_head, d_k, d_v, drops=enc_dr(2))
    x_l, _ = encoder_layer(P, "encoder_l.", x_l, n_head, d_k, d_v, drops=enc_dr(1))
    x_a, _ = encoder_layer(P, "encoder_a.", x_a, n_head, d_k, d_v, drops=enc_dr(3))
    x_l = x_l.permute(1, 0, 2)
    x_a = x_a.permute(1, 0, 2)
    h_f = marn_cell_onlysp(P, "marn_cell_f.", x_l, x_a, qmask, H, cell_dr(0))
    if "rec0" in dr:
        h_f = h_f * dr["rec0"]
    h_b = marn_cell_onlysp(P, "marn_cell_b.", reverse_seq(x_l, umask), reverse_seq(x_a, umask), reverse_seq(qmask, umask), H, cell_dr(1))
    h_b = reverse_seq(h_b, umask)
    if "rec1" in dr:
        h_b = h_b * dr["rec1"]
    w, v, v1, v2 = P["w"], P["v"], P["v1"], P["v2"]
    attn1 = cross_attention_seq(P, "crossatt_l2a.", w * x_l, v * x_a, 1, dr.get("xattn0"))
    attn2 = cross_attention_seq(P, "crossatt_a2l.", v * x_a, w * x_l, 1, dr.get("xattn1"))
    attn1 = cross_attention_seq(P, "crossatt_l2a_1.", v * x_a, v1 * attn1, 1, dr.get("xattn2"))
    attn2 = cross_attention_seq(P, "crossatt_a2l_1.", w * x_l, v2 * attn2, 1, dr.get("xattn3"))
    out = F.relu(linear(torch.cat([h_f, h_b, attn1, attn2], -1), P["nn_out.0.weight"], P["nn_out.0.bias"]))
    if "out" in dr:
        out = out * dr["out"]
    out = linear(out, P["nn_out.3.weight"], P["nn_out.3.bias"])
    lp = F.log_softmax(out, 2).permute(1, 0, 2)
    return lp.reshape(-1, lp.shape[-1]), x_l, x_a


# --------------------------------------------------------------------------------------
# GRU-speaker variants with the LayerNorm'd sequence attention and the softmax-weighted fusion:
# model/lsthm_nsps.py (MARN1_nsps) and model/lsthm_no_en.py (MARN1_no_en: the same without the text encoder)
# --------------------------------------------------------------------------------------
def cross_attention_seq_ln(P: Params, pre: str, x1: Tensor, x2: Tensor, drop: Optional[Tensor] = None, eps: float = 1e-6) -> Tensor:
    """CrossAttention2 of model/lsthm_nsps.py:75-108: single-head attention across the utterance axis with dh = dk = dv (100 in
    MARN1_nsps, :287-288), then ``output += residual`` (x_1) and LayerNorm(dh, eps=1e-6).  x1, x2 [L,B,D] -> [L,B,D]."""
    Wq, Wk, Wv = P[pre + "Wq"], P[pre + "Wk"], P[pre + "Wv"]
    a, b = x1.permute(1, 0, 2), x2.permute(1, 0, 2)
    Q, K, V = a @ Wq, b @ Wk, b @ Wv                                         # :96-98
    attn = torch.softmax((Q / (Wq.shape[1] ** 0.5)) @ K.transpose(1, 2), -1)  # :100
    if drop is not None:
        attn = attn * drop                                                   # :101
    out = (attn @ V).permute(1, 0, 2) + x1                                   # :102-105
    return F.layer_norm(out, (out.shape[-1],), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], eps)


def marn_cell_nsps(P: Params, pre: str, x: Tensor, x_l: Tensor, x_a: Tensor, qmask: Tensor, H: int = 128,
                   drops: Optional[Dict[str, Tensor]] = None):
    """MARN_cell.forward of model/lsthm_nsps.py:158-216.  Against marn_cell_onlysp: the speaker GRU is fed ``x[t]`` (the caller's
    cat[linear_in(text) | audio], i.e. the PRE-encoder features, :305,:177), the new party states are
    q[b,p] = ql_0[b] (1 - m[b,p]) + h_s[b] m[b,p] with ql_0 = the state of the party NOT speaking (:183-191; ``gru_l`` is
    constructed but its update is commented out, h_l_ = ql_0) -- for a one-hot qmask row that is the onlysp update, for a padded
    (all-zero) row both party states become the listener's -- and the output rows are cat(h_l, h_a, z_l) (:202).
    Returns (h [T,B,3H], h_l, h_a, h_sp, h_li [T,B,H]).  ``drops``: "hs" [T,B,H] (:182), "h" [T,2,B,H] (:195,:197), "attn" [T,B,H,H]."""
    drops = drops or {}
    T, B, _ = x_l.shape
    dt, dev = x_l.dtype, x_l.device
    qm = qmask.to(dt)
    q = torch.zeros(B, 2, H, dtype=dt, device=dev)
    h_l = torch.zeros(B, H, dtype=dt, device=dev)
    h_a, c_l, c_a, z = (torch.zeros_like(h_l) for _ in range(4))
    rows = torch.arange(B, device=dev)
    outs, hl, ha, hsp, hli = [], [], [], [], []
    for t in range(T):
        idx = torch.argmax(qm[t], 1)                                        # :178
        qs_0, ql_0 = q[rows, idx], q[rows, 1 - idx]                         # :180, :231-239
        h_s = gru_cell(P, pre + "gru_s.", x[t], qs_0)                       # :182
        if "hs" in drops:
            h_s = h_s * drops["hs"][t]
        m = qm[t].unsqueeze(2)
        q = ql_0.unsqueeze(1) * (1 - m) + h_s.unsqueeze(1) * m              # :188-191
        c_l, h_l = lsthm1(P, pre + "lsthm_l.", x_l[t], c_l, h_l, z, h_s)     # :194
        c_a, h_a = lsthm1(P, pre + "lsthm_a.", x_a[t], c_a, h_a, z, h_s)     # :196
        if "h" in drops:
            h_l, h_a = h_l * drops["h"][t, 0], h_a * drops["h"][t, 1]
        z = cross_attention(P, pre + "crossatt_l2a.", c_l, c_a, drops["attn"][t] if "attn" in drops else None)   # :199
        outs.append(torch.cat([h_l, h_a, z], 1))
        hl.append(h_l); ha.append(h_a); hsp.append(h_s); hli.append(ql_0)
    return tuple(torch.stack(v, 0) for v in (outs, hl, ha, hsp, hli))


def marn1_nsps_forward(P: Params, x: Tensor, qmask: Tensor, umask: Tensor, d_r: int = 1024, d_a: int = 100, H: int = 128,
                       n_head: int = 8, d_k: int = 40, d_v: int = 40, no_en: bool = False,
                       drops: Optional[Dict[str, Tensor]] = None):
    """MARN1_nsps.forward -- model/lsthm_nsps.py:300-360; ``no_en=True``: MARN1_no_en.forward -- model/lsthm_no_en.py:300-360 (the
    text stream skips its encoder, :306,:309).  Encoder passes as in MARN1_sps (second pass on x + first pass); the cells are
    marn_cell_nsps on x = cat[linear_in(text), audio]; only h_l and h_a of the cells reach the head (h, h_sp are computed and never
    used, :335-339); CrossAttention2 is the LayerNorm'd form on the unscaled encoder outputs (:341-342); the fusion weights are
    softmax(p) (:347-348); head = nn_out(cat[w1 [h_l | attn2], w2 [h_a | attn1]] + fc(x_l)) (:350-355; fc2(x_a) is computed and
    never used).  ``drops``: "enc{k}.*" as in marn1_sps_forward ("enc0"/"enc1" absent with no_en), "xattn0"/"xattn1",
    "cell{k}.{hs,h,attn}", "rec{k}.{l,a}" [L,B,H] (dropout_rec on hf_l, hf_a / hb_l, hb_a, :317-318,:330-331), "fc" [L,B,712], "out"."""
    dr = drops or {}

    def enc_dr(k):
        return tuple(dr.get(f"enc{k}.{n}") for n in ("attn", "fc", "ffn"))

    def cell_dr(k):
        return {n: dr[f"cell{k}.{n}"] for n in ("hs", "h", "attn") if f"cell{k}.{n}" in dr}

    x_l = x[:, :, :d_r].permute(1, 0, 2)
    x_a = x[:, :, d_r:d_r + d_a].permute(1, 0, 2)
    x_l = linear(x_l, P["linear_in.weight"], P["linear_in.bias"])
    xin = torch.cat([x_l, x_a], 2).permute(1, 0, 2)                           # :305  [L,B,2D]
    if not no_en:
        x_l_1, _ = encoder_layer(P, "encoder_l.", x_l, n_head, d_k, d_v, drops=enc_dr(0))
    x_a_1, _ = encoder_layer(P, "encoder_a.", x_a, n_head, d_k, d_v, drops=enc_dr(2))
    if not no_en:
        x_l, _ = encoder_layer(P, "encoder_l.", x_l + x_l_1, n_head, d_k, d_v, drops=enc_dr(1))
    x_a, _ = encoder_layer(P, "encoder_a.", x_a + x_a_1, n_head, d_k, d_v, drops=enc_dr(3))
    x_l = x_l.permute(1, 0, 2)
    x_a = x_a.permute(1, 0, 2)
    _, hf_l, hf_a, _, _ = marn_cell_nsps(P, "marn_cell_f.", xin, x_l, x_a, qmask, H, cell_dr(0))
    _, hb_l, hb_a, _, _ = marn_cell_nsps(P, "marn_cell_b.", reverse_seq(xin, umask), reverse_seq(x_l, umask), reverse_seq(x_a, umask),
                                         reverse_seq(qmask, umask), H, cell_dr(1))
    hb_l, hb_a = reverse_seq(hb_l, umask), reverse_seq(hb_a, umask)
    if "rec0.l" in dr:
        hf_l, hf_a, hb_l, hb_a = hf_l * dr["rec0.l"], hf_a * dr["rec0.a"], hb_l * dr["rec1.l"], hb_a * dr["rec1.a"]
    attn1 = cross_attention_seq_ln(P, "crossatt_l2a.", x_l, x_a, dr.get("xattn0"))
    attn2 = cross_attention_seq_ln(P, "crossatt_a2l.", x_a, x_l, dr.get("xattn1"))
    wsm = torch.softmax(P["p"], 0)                                            # :347-348
    resid = F.relu(linear(x_l, P["fc.0.weight"], P["fc.0.bias"]))             # :350
    if "fc" in dr:
        resid = resid * dr["fc"]
    l = torch.cat([hf_l, hb_l, attn2], 2)                                     # :352
    a = torch.cat([hf_a, hb_a, attn1], 2)                                     # :353
    out = torch.cat([wsm[0] * l, wsm[1] * a], -1) + resid                     # :355
    out = F.relu(linear(out, P["nn_out.0.weight"], P["nn_out.0.bias"]))
    if "out" in dr:
        out = out * dr["out"]
    out = linear(out, P["nn_out.3.weight"], P["nn_out.3.bias"])
    lp = F.log_softmax(out, 2).permute(1, 0, 2)
    return lp.reshape(-1, lp.shape[-1]), x_l, x_a


def reverse_seq(X: Tensor, umask: Tensor) -> Tensor:
    """MARN1_sps._reverse_seq -- model/lsthm_sps.py:396-409 (flip the first len_b steps, zero-pad)."""
    L, B = X.shape[0], X.shape[1]
    lens = umask.sum(1).to(torch.int64)
    Lmax = int(lens.max())
    t = torch.arange(Lmax, device=X.device).unsqueeze(1)                   # [Lmax,1]
    src = lens.unsqueeze(0) - 1 - t                                        # [Lmax,B]
    valid = src >= 0
    src = src.clamp(min=0)
    g = X[src, torch.arange(B, device=X.device).unsqueeze(0)]              # [Lmax,B,...]
    return g * valid.view(Lmax, B, *([1] * (X.dim() - 2))).to(X.dtype)


def marn1_sps_forward(P: Params, x: Tensor, qmask: Tensor, umask: Tensor, d_r: int = 1024, d_a: int = 100,
                      H: int = 128, n_head: int = 8, d_k: int = 40, d_v: int = 40, xattn_heads: int = 1,
                      return_intermediates: bool = False, drops: Optional[Dict[str, Tensor]] = None):
    """MARN1_sps.forward -- model/lsthm_sps.py:349-394.  ``drops`` = None: eval mode (every Dropout is the identity).  Train mode
    is restated with the dropout factors (0 | 1/(1-p)) as explicit inputs, keyed by site: "enc{0..3}.{attn,fc,ffn}" (text first /
    second pass, audio first / second pass; [B,nh,L,L], [B,L,D], [B,L,D]), "xattn{0..3}" (crossatt_l2a, crossatt_a2l,
    crossatt_l2a_1, crossatt_a2l_1; [B,L,L]), "rec0"/"rec1" ([L,B,4H], :365/:374), "fc" ([L,B,100], :318), "out" ([L,B,32], :323)
    and "cell{0,1}.{hq,h,attn}" (see marn_cell).  Missing keys are identities."""
    dr = drops or {}

    def enc_dr(k):
        return tuple(dr.get(f"enc{k}.{n}") for n in ("attn", "fc", "ffn"))

    def cell_dr(k):
        return {n: dr[f"cell{k}.{n}"] for n in ("hq", "h", "attn") if f"cell{k}.{n}" in dr}
    x_l = x[:, :, :d_r].permute(1, 0, 2)
    x_a = x[:, :, d_r:d_r + d_a].permute(1, 0, 2)
    x_l = linear(x_l, P["linear_in.weight"], P["linear_in.bias"])
    x_l_1, _ = encoder_layer(P, "encoder_l.", x_l, n_head, d_k, d_v, drops=enc_dr(0))
    x_a_1, _ = encoder_layer(P, "encoder_a.", x_a, n_head, d_k, d_v, drops=enc_dr(2))
    x_l, _ = encoder_layer(P, "encoder_l.", x_l + x_l_1, n_head, d_k, d_v, drops=enc_dr(1))
    x_a, _ = encoder_layer(P, "encoder_a.", x_a + x_a_1, n_head, d_k, d_v, drops=enc_dr(3))
    x_l = x_l.permute(1, 0, 2)
    x_a = x_a.permute(1, 0, 2)

    h_f = marn_cell(P, "marn_cell_f.", x_l, x_a, qmask, H, H, cell_dr(0))
    if "rec0" in dr:
        h_f = h_f * dr["rec0"]                                                # :365
    rev_x_l = reverse_seq(x_l, umask)
    rev_x_a = reverse_seq(x_a, umask)
    rev_qmask = reverse_seq(qmask, umask)
    h_b_raw = marn_cell(P, "marn_cell_b.", rev_x_l, rev_x_a, rev_qmask, H, H, cell_dr(1))
    h_b = reverse_seq(h_b_raw, umask)
    if "rec1" in dr:
        h_b = h_b * dr["rec1"]                                                # :374
    h = torch.cat([h_f, h_b], -1)

    w, v, v1, v2 = P["w"], P["v"], P["v1"], P["v2"]
    attn1 = cross_attention_seq(P, "crossatt_l2a.", w * x_l, v * x_a, xattn_heads, dr.get("xattn0"))
    attn2 = cross_attention_seq(P, "crossatt_a2l.", v * x_a, w * x_l, xattn_heads, dr.get("xattn1"))
    attn1 = cross_attention_seq(P, "crossatt_l2a_1.", v * x_a, v1 * attn1, xattn_heads, dr.get("xattn2"))
    attn2 = cross_attention_seq(P, "crossatt_a2l_1.", w * x_l, v2 * attn2, xattn_heads, dr.get("xattn3"))

    out = F.relu(linear(torch.cat([h, attn1, attn2], -1), P["fc.0.weight"], P["fc.0.bias"]))
    if "fc" in dr:
        out = out * dr["fc"]                                                  # :318
    out = out + x_l + x_a
    out = F.relu(linear(out, P["nn_out.0.weight"], P["nn_out.0.bias"]))
    if "out" in dr:
        out = out * dr["out"]                                                 # :323
    out = linear(out, P["nn_out.3.weight"], P["nn_out.3.bias"])
    lp = F.log_softmax(out, 2).permute(1, 0, 2)
    lp = lp.reshape(-1, lp.shape[-1])
    if return_intermediates:
        return lp, x_l, x_a, dict(h_f=h_f, h_b=h_b, attn1=attn1, attn2=attn2)
    return lp, x_l, x_a


def masked_nll(pred: Tensor, target: Tensor, mask: Tensor) -> Tensor:
    """MaskedLoss.forward with NLLLoss(sum), weight=None -- loss.py:13-21."""
    m = mask.reshape(-1, 1)
    return F.nll_loss(pred * m, target, reduction="sum") / mask.sum()


def masked_loss(pred: Tensor, target: Tensor, mask: Tensor, weight: Optional[Tensor] = None, is_ce: bool = False) -> Tensor:
    """MaskedLoss.forward in full -- loss.py:13-25: losser(weight, reduction='sum')(pred * mask, target) divided by sum(mask)
    or, with class weights, by sum(weight[target] * mask).  Written out instead of calling the torch losses: the
    CrossEntropyLoss branch re-applies log_softmax to pred * mask (identity on the log-probabilities of a valid row; a masked
    row becomes log_softmax(0) = -log C and contributes weight[y] * log C)."""
    m = mask.reshape(-1, 1)
    z = pred * m
    if is_ce:
        z = z - torch.logsumexp(z, dim=1, keepdim=True)
    w = weight[target] if weight is not None else torch.ones_like(z[:, 0])
    num = -(w * z.gather(1, target.view(-1, 1)).squeeze(1)).sum()
    den = (w * m.squeeze(1)).sum() if weight is not None else mask.sum()
    return num / den


# --------------------------------------------------------------------------------------
# DialogueRNN BiModel (BASELINE configs[3]): model/DialogueRNN.py:24-77 (MatchingAttention), :80-166
# (DialogueRNNCell), :169-198 (DialogueRNN), :201-277 (BiModel); constructed by model_trainer.py:35-47 with D_m 712, D_g = D_p = 500,
# D_e = D_h = 300, listener_state=True, context_attention='general', dropout_rec = dropout = 0.1
# --------------------------------------------------------------------------------------
def dialogue_rnn(P: Params, pre: str, U: Tensor, qmask: Tensor, drops: Optional[Dict[str, Tensor]] = None):
    """DialogueRNN.forward (:183-198) with DialogueRNNCell.forward (:119-166), listener_state=True, 'general' MatchingAttention over
    the growing history of global states.  U [T,B,D_m], qmask [T,B,2] -> (e [T,B,D_e], alphas: list of [B,t] for t = 1..T-1).
    Per step: g = dropout(GRU_g([U_t | q[b,s_b]], g_{t-1}));  c = sum_s softmax_s(<W_att U_t, g_s>) g_s over s < t (zeros at t = 0);
    qs = dropout(GRU_p([U_t | c], q[b,p])) for both parties;  ql = dropout(GRU_l([U_t | qs[b,s_b]], q[b,p]));
    q = ql (1 - qmask) + qs qmask;  e = dropout(GRU_e(q[b,s_b], e_{t-1})).
    ``drops`` (factors): "g" [T,B,D_g], "qs" / "ql" [T,B,2,D_p], "e" [T,B,D_e]."""
    dr = drops or {}
    T, B, _ = U.shape
    c_pre = pre + "dialogue_cell."
    Dg = P[c_pre + "g_cell.weight_hh"].shape[1]
    Dp = P[c_pre + "p_cell.weight_hh"].shape[1]
    De = P[c_pre + "e_cell.weight_hh"].shape[1]
    dt, dev = U.dtype, U.device
    qm = qmask.to(dt)
    q = torch.zeros(B, 2, Dp, dtype=dt, device=dev)
    g_prev = torch.zeros(B, Dg, dtype=dt, device=dev)
    e_prev = torch.zeros(B, De, dtype=dt, device=dev)
    rows = torch.arange(B, device=dev)
    ghist, es, alphas = [], [], []
    Watt = P[c_pre + "attention.transform.weight"]
    for t in range(T):
        u = U[t]
        idx = torch.argmax(qm[t], 1)                                          # :129
        q0_sel = q[rows, idx]                                                 # :131
        g = gru_cell(P, c_pre + "g_cell.", torch.cat([u, q0_sel], 1), g_prev)  # :133-135
        if "g" in dr:
            g = g * dr["g"][t]
        if t == 0:
            c = torch.zeros(B, Dg, dtype=dt, device=dev)                      # :137-139
        else:
            M = torch.stack(ghist, 0)                                         # [t,B,Dg]
            x_ = linear(u, Watt)                                              # :58
            alpha = torch.softmax(torch.einsum("bd,tbd->bt", x_, M), 1)       # :59
            c = torch.einsum("bt,tbd->bd", alpha, M)                          # :75
            alphas.append(alpha)
        uc = torch.cat([u, c], 1).unsqueeze(1).expand(-1, 2, -1).reshape(B * 2, -1)          # :144
        qs = gru_cell(P, c_pre + "p_cell.", uc, q.reshape(B * 2, Dp)).view(B, 2, Dp)          # :145
        if "qs" in dr:
            qs = qs * dr["qs"][t]
        ss = qs[rows, idx].unsqueeze(1).expand(-1, 2, -1).reshape(B * 2, Dp)                  # :150
        u2 = u.unsqueeze(1).expand(-1, 2, -1).reshape(B * 2, -1)
        ql = gru_cell(P, c_pre + "l_cell.", torch.cat([u2, ss], 1), q.reshape(B * 2, Dp)).view(B, 2, Dp)   # :151-152
        if "ql" in dr:
            ql = ql * dr["ql"][t]
        m = qm[t].unsqueeze(2)
        q = ql * (1 - m) + qs * m                                             # :157
        e = gru_cell(P, c_pre + "e_cell.", q[rows, idx], e_prev)              # :160
        if "e" in dr:
            e = e * dr["e"][t]
        ghist.append(g)
        g_prev, e_prev = g, e
        es.append(e)
    return torch.stack(es, 0), alphas


def matching_attention_general2(P: Params, pre: str, M: Tensor, x: Tensor, mask: Tensor):
    """MatchingAttention(att_type='general2').forward (:61-68,:75): M [S,B,D], x [B,D], mask [B,S] ->
    alpha_ = softmax_s(<W x + b, M_s> * mask_s); alpha = alpha_ mask / sum_s(alpha_ mask); pool = sum_s alpha_s M_s."""
    x_ = linear(x, P[pre + "transform.weight"], P[pre + "transform.bias"])
    sc = torch.einsum("bd,sbd->bs", x_, M) * mask
    a_ = torch.softmax(sc, 1) * mask
    alpha = a_ / a_.sum(1, keepdim=True)
    return torch.einsum("bs,sbd->bd", alpha, M), alpha


def bimodel_forward(P: Params, U: Tensor, qmask: Tensor, umask: Tensor, drops: Optional[Dict[str, Tensor]] = None):
    """BiModel.forward(U, qmask, umask, att2=True) (:236-277): forward DialogueRNN, reversed DialogueRNN (``_reverse_seq``), the
    'general2' matching attention of every position over all positions of the concatenated emotion states, linear + ReLU, smax_fc,
    log_softmax.  Returns (log_prob [L,B,C], alpha [L][B,L], alpha_f, alpha_b).
    ``drops``: "f.*" / "b.*" (dialogue_rnn keys per direction), "rec_f" / "rec_b" [L,B,D_e] (:245,:252), "hidden" [L,B,2 D_h] (:268)."""
    dr = drops or {}

    def sub(k):
        return {n[len(k) + 1:]: v for n, v in dr.items() if n.startswith(k + ".")}
    e_f, a_f = dialogue_rnn(P, "dialog_rnn_f.", U, qmask, sub("f"))
    if "rec_f" in dr:
        e_f = e_f * dr["rec_f"]
    e_b, a_b = dialogue_rnn(P, "dialog_rnn_r.", reverse_seq(U, umask), reverse_seq(qmask, umask), sub("b"))
    e_b = reverse_seq(e_b, umask)
    if "rec_b" in dr:
        e_b = e_b * dr["rec_b"]
    em = torch.cat([e_f, e_b], -1)
    att, alpha = [], []
    for t in range(em.shape[0]):
        pool, a = matching_attention_general2(P, "matchatt.", em, em[t], umask)       # :259
        att.append(pool)
        alpha.append(a)
    att = torch.stack(att, 0)
    hidden = F.relu(linear(att, P["linear.weight"], P["linear.bias"]))                # :264
    if "hidden" in dr:
        hidden = hidden * dr["hidden"]
    lp = F.log_softmax(linear(hidden, P["smax_fc.weight"], P["smax_fc.bias"]), 2)     # :269
    return lp, alpha, a_f, a_b


def bimodel_param_shapes(D_m: int = 712, D_g: int = 500, D_p: int = 500, D_e: int = 300, D_h: int = 300,
                         n_classes: int = 6) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape of BiModel(listener_state=True, context_attention='general') in registration order."""
    S: Dict[str, Tuple[int, ...]] = {}
    for d in ("dialog_rnn_f.", "dialog_rnn_r."):
        c = d + "dialogue_cell."
        for cell, din, dh in (("g_cell.", D_m + D_p, D_g), ("p_cell.", D_m + D_g, D_p), ("e_cell.", D_p, D_e), ("l_cell.", D_m + D_p, D_p)):
            S[c + cell + "weight_ih"] = (3 * dh, din)
            S[c + cell + "weight_hh"] = (3 * dh, dh)
            S[c + cell + "bias_ih"] = (3 * dh,)
            S[c + cell + "bias_hh"] = (3 * dh,)
        S[c + "attention.transform.weight"] = (D_g, D_m)
    S["linear.weight"] = (2 * D_h, 2 * D_e)
    S["linear.bias"] = (2 * D_h,)
    S["smax_fc.weight"] = (n_classes, 2 * D_h)
    S["smax_fc.bias"] = (n_classes,)
    S["matchatt.transform.weight"] = (2 * D_e, 2 * D_e)
    S["matchatt.transform.bias"] = (2 * D_e,)
    return S


def bimodel_seeded_params(seed: int = 0, dtype=torch.float32, **dims) -> Params:
    """Deterministic BiModel parameters (numpy RandomState per name): U(-1/sqrt(fan_in), +) like nn.Linear / nn.GRUCell; the two
    attention transforms are scaled up (x3) so that the softmaxes over the history are far from uniform and the tests bite."""
    import zlib

    import numpy as np

    P: Params = {}
    for name, shp in bimodel_param_shapes(**dims).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 104729 * seed) % (2 ** 31))
        if len(shp) == 2:
            k = 1.0 / math.sqrt(shp[1])
            if "transform" in name:
                k *= 3.0
            a = rs.uniform(-k, k, shp)
        else:
            a = rs.uniform(-0.05, 0.05, shp)
        P[name] = torch.tensor(a, dtype=dtype)
    return P


def bimodel_seeded_batch(B: int, L: int, D_m: int = 712, seed: int = 1, ragged: bool = False, n_classes: int = 6):
    """U [L,B,D_m], qmask [L,B,2], umask [B,L], label [B,L] (model_trainer_d.py:62-63: U = cat(textf, acouf, visuf))."""
    x, qmask, umask, label = seeded_batch(B, L, d_r=D_m, d_a=0, seed=seed, ragged=ragged, n_classes=n_classes)
    return x, qmask, umask, label


# --------------------------------------------------------------------------------------
# optimiser (model_trainer.py:82-83, :92)
# --------------------------------------------------------------------------------------
def step_lr(lr0: float, gamma: float, step_size: int, epoch: int) -> float:
    """Closed form of scheduler.step(epoch-1) -- model_trainer.py:92."""
    return lr0 * gamma ** ((epoch - 1) // step_size)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float = 2e-5,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor update with L2-coupled weight decay (model_trainer.py:82)."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# deterministic parameter / input generators shared by tests, golden maker, bench
# --------------------------------------------------------------------------------------
def param_shapes(n_classes: int = 6, d_r: int = 1024, D: int = 100, H: int = 128, n_head: int = 8,
                 d_k: int = 40, d_v: int = 40, d_inner: int = 40, h_out: int = 32, variant: str = "sps") -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape, in the reference's registration order.  variant "onlysp": MARN1_onlysp
    (model/lsthm_onlysp.py:209-258): every cell also owns ``gru_s`` = GRUCell(2D, H); the head is ``nn_out`` on the 10H-wide
    concatenation (plus a dead ``linear``), there is no ``fc``.  variant "nsps": MARN1_nsps / MARN1_no_en
    (model/lsthm_nsps.py:242-298): ``p`` [2] instead of w, v, v1, v2; cells own gru_s and gru_l and no LSTM cells; fc / fc2 =
    Linear(D, 712); nn_out on 712; CrossAttention2 [D,D] with a LayerNorm; no crossatt_*_1."""
    S: Dict[str, Tuple[int, ...]] = {}
    nsps = variant == "nsps"
    if nsps:
        S["p"] = (2,)
    else:
        for k in ("w", "v", "v1", "v2"):
            S[k] = (1,)
    S["linear_in.weight"] = (D, d_r)
    S["linear_in.bias"] = (D,)
    for cell in ("marn_cell_f.", "marn_cell_b."):
        for ca in ("crossatt_l2a.", "crossatt_a2l."):
            for wn in ("Wq", "Wk", "Wv"):
                S[cell + ca + wn] = (1, H)
        for st in ("lsthm_l.", "lsthm_a."):
            for nm, k in (("W", D), ("U", H), ("V", H), ("S", H)):
                S[cell + st + nm + ".weight"] = (4 * H, k)
                S[cell + st + nm + ".bias"] = (4 * H,)
        for lc in (("gru_s.", "gru_l.") if nsps else ("lstm_q0.", "lstm_q1.", "gru_s.", "lstm_s.")):
            if lc.startswith("gru_"):                # model/lsthm_onlysp.py:152 registers gru_s between lstm_q1 and lstm_s
                if variant in ("onlysp", "nsps"):
                    S[cell + lc + "weight_ih"] = (3 * H, 2 * D)
                    S[cell + lc + "weight_hh"] = (3 * H, H)
                    S[cell + lc + "bias_ih"] = (3 * H,)
                    S[cell + lc + "bias_hh"] = (3 * H,)
                continue
            S[cell + lc + "weight_ih"] = (4 * H, H)
            S[cell + lc + "weight_hh"] = (4 * H, H)
            S[cell + lc + "bias_ih"] = (4 * H,)
            S[cell + lc + "bias_hh"] = (4 * H,)
    final = 2 * (2 * H + D)
    if variant == "onlysp":
        S["linear.weight"] = (h_out, 10 * H)
        S["linear.bias"] = (h_out,)
    elif nsps:
        for f in ("fc.0.", "fc2.0."):
            S[f + "weight"] = (final, D)
            S[f + "bias"] = (final,)
    else:
        S["fc.0.weight"] = (D, 8 * H + 2 * H)
        S["fc.0.bias"] = (D,)
    S["nn_out.0.weight"] = (h_out, 10 * H if variant == "onlysp" else (final if nsps else D))
    S["nn_out.0.bias"] = (h_out,)
    S["nn_out.3.weight"] = (n_classes, h_out)
    S["nn_out.3.bias"] = (n_classes,)
    for enc in ("encoder_l.", "encoder_a."):
        S[enc + "slf_attn.w_qs.weight"] = (n_head * d_k, D)
        S[enc + "slf_attn.w_ks.weight"] = (n_head * d_k, D)
        S[enc + "slf_attn.w_vs.weight"] = (n_head * d_v, D)
        S[enc + "slf_attn.fc.weight"] = (D, n_head * d_v)
        S[enc + "slf_attn.layer_norm.weight"] = (D,)
        S[enc + "slf_attn.layer_norm.bias"] = (D,)
        S[enc + "pos_ffn.w_1.weight"] = (d_inner, D)
        S[enc + "pos_ffn.w_1.bias"] = (d_inner,)
        S[enc + "pos_ffn.w_2.weight"] = (D, d_inner)
        S[enc + "pos_ffn.w_2.bias"] = (D,)
        S[enc + "pos_ffn.layer_norm.weight"] = (D,)
        S[enc + "pos_ffn.layer_norm.bias"] = (D,)
        S[enc + "pos_ffn.fc.weight"] = (100, D)
        S[enc + "pos_ffn.fc.bias"] = (100,)
    for ca in ("crossatt_l2a.", "crossatt_a2l."):
        for wn in ("Wq", "Wk", "Wv"):
            S[ca + wn] = (D, D if nsps else H)
        if nsps:
            S[ca + "layer_norm.weight"] = (D,)
            S[ca + "layer_norm.bias"] = (D,)
    if not nsps:
        for ca in ("crossatt_l2a_1.", "crossatt_a2l_1."):
            S[ca + "Wq"] = (D, H)
            S[ca + "Wk"] = (H, H)
            S[ca + "Wv"] = (H, H)
    return S


def seeded_params(seed: int = 0, dtype=torch.float32, **dims) -> Params:
    """Deterministic, platform-independent parameters keyed by name (numpy RandomState stream).

    Scales follow the reference's initialisers in spirit (U(-1/sqrt(fan_in), +) for Linear/LSTMCell)
    but the attention matrices, which the reference initialises to ones (model/lsthm_sps.py:53-55,
    82-84, 110-112), are drawn N(0, s^2) so that every softmax is non-uniform and the tests bite.
    """
    import zlib

    import numpy as np

    P: Params = {}
    for name, shp in param_shapes(**dims).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if name in ("w", "v", "v1", "v2"):
            a = 1.0 + 0.1 * rs.standard_normal(shp)
        elif name == "p":
            a = 1.0 + 0.5 * rs.standard_normal(shp)
        elif name.endswith("layer_norm.weight"):
            a = 1.0 + 0.1 * rs.standard_normal(shp)
        elif name.endswith("layer_norm.bias"):
            a = 0.1 * rs.standard_normal(shp)
        elif ".crossatt_" in name:                       # per-step rank-1 attention vectors [1,H]
            a = 0.5 * rs.standard_normal(shp)
        elif name.startswith("crossatt_"):               # sequence-level attention matrices
            a = rs.standard_normal(shp) * (0.6 / math.sqrt(shp[0]))
        else:
            fan_in = shp[1] if len(shp) == 2 else None
            if fan_in is None:                           # bias: use the matching weight's fan-in scale
                a = rs.uniform(-0.08, 0.08, shp)
            else:
                k = 1.0 / math.sqrt(fan_in)
                a = rs.uniform(-k, k, shp)
        P[name] = torch.tensor(a, dtype=dtype)
    return P


DEFAULT_DROPOUT = dict(enc=0.1, xattn=0.2, fc=0.5, out=0.5, rec=0.5, cell=0.5, cell_attn=0.2)   # the reference's constructor defaults


def seeded_drops(B: int, L: int, H: int = 128, seed: int = 0, D: int = 100, n_head: int = 8, F_out: int = 32,
                 p: Optional[Dict[str, float]] = None) -> Dict[str, Tensor]:
    """Deterministic dropout factors (0 | 1/(1-p)) for every site of marn1_sps_forward, keyed and shaped as its ``drops`` argument
    expects; one numpy RandomState stream per key (platform independent)."""
    import zlib

    import numpy as np

    p = {**DEFAULT_DROPOUT, **(p or {})}
    shapes = {}
    for k in range(4):
        shapes[f"enc{k}.attn"] = ((B, n_head, L, L), p["enc"])
        shapes[f"enc{k}.fc"] = ((B, L, D), p["enc"])
        shapes[f"enc{k}.ffn"] = ((B, L, D), p["enc"])
        shapes[f"xattn{k}"] = ((B, L, L), p["xattn"])
    shapes["fc"] = ((L, B, D), p["fc"])
    shapes["out"] = ((L, B, F_out), p["out"])
    for k in range(2):
        shapes[f"rec{k}"] = ((L, B, 4 * H), p["rec"])
        shapes[f"cell{k}.hq"] = ((L, 2, B, H), p["cell"])
        shapes[f"cell{k}.h"] = ((L, 2, B, H), p["cell"])
        shapes[f"cell{k}.attn"] = ((L, B, H, H), p["cell_attn"])
    out = {}
    for key, (shp, pk) in shapes.items():
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        keep = rs.random_sample(shp) >= pk
        out[key] = torch.tensor(keep.astype(np.float32) / np.float32(1.0 - pk))
    return out


def seeded_batch(B: int, L: int, d_r: int = 1024, d_a: int = 100, seed: int = 1, ragged: bool = False,
                 n_classes: int = 6, dtype=torch.float32):
    """Synthetic batch in the reference's layout: x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L], label [B,L]."""
    import numpy as np

    rs = np.random.RandomState(seed)
    x = rs.standard_normal((L, B, d_r + d_a)).astype(np.float32)
    spk = rs.randint(0, 2, (L, B))
    qmask = np.eye(2, dtype=np.float32)[spk]
    lens = np.full(B, L)
    if ragged and B > 1:
        lens = rs.randint(max(1, L // 2), L + 1, B)
        lens[0] = L                                       # the reference needs max(len) == L (:373-375)
    umask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    tm = umask.T[:, :, None]
    x = x * tm
    qmask = qmask * tm
    label = (rs.randint(0, n_classes, (B, L)) * umask).astype(np.int64)
    return (torch.tensor(x, dtype=dtype), torch.tensor(qmask, dtype=dtype),
            torch.tensor(umask, dtype=dtype), torch.tensor(label))
