"""MI355X-native mirror of the reference's ``models.lsthm_sps`` (reference file model/lsthm_sps.py).

Class names, constructor / forward signatures, parameter names, shapes and initialisers follow the reference, so
``state_dict`` files interchange and ``torch.manual_seed(s); MARN1_sps(6)`` draws the same initial weights.  All the
arithmetic runs in libmser (HIP, gfx950): ``MARN1_sps.forward`` is ONE autograd node whose forward and backward are the
explicit kernel sequences of ``mser.model_fn``; parameters live in one flat buffer (``mser.flat``).

Differences that are deliberate and documented: in train mode the 13 Dropout sites draw their masks from libmser's
counter-based generator (the reference's CPU generator streams cannot be reproduced on a GPU; parity is defined at p = 0 / eval
and, in train mode, mask for mask against the oracle);
extra keyword-only constructor arguments (``d_r``, ``xattn_heads``, ``hidden``) default to the reference's hard-coded values;
there is no CPU execution path.  ``hidden`` (128, 256; 512-2048 on per-step launches) sets every width the reference hard-codes as 128 (LSTHM cell, speaker
cell, rank-1 attention, sequence-level attention); at 256 parity is checked against the oracle only (the reference cannot be
constructed at that width).
"""
import torch
import torch.nn as nn

from mser import functional as F_
from mser import ops
from mser.autograd import ModuleFn, require_gpu
from mser.flat import FlatStore
from mser.functional import Layout
from mser.model_fn import DropCfg, ModelDims, marn1_backward, marn1_forward
from models.encoder import EncoderLayer, _Grads


class LSTHM1(nn.Module):
    """Reference model/lsthm_sps.py:11-44.  forward(x, ctm, htm, ztm, speaker_affine) -> (c_t, h_t); gate order f,i,o,c~."""

    def __init__(self, cell_size, in_size, hybrid_in_size, speaker_dim):
        super(LSTHM1, self).__init__()
        self.cell_size = cell_size
        self.in_size = in_size
        self.W = nn.Linear(in_size, 4 * self.cell_size)
        self.U = nn.Linear(cell_size, 4 * self.cell_size)
        self.V = nn.Linear(hybrid_in_size, 4 * self.cell_size)
        self.S = nn.Linear(speaker_dim, 4 * self.cell_size)

    def forward(self, x, ctm, htm, ztm, speaker_affine):
        require_gpu(x, ctm, htm, ztm, speaker_affine)
        names = ["W.weight", "W.bias", "U.weight", "U.bias", "V.weight", "V.bias", "S.weight", "S.bias"]
        params = dict(self.named_parameters())

        class Impl:
            @staticmethod
            def fwd(x, c, h, z, s, *pv):
                x, c, h, z, s = (t.contiguous() for t in (x, c, h, z, s))
                c2, h2 = torch.empty_like(c), torch.empty_like(c)
                gates = torch.empty(c.shape[0], 4 * c.shape[1], device=c.device)
                ops.lsthm_step_fwd(x, c, h, z, s, *pv, c2, h2, gates)
                return (c2, h2), (x, c, h, z, s, pv, c2, gates)

            @staticmethod
            def bwd(saved, tensors, dc2, dh2):
                # element-wise gate backward in one small launch; the four input products and the four weight gradients reuse
                # the MFMA GEMM.  (Training a whole model goes through the fused BPTT of MARN_cell / MARN1_sps instead.)
                x, c, h, z, s, pv, c2, gates = saved
                W, Wb, U, Ub, V, Vb, S, Sb = pv
                B, H = c.shape
                dgates = torch.empty(B, 4 * H, device=c.device)
                dc = torch.empty_like(c)
                ops.lsthm_step_bwd(gates, c, c2, dc2.contiguous() if dc2 is not None else None,
                                   dh2.contiguous() if dh2 is not None else None, dgates, dc)
                grads_in, grads_w = [], []
                for inp, Wm in ((x, W), (h, U), (z, V), (s, S)):
                    d_in = torch.empty_like(inp)
                    ops.matmul(dgates, Wm, d_in)                     # [B,4H] @ [4H,K]
                    gW = torch.zeros_like(Wm)
                    ops.grad_weight(dgates, inp, gW)                 # dW += dgates^T inp
                    grads_in.append(d_in)
                    grads_w.append(gW)
                gb = torch.zeros(4 * H, device=c.device)
                ops.colsum_acc(dgates, gb)                           # the four biases are summed into one pre-activation (:33)
                dx, dh, dz, ds = grads_in
                return (dx, dc, dh, dz, ds, grads_w[0], gb, grads_w[1], gb.clone(), grads_w[2], gb.clone(), grads_w[3], gb.clone())

        return ModuleFn.apply(Impl, x, ctm, htm, ztm, speaker_affine, *[params[n] for n in names])


class CrossAttention(nn.Module):
    """Reference model/lsthm_sps.py:47-72: per-step attention over the FEATURE axis; computed in its rank-1 form
    (logits[b,i,j] = x1[b,i] * <Wq,x2[b]>/sqrt(dh) * Wk[j]) without materialising [B,dh,dh].  ``Wv`` is unused there too."""

    def __init__(self, attn_dropout=0.2, *, dh=128):
        super(CrossAttention, self).__init__()
        self.dh = dh
        self.Wq = nn.Parameter(torch.ones(self.dh).unsqueeze(0))
        self.Wk = nn.Parameter(torch.ones(self.dh).unsqueeze(0))
        self.Wv = nn.Parameter(torch.ones(self.dh).unsqueeze(0))
        self.dropout = nn.Dropout(attn_dropout)

    def forward(self, x_1, x_2):
        require_gpu(x_1, x_2)
        drop = self._last_drop = F_.module_site(self.dropout, x_1.device, 3)

        class Impl:
            @staticmethod
            def fwd(x1, x2, Wq, Wk):
                x1, x2 = x1.contiguous(), x2.contiguous()
                out = torch.empty_like(x1)
                ops.rank1_attention_fwd(x1, x2, Wq, Wk, out, drop=drop)
                return out, (x1, x2, Wq, Wk)

            @staticmethod
            def bwd(saved, tensors, dout):
                x1, x2, Wq, Wk = saved
                dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
                gWq, gWk = torch.zeros_like(Wq), torch.zeros_like(Wk)
                ops.rank1_attention_bwd(x1, x2, Wq, Wk, dout.contiguous(), dx1, dx2, gWq, gWk, drop=drop)
                return (dx1, dx2, gWq, gWk)

        return ModuleFn.apply(Impl, x_1, x_2, self.Wq, self.Wk)


class _SeqCrossAttention(nn.Module):
    """Shared body of CrossAttention2 / CrossAttention3 (reference :75-101, :103-129): single-head (or, as an extension for the
    MFMA stress config, ``heads``-way) attention over the utterance axis between two time-major streams."""

    def _init(self, d1, d2, heads, hidden=128):
        self.dh, self.dk, self.dv = 100, hidden, hidden    # the reference ignores its ctor arguments and hard-codes 100, 128, 128
        self.heads = heads
        self.Wq = nn.Parameter(torch.ones(d1, self.dk))
        self.Wk = nn.Parameter(torch.ones(d2, self.dk))
        self.Wv = nn.Parameter(torch.ones(d2, self.dv))

    def forward(self, x_1, x_2):
        require_gpu(x_1, x_2)
        heads = self.heads
        drop = self._last_drop = F_.module_site(self.dropout, x_1.device, 4)

        class Impl:
            @staticmethod
            def fwd(x1, x2, Wq, Wk, Wv):
                L1, B, D1 = x1.shape
                L2 = x2.shape[0]
                a = x1.contiguous().view(L1 * B, D1)
                b = x2.contiguous().view(L2 * B, -1)
                out = torch.empty(L1 * B, Wv.shape[1], device=x1.device)
                c = F_.xattn_fwd(a, None, b, None, Wq, Wk, Wv, Layout.time_major(L1, B), Layout.time_major(L2, B), out, heads, drop=drop)
                return out.view(L1, B, -1), (c, Wq, Wk, Wv)

            @staticmethod
            def bwd(saved, tensors, dout):
                c, Wq, Wk, Wv = saved
                gq, gk, gv = torch.zeros_like(Wq), torch.zeros_like(Wk), torch.zeros_like(Wv)
                dx1, dx2 = torch.zeros_like(c.x1), torch.zeros_like(c.x2)
                F_.xattn_bwd(c, dout.contiguous().view(dx1.shape[0], -1), Wq, Wk, Wv, gq, gk, gv, dx1, dx2, None, None)
                return (dx1.view(tensors[0].shape), dx2.view(tensors[1].shape), gq, gk, gv)

        return ModuleFn.apply(Impl, x_1, x_2, self.Wq, self.Wk, self.Wv)


class CrossAttention2(_SeqCrossAttention):
    def __init__(self, dh, dk, dv, attn_dropout=0.2, *, heads=1, hidden=128):
        super(CrossAttention2, self).__init__()
        self._init(100, 100, heads, hidden)
        self.dropout = nn.Dropout(attn_dropout)


class CrossAttention3(_SeqCrossAttention):
    def __init__(self, dh, dk, dv, attn_dropout=0.2, *, heads=1, hidden=128):
        super(CrossAttention3, self).__init__()
        self._init(100, hidden, heads, hidden)
        self.dropout = nn.Dropout(attn_dropout)


_CELL_LIVE = ["lsthm_l.W.weight", "lsthm_l.W.bias", "lsthm_l.U.weight", "lsthm_l.U.bias", "lsthm_l.V.weight", "lsthm_l.V.bias",
              "lsthm_l.S.weight", "lsthm_l.S.bias", "lsthm_a.W.weight", "lsthm_a.W.bias", "lsthm_a.U.weight", "lsthm_a.U.bias",
              "lsthm_a.V.weight", "lsthm_a.V.bias", "lsthm_a.S.weight", "lsthm_a.S.bias",
              "lstm_q0.weight_ih", "lstm_q0.weight_hh", "lstm_q0.bias_ih", "lstm_q0.bias_hh",
              "lstm_q1.weight_ih", "lstm_q1.weight_hh", "lstm_q1.bias_ih", "lstm_q1.bias_hh",
              "crossatt_l2a.Wq", "crossatt_l2a.Wk"]
# parameters that never receive a gradient in the reference (SURVEY.md 7 "Dead parameters")
_CELL_DEAD = ["crossatt_l2a.Wv", "crossatt_a2l.Wq", "crossatt_a2l.Wk", "crossatt_a2l.Wv",
              "lstm_s.weight_ih", "lstm_s.weight_hh", "lstm_s.bias_ih", "lstm_s.bias_hh"]


class MARN_cell(nn.Module):
    """Reference model/lsthm_sps.py:132-221.  forward(x, x_l, x_a, qmask) -> h [T,N,3*dh+dh_s] = cat(h_l,h_a,z_l,h_q)."""

    def __init__(self, dh_l, dh_a, d_l, d_a, dropout=0.5, *, dh_s=128) -> None:
        super(MARN_cell, self).__init__()
        self.crossatt_l2a = CrossAttention(dh=dh_s)
        self.crossatt_a2l = CrossAttention(dh=dh_s)
        self.dh_l, self.dh_a = dh_l, dh_a
        self.dh_q = dh_l
        self.d_l, self.d_a = d_l, d_a
        self.speaker_size = 4 * self.dh_l
        self.dh_s = dh_s
        self.lsthm_l = LSTHM1(self.dh_l, self.d_l, self.dh_l, self.dh_s)
        self.lsthm_a = LSTHM1(self.dh_a, self.d_a, self.dh_l, self.dh_s)
        self.lstm_q0 = nn.LSTMCell(self.dh_s, self.dh_s)
        self.lstm_q1 = nn.LSTMCell(self.dh_s, self.dh_s)
        self.lstm_s = nn.LSTMCell(self.dh_s, self.dh_s)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, x_l, x_a, qmask):
        require_gpu(x_l, x_a, qmask)
        if not (self.dh_l == self.dh_a == self.dh_s):
            raise RuntimeError("MARN_cell: the reference only runs with dh_l == dh_a == dh_s (128 there)")
        H, D = self.dh_l, self.d_l
        params = dict(self.named_parameters())
        names = _CELL_LIVE
        # train mode on its own (outside MARN1_sps): sites +8 (h_q), +9 (h_l/h_a), +10 (rank-1 attention) of the module generator
        ds, da = F_.module_site(self.dropout, x_l.device, 8), F_.module_site(self.crossatt_l2a.dropout, x_l.device, 8)
        self._last_drops = (ds, da)
        cell_drop = None
        if ds is not None or da is not None:
            cell_drop = ((ds or da).rng, [F_.SITE_MODULE + 8], [ds.p if ds else 0.0], [da.p if da else 0.0])

        class Impl:
            @staticmethod
            def fwd(x_l, x_a, qmask, *pv):
                P = dict(zip(names, pv)).get
                T, N, _ = x_l.shape
                xl2 = x_l.contiguous().view(T * N, D)
                xa2 = x_a.contiguous().view(T * N, D)
                qm = qmask.contiguous().float()
                out = torch.empty(T * N, 4 * H, device=x_l.device)
                ws = torch.empty(ops.cell_workspace_bytes(T, N, D, H, 1), device=x_l.device, dtype=torch.uint8)
                dirs = [dict(p=ops.cell_param_struct(P), qmask=qm, rev=None, out=out)]
                ops.marn_cell_fwd(ops.make_cell_desc(T, N, D, H, xl2, xa2, dirs, 4 * H, ws, drop=cell_drop))
                return out.view(T, N, 4 * H), (xl2, xa2, dirs, ws, T, N)

            @staticmethod
            def bwd(saved, tensors, dout):
                xl2, xa2, dirs, ws, T, N = saved
                G = _Grads({n: params[n] for n in names})
                dirs[0]["g"] = ops.cell_param_struct(G.g.get)
                dirs[0]["dout"] = dout.contiguous().view(T * N, 4 * H)
                dx_l, dx_a = torch.zeros_like(xl2), torch.zeros_like(xa2)
                ops.marn_cell_bwd(ops.make_cell_desc(T, N, D, H, xl2, xa2, dirs, 4 * H, ws, dx_l=dx_l, dx_a=dx_a, drop=cell_drop))
                return (dx_l.view(tensors[0].shape), dx_a.view(tensors[1].shape), None, *[G(n) for n in names])

        return ModuleFn.apply(Impl, x_l, x_a, qmask, *[params[n] for n in names])


class _MARN1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, hook, x, qmask, umask):
        store = model._store
        lp, x_l, x_a, c = marn1_forward(store.p, x, qmask, umask, model.dims, use_streams=model.use_streams,
                                        drop=model._drop_cfg(x.device), prep_backward=bool(ctx.needs_input_grad[1]))
        ctx.model, ctx.c = model, c
        ctx.set_materialize_grads(False)
        return lp, x_l, x_a

    @staticmethod
    def backward(ctx, dlp, dxl, dxa):
        model, c = ctx.model, ctx.c
        if c is None:           # the saved activations are released by the first backward (as autograd does without retain_graph)
            raise RuntimeError("backward through this forward a second time: its saved activations have been freed; run the forward again")
        store = model._store
        if store.grads_were_reset():
            store.zero_grad()
        if dlp is None:
            dlp = torch.zeros_like(c.lp)
        marn1_backward(c, store.p, store.g, dlp, dxl, dxa, use_streams=model.use_streams)
        store.publish_grads()
        ctx.c = None
        return None, None, None, None, None


class MARN1_sps(nn.Module):
    """Reference model/lsthm_sps.py:298-409.  forward(x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L]) ->
    (log_probs [B*L, n_classes], x_l [L,B,100], x_a [L,B,100])."""

    def __init__(self, n_classes, *, d_r=1024, xattn_heads=1, hidden=128):
        super(MARN1_sps, self).__init__()
        if hidden not in (128, 256, 512, 1024, 2048):
            raise ValueError(f"hidden={hidden}: the recurrent cell is built for 128 (the reference's width) and 256 (persistent chains) "
                             f"and for 512, 1024, 2048 (one launch per phase and step)")
        self.d_l, self.d_a, self.d_r = 100, 100, d_r
        self.dh_l, self.dh_a = hidden, hidden
        self.dh_sp, self.dh_li = hidden, hidden
        self.total_h_dim = self.dh_l + self.dh_a

        self.linear_in = nn.Linear(self.d_r, self.d_l)
        self.marn_cell_f = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a, dh_s=hidden)
        self.marn_cell_b = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a, dh_s=hidden)

        output_dim = n_classes
        final_out = 2 * (self.total_h_dim + self.dh_l + self.dh_l) + self.dh_l + self.dh_a
        h_out = 32
        out_dropout = 0.5
        self.fc = nn.Sequential(nn.Linear(final_out, self.d_l), nn.ReLU(), nn.Dropout(out_dropout))
        self.nn_out = nn.Sequential(nn.Linear(self.d_l, h_out), nn.ReLU(), nn.Dropout(out_dropout), nn.Linear(h_out, output_dim))
        self.dropout_rec = nn.Dropout(0.5)

        d_inner, n_head, d_k, d_v = 40, 8, 40, 40
        self.encoder_l = EncoderLayer(100, d_inner, n_head, d_k, d_v)
        self.encoder_a = EncoderLayer(100, d_inner, n_head, d_k, d_v)
        self.crossatt_l2a = CrossAttention2(self.d_l, self.dh_l, self.dh_l, heads=xattn_heads, hidden=hidden)
        self.crossatt_a2l = CrossAttention2(self.d_a, self.dh_a, self.dh_a, heads=xattn_heads, hidden=hidden)
        self.crossatt_l2a_1 = CrossAttention3(self.dh_l, self.d_l, self.d_l, heads=xattn_heads, hidden=hidden)
        self.crossatt_a2l_1 = CrossAttention3(self.dh_a, self.d_a, self.d_a, heads=xattn_heads, hidden=hidden)

        self.w = nn.Parameter(torch.ones(1))
        self.v = nn.Parameter(torch.ones(1))
        self.v1 = nn.Parameter(torch.ones(1))
        self.v2 = nn.Parameter(torch.ones(1))

        self.dims = ModelDims(d_r=d_r, d_a=self.d_a, D=self.d_l, H=self.dh_l, n_head=n_head, d_k=d_k, d_v=d_v,
                              n_classes=n_classes, xattn_heads=xattn_heads)
        self.use_streams = True
        self.dropout_seed = 0x5EED
        self.dropout_enabled = True          # train mode draws the 13 dropout sites like the reference; False: identities
        self._rng = None
        dead = [c + n for c in ("marn_cell_f.", "marn_cell_b.") for n in _CELL_DEAD]
        dead += [e + n for e in ("encoder_l.", "encoder_a.") for n in ("pos_ffn.fc.weight", "pos_ffn.fc.bias")]
        self._store = FlatStore(self, dead=dead)
        self._hook = None

    # -- flat storage -------------------------------------------------------------------------------------------------
    @property
    def flat_store(self) -> FlatStore:
        return self._store

    def _ensure_attached(self, device):
        if not self._store.is_attached(device):
            for p in self.parameters():
                if p.device != device:
                    raise RuntimeError(f"model parameters are on {p.device} but the input is on {device}: call .to(device) first")
                break
            self._store.attach(device)
            self._hook = torch.zeros(1, device=device, requires_grad=True)

    def _drop_cfg(self, device):
        """Train mode: the p of every nn.Dropout of the module tree (so ``m.p = 0`` switches a site off, as in the reference) and
        this step's generator words; eval mode: None.  The step word advances on the device (capturable)."""
        if not (self.training and self.dropout_enabled):
            return None
        el, ea = self.encoder_l, self.encoder_a
        cfg = DropCfg(
            p_enc_l=(el.slf_attn.attention.dropout.p, el.slf_attn.dropout.p, el.pos_ffn.dropout.p),
            p_enc_a=(ea.slf_attn.attention.dropout.p, ea.slf_attn.dropout.p, ea.pos_ffn.dropout.p),
            p_xattn=(self.crossatt_l2a.dropout.p, self.crossatt_a2l.dropout.p, self.crossatt_l2a_1.dropout.p, self.crossatt_a2l_1.dropout.p),
            p_fc=self.fc[2].p, p_out=self.nn_out[2].p, p_rec=self.dropout_rec.p,
            p_cell=(self.marn_cell_f.dropout.p, self.marn_cell_b.dropout.p),
            p_cell_attn=(self.marn_cell_f.crossatt_l2a.dropout.p, self.marn_cell_b.crossatt_l2a.dropout.p))
        if not cfg.any():
            return None
        if self._rng is None or self._rng.device != device:
            seed = self.dropout_seed
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed += 0x9E3779B1 * torch.distributed.get_rank()        # data-parallel ranks draw independent masks
            self._rng = torch.tensor([seed & 0x7FFFFFFF, 0], dtype=torch.int32, device=device)
        ops.rng_advance_(self._rng)
        cfg.rng = self._rng.clone()          # this step's words: the backward reads them even if another forward has run since
        return cfg

    def forward(self, x, qmask, umask):
        require_gpu(x, qmask, umask)
        self._ensure_attached(x.device)
        hook = self._hook if torch.is_grad_enabled() else self._hook.detach()
        return _MARN1Fn.apply(self, hook, x, qmask, umask)

    def _reverse_seq(self, X, mask):
        """Reference :396-409 -- flip the first len_b steps of every dialogue, zero-pad (HIP gather kernel)."""
        require_gpu(X, mask)
        Ln, B = X.shape[0], X.shape[1]
        X2 = X.contiguous().float().view(Ln * B, -1)
        lens = torch.empty(B, device=X.device, dtype=torch.int32)
        rev = torch.empty(Ln, B, device=X.device, dtype=torch.int32)
        ops.build_reverse_index(mask.contiguous().float(), lens, rev)
        out = torch.empty_like(X2)
        ops.reverse_by_length(X2, rev, out, Ln, B)
        return out.view(X.shape)
