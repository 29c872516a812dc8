"""MI355X-native mirror of the reference's ``models.lsthm_nsps`` (reference file model/lsthm_nsps.py).

Same class names, constructor / forward signatures, parameter names, shapes, initialisers and registration order as the reference
(``state_dict`` files interchange; ``torch.manual_seed(s); MARN1_nsps(6, "IEMOCAP")`` draws the same initial weights).  Against
``MARN1_onlysp``: the speaker GRU is fed the PRE-encoder features ``cat[linear_in(text) | audio]`` and blends the listener's state
into both parties (:177-191; a ``gru_l`` is constructed and never used), the cell returns ``(h, h_l, h_a, h_sp, h_li)``,
``CrossAttention2`` honours its (dh, dk, dv) arguments and ends in residual + LayerNorm (:75-108), and the head fuses
``softmax(p)``-weighted concatenations with an ``fc`` residual (:347-355).  The arithmetic is ``mser.nsps_fn`` (HIP only).
"""
import torch
import torch.nn as nn

from mser import functional as F_
from mser import ops
from mser.autograd import ModuleFn, require_gpu
from mser.flat import FlatStore
from mser.functional import Layout
from mser.gru_cell_fn import GRU_CELL_LIVE, gru_cell_backward, gru_cell_forward
from mser.model_fn import DropCfg, ModelDims
from mser.nsps_fn import LN_EPS, nsps_backward, nsps_forward
from models.encoder import EncoderLayer, MultiHeadAttention, _Grads  # noqa: F401  (the reference file imports both, :7)
from models.lsthm_sps import LSTHM1, CrossAttention, CrossAttention3  # noqa: F401  (same classes in the reference file)

# parameters that never receive a gradient in the reference: the unused attention modules of the cell and the listener GRU
_CELL_DEAD = ["crossatt_l2a.Wv", "crossatt_a2l.Wq", "crossatt_a2l.Wk", "crossatt_a2l.Wv"] + \
             [f"gru_l.{n}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


class CrossAttention2(nn.Module):
    """Reference model/lsthm_nsps.py:75-108: single-head attention across the utterance axis, ``output += residual`` (x_1),
    LayerNorm(dh, eps=1e-6).  x_1 [L1,B,dh], x_2 [L2,B,dh] -> [L1,B,dv] (dv must equal dh for the residual, as in the reference)."""

    def __init__(self, dh, dk, dv, attn_dropout=0.2):
        super(CrossAttention2, self).__init__()
        self.dh = dh
        self.dk = dk
        self.dv = dv
        self.Wq = nn.Parameter(torch.ones(self.dh, self.dk))
        self.Wk = nn.Parameter(torch.ones(self.dh, self.dk))
        self.Wv = nn.Parameter(torch.ones(self.dh, self.dv))
        self.dropout = nn.Dropout(attn_dropout)
        self.layer_norm = nn.LayerNorm(self.dh, eps=1e-6)

    def forward(self, x_1, x_2):
        require_gpu(x_1, x_2)
        if self.dv != self.dh:
            raise RuntimeError("CrossAttention2: dv must equal dh (output += residual, model/lsthm_nsps.py:105)")
        drop = self._last_drop = F_.module_site(self.dropout, x_1.device, 4)
        eps = self.layer_norm.eps

        class Impl:
            @staticmethod
            def fwd(x1, x2, Wq, Wk, Wv, gamma, beta):
                L1, B, D1 = x1.shape
                L2 = x2.shape[0]
                a = x1.contiguous().view(L1 * B, D1)
                b = x2.contiguous().view(L2 * B, -1)
                att = torch.empty(L1 * B, Wv.shape[1], device=x1.device)
                c = F_.xattn_fwd(a, None, b, None, Wq, Wk, Wv, Layout.time_major(L1, B), Layout.time_major(L2, B), att, 1, drop=drop)
                y, ssum, st = torch.empty_like(att), torch.empty_like(att), torch.empty(2, L1 * B, device=x1.device)
                ops.add_layernorm_fwd(att, a, gamma, beta, y, ssum, st[0], st[1], eps)
                return y.view(L1, B, -1), (c, Wq, Wk, Wv, gamma, ssum, st)

            @staticmethod
            def bwd(saved, tensors, dout):
                c, Wq, Wk, Wv, gamma, ssum, st = saved
                gq, gk, gv = torch.zeros_like(Wq), torch.zeros_like(Wk), torch.zeros_like(Wv)
                gg, gb = torch.zeros_like(gamma), torch.zeros_like(gamma)
                dsum = torch.empty_like(ssum)
                ops.layernorm_bwd(dout.contiguous().view(ssum.shape), ssum, st[0], st[1], gamma, dsum, gg, gb)
                dx1, dx2 = dsum.clone(), torch.zeros_like(c.x2)
                F_.xattn_bwd(c, dsum, Wq, Wk, Wv, gq, gk, gv, dx1, dx2, None, None)
                return (dx1.view(tensors[0].shape), dx2.view(tensors[1].shape), gq, gk, gv, gg, gb)

        return ModuleFn.apply(Impl, x_1, x_2, self.Wq, self.Wk, self.Wv, self.layer_norm.weight, self.layer_norm.bias)


class MARN_cell(nn.Module):
    """Reference model/lsthm_nsps.py:140-240.  forward(x [T,N,d_l+d_a], x_l, x_a [T,N,100], qmask [T,N,2]) ->
    (h [T,N,3*128] = cat(h_l, h_a, z_l), h_l, h_a, h_sp, h_li [T,N,128]) (:158-216)."""

    def __init__(self, dh_l, dh_a, d_l, d_a, dropout=0.5) -> None:
        super(MARN_cell, self).__init__()
        self.crossatt_l2a = CrossAttention()
        self.crossatt_a2l = CrossAttention()
        self.dh_l, self.dh_a = dh_l, dh_a
        self.dh_q = dh_l
        self.d_l, self.d_a = d_l, d_a
        self.speaker_size = 4 * self.dh_l
        self.dh_s = 128
        self.lsthm_l = LSTHM1(self.dh_l, self.d_l, self.dh_l, self.dh_s)
        self.lsthm_a = LSTHM1(self.dh_a, self.d_a, self.dh_l, self.dh_s)
        self.gru_s = nn.GRUCell(self.d_l + self.d_a, self.dh_s)
        self.gru_l = nn.GRUCell(self.d_l + self.d_a, self.dh_s)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, x_l, x_a, qmask):
        require_gpu(x, x_l, x_a, qmask)
        if not (self.dh_l == self.dh_a == self.dh_s == 128) or self.d_l != self.d_a:
            raise RuntimeError("MARN_cell: the reference only runs with dh_l == dh_a == dh_s == 128 and d_l == d_a")
        H, D = self.dh_l, self.d_l
        params = dict(self.named_parameters())
        names = GRU_CELL_LIVE
        ds, da = F_.module_site(self.dropout, x_l.device, 8), F_.module_site(self.crossatt_l2a.dropout, x_l.device, 10)
        if ds is not None and da is not None:
            da = F_.DropSite(ds.rng, ds.site + 2, da.p)
        self._last_drops = (ds, da)

        class Impl:
            @staticmethod
            def fwd(x, x_l, x_a, qmask, *pv):
                P = dict(zip(names, pv)).get
                T, N, _ = x_l.shape
                x2 = x.contiguous().view(T * N, 2 * D)
                xl2, xa2 = x_l.contiguous().view(T * N, D), x_a.contiguous().view(T * N, D)
                out, hli, ctx = gru_cell_forward(P, x2[:, :D], x2[:, D:], xl2, xa2, qmask.contiguous().float(), T, N, D, H, True, ds, da)
                o = out.view(T, N, 4 * H)
                # (h, h_l, h_a, h_sp, h_li): views / copies of the cell's output rows (:202-216)
                return (o[:, :, :3 * H].contiguous(), o[:, :, :H].contiguous(), o[:, :, H:2 * H].contiguous(),
                        o[:, :, 3 * H:].contiguous(), hli.view(T, N, H)), (ctx, T, N)

            @staticmethod
            def bwd(saved, tensors, dh, dhl, dha, dhsp, dhli):
                ctx, T, N = saved
                dev = tensors[1].device
                dout = torch.zeros(T * N, 4 * H, device=dev)
                if dh is not None:
                    ops.add_rows(dout[:, :3 * H], dout[:, :3 * H], dh.contiguous().view(T * N, 3 * H))
                for g_, sl in ((dhl, slice(0, H)), (dha, slice(H, 2 * H)), (dhsp, slice(3 * H, 4 * H))):
                    if g_ is not None:
                        ops.add_rows(dout[:, sl], dout[:, sl], g_.contiguous().view(T * N, H))
                P = dict(zip(names, [params[n].detach() for n in names])).get
                G = _Grads({n: params[n] for n in names})
                dli = dhli.contiguous().view(T * N, H) if dhli is not None else None
                dx_l, dx_a, du_l, du_a = gru_cell_backward(ctx, P, G.g.get, dout, dli, True)
                dx = torch.empty(T * N, 2 * D, device=dev)
                ops.add_rows(dx[:, :D], du_l)
                ops.add_rows(dx[:, D:], du_a)
                return (dx.view(tensors[0].shape), dx_l.view(tensors[1].shape), dx_a.view(tensors[2].shape), None,
                        *[G(n) for n in names])

        return ModuleFn.apply(Impl, x, x_l, x_a, qmask, *[params[n] for n in names])


class _NspsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, hook, x, qmask, umask):
        store = model._store
        lp, x_l, x_a, c = nsps_forward(store.p, x, qmask, umask, model.dims, no_en=model._no_en, drop=model._drop_cfg(x.device))
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
        nsps_backward(c, store.p, store.g, dlp, dxl, dxa)
        store.publish_grads()
        ctx.c = None
        return None, None, None, None, None


class MARN1_nsps(nn.Module):
    """Reference model/lsthm_nsps.py:242-360.  forward(x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L]) ->
    (log_probs [B*L, n_classes], x_l [L,B,100], x_a [L,B,100]).  ``dataset`` is accepted and unused, as in the reference (:243)."""

    _no_en = False

    def __init__(self, n_classes, dataset, *, d_r=1024):
        super(MARN1_nsps, self).__init__()
        self.d_l, self.d_a, self.d_r = 100, 100, d_r
        self.dh_l, self.dh_a = 128, 128
        self.dh_sp, self.dh_li = 128, 128
        self.total_h_dim = self.dh_l + self.dh_a

        self.linear_in = nn.Linear(self.d_r, self.d_l)
        self.marn_cell_f = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a)
        self.marn_cell_b = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a)

        output_dim = n_classes
        final_out = 2 * (self.total_h_dim + self.d_l)
        h_out = 32
        out_dropout = 0.5
        self.fc = nn.Sequential(nn.Linear(self.d_l, final_out), nn.ReLU(), nn.Dropout(out_dropout))
        self.fc2 = nn.Sequential(nn.Linear(self.d_a, final_out), nn.ReLU(), nn.Dropout(out_dropout))
        self.nn_out = nn.Sequential(nn.Linear(final_out, h_out), nn.ReLU(), nn.Dropout(out_dropout), nn.Linear(h_out, output_dim))
        self.dropout_rec = nn.Dropout(0.5)

        d_inner, n_head, d_k, d_v = 40, 8, 40, 40
        self.encoder_l = EncoderLayer(self.d_l, d_inner, n_head, d_k, d_v)
        self.encoder_a = EncoderLayer(self.d_a, d_inner, n_head, d_k, d_v)
        self.crossatt_l2a = CrossAttention2(self.d_l, self.d_l, self.d_l)
        self.crossatt_a2l = CrossAttention2(self.d_a, self.d_a, self.d_a)

        self.p = nn.Parameter(torch.ones(2))

        self.dims = ModelDims(d_r=d_r, d_a=self.d_a, D=self.d_l, H=self.dh_l, n_head=n_head, d_k=d_k, d_v=d_v, n_classes=n_classes)
        self.dropout_seed = 0x5EED
        self.dropout_enabled = True
        self._rng = None
        dead = [c + n for c in ("marn_cell_f.", "marn_cell_b.") for n in _CELL_DEAD]
        dead += [e + n for e in ("encoder_l.", "encoder_a.") for n in ("pos_ffn.fc.weight", "pos_ffn.fc.bias")]
        dead += ["fc2.0.weight", "fc2.0.bias"]                     # resid_a = fc2(x_a) is computed and never used (:351)
        if self._no_en:                                            # lsthm_no_en.py:306,:309: the text encoder never runs
            dead += [n for n, _ in self.named_parameters() if n.startswith("encoder_l.") and n not in dead]
        self._store = FlatStore(self, dead=dead)
        self._hook = None

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
        """Train mode: the p of every nn.Dropout of the module tree and this step's generator words; eval mode: None."""
        if not (self.training and self.dropout_enabled):
            return None
        el, ea = self.encoder_l, self.encoder_a
        cfg = DropCfg(
            p_enc_l=(0.0, 0.0, 0.0) if self._no_en else (el.slf_attn.attention.dropout.p, el.slf_attn.dropout.p, el.pos_ffn.dropout.p),
            p_enc_a=(ea.slf_attn.attention.dropout.p, ea.slf_attn.dropout.p, ea.pos_ffn.dropout.p),
            p_xattn=(self.crossatt_l2a.dropout.p, self.crossatt_a2l.dropout.p, 0.0, 0.0),
            p_fc=self.fc[2].p, p_out=self.nn_out[2].p, p_rec=self.dropout_rec.p,
            p_cell=(self.marn_cell_f.dropout.p, self.marn_cell_b.dropout.p),
            p_cell_attn=(self.marn_cell_f.crossatt_l2a.dropout.p, self.marn_cell_b.crossatt_l2a.dropout.p))
        if not cfg.any():
            return None
        if self._rng is None or self._rng.device != device:
            seed = self.dropout_seed
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed += 0x9E3779B1 * torch.distributed.get_rank()
            self._rng = torch.tensor([seed & 0x7FFFFFFF, 0], dtype=torch.int32, device=device)
        ops.rng_advance_(self._rng)
        cfg.rng = self._rng.clone()
        return cfg

    def forward(self, x, qmask, umask):
        require_gpu(x, qmask, umask)
        self._ensure_attached(x.device)
        hook = self._hook if torch.is_grad_enabled() else self._hook.detach()
        return _NspsFn.apply(self, hook, x, qmask, umask)

    def _reverse_seq(self, X, mask):
        """Reference :362-374 -- flip the first len_b steps of every dialogue, zero-pad (HIP gather kernel)."""
        require_gpu(X, mask)
        Ln, B = X.shape[0], X.shape[1]
        X2 = X.contiguous().float().view(Ln * B, -1)
        lens = torch.empty(B, device=X.device, dtype=torch.int32)
        rev = torch.empty(Ln, B, device=X.device, dtype=torch.int32)
        ops.build_reverse_index(mask.contiguous().float(), lens, rev)
        out = torch.empty_like(X2)
        ops.reverse_by_length(X2, rev, out, Ln, B)
        return out.view(X.shape)
