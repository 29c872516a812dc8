"""MI355X-native mirror of the reference's ``models.lsthm_onlysp`` (reference file model/lsthm_onlysp.py; ``MARN1_onlysp`` is the
reference CLI's default model, train.py:126).

Same class names, constructor / forward signatures, parameter names, shapes, initialisers and registration order (``state_dict``
files interchange; ``torch.manual_seed(s); MARN1_onlysp(6)`` draws the same initial weights).  Against ``MARN1_sps``: every cell
owns ``gru_s`` = GRUCell(d_l + d_a, 128), the speaker state of a dialogue (no slot compaction -- batch-independent); the second
encoder pass takes the first pass's output; the head is ``nn_out`` on the concatenation.  The arithmetic is ``mser.onlysp_fn``
(HIP only, no CPU path).  ``LSTHM1``, ``CrossAttention`` and ``CrossAttention2/3`` are the same modules as in ``models.lsthm_sps``.
"""
import torch
import torch.nn as nn

from mser import fault, ops
from mser import functional as F_
from mser.autograd import ModuleFn, require_gpu
from mser.gru_cell_fn import GRU_CELL_LIVE, gru_cell_backward, gru_cell_forward
from mser.flat import FlatStore
from mser.model_fn import DropCfg, ModelDims
from mser.onlysp_fn import onlysp_backward, onlysp_forward
from models.encoder import EncoderLayer, _Grads
from models.lsthm_sps import LSTHM1, CrossAttention, CrossAttention2, CrossAttention3  # noqa: F401  (same classes in the reference file)

# parameters that never receive a gradient in the reference: the LSTM cells kept from the sps variant, the unused attention modules
_CELL_DEAD = ["crossatt_l2a.Wv", "crossatt_a2l.Wq", "crossatt_a2l.Wk", "crossatt_a2l.Wv"] + \
             [f"{c}.{n}" for c in ("lstm_q0", "lstm_q1", "lstm_s") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


class MARN_cell(nn.Module):
    """Reference model/lsthm_onlysp.py:131-206.  forward(x, x_l, x_a, qmask) -> h [T,N,4*128] = cat(h_l, h_a, z_l, h_s) (:158-197;
    ``x`` only supplies the shape, as in the reference).  Inside ``MARN1_onlysp`` both directions share the launches
    (mser.onlysp_fn); this forward is the same kernels for one direction (mser.gru_cell_fn)."""

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
        self.lstm_q0 = nn.LSTMCell(self.dh_s, self.dh_s)
        self.lstm_q1 = nn.LSTMCell(self.dh_s, self.dh_s)
        self.gru_s = nn.GRUCell(self.d_l + self.d_a, self.dh_s)
        self.lstm_s = nn.LSTMCell(self.dh_s, self.dh_s)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, x_l, x_a, qmask):
        require_gpu(x_l, x_a, qmask)
        if not (self.dh_l == self.dh_a == self.dh_s == 128) or self.d_l != self.d_a:
            raise RuntimeError("MARN_cell: the reference only runs with dh_l == dh_a == dh_s == 128 and d_l == d_a")
        H, D = self.dh_l, self.d_l
        params = dict(self.named_parameters())
        names = GRU_CELL_LIVE
        # train mode on its own: sites +8 (h_s), +9 (h_l / h_a), +10 (rank-1 attention) of the module generator
        ds, da = F_.module_site(self.dropout, x_l.device, 8), F_.module_site(self.crossatt_l2a.dropout, x_l.device, 10)
        if ds is not None and da is not None:
            da = F_.DropSite(ds.rng, ds.site + 2, da.p)        # one generator word pair for the whole call
        self._last_drops = (ds, da)

        class Impl:
            @staticmethod
            def fwd(x_l, x_a, qmask, *pv):
                P = dict(zip(names, pv)).get
                T, N, _ = x_l.shape
                xl2, xa2 = x_l.contiguous().view(T * N, D), x_a.contiguous().view(T * N, D)
                out, _, ctx = gru_cell_forward(P, xl2, xa2, xl2, xa2, qmask.contiguous().float(), T, N, D, H, False, ds, da)
                return out.view(T, N, 4 * H), (ctx, T, N)

            @staticmethod
            def bwd(saved, tensors, dout):
                ctx, T, N = saved
                P = dict(zip(names, [params[n].detach() for n in names])).get
                G = _Grads({n: params[n] for n in names})
                dx_l, dx_a, du_l, du_a = gru_cell_backward(ctx, P, G.g.get, dout.contiguous().view(T * N, 4 * H), None, True)
                ops.add_rows(dx_l, dx_l, du_l)                 # U = cat(x_l[t], x_a[t]) (:172): the GRU reads the same rows
                ops.add_rows(dx_a, dx_a, du_a)
                return (dx_l.view(tensors[0].shape), dx_a.view(tensors[1].shape), None, *[G(n) for n in names])

        return ModuleFn.apply(Impl, x_l, x_a, qmask, *[params[n] for n in names])


class _OnlyspFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, hook, x, qmask, umask):
        store = model._store
        lp, x_l, x_a, c = onlysp_forward(store.p, x, qmask, umask, model.dims, drop=model._drop_cfg(x.device))
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
        onlysp_backward(c, store.p, store.g, dlp, dxl, dxa)
        store.publish_grads()
        ctx.c = None
        return None, None, None, None, None


class MARN1_onlysp(nn.Module):
    """Reference model/lsthm_onlysp.py:209-300.  forward(x [L,B,d_r+d_a], qmask [L,B,2], umask [B,L]) ->
    (log_probs [B*L, n_classes], x_l [L,B,100], x_a [L,B,100])."""

    def __init__(self, n_classes, *, d_r=1024):
        super(MARN1_onlysp, self).__init__()
        self.d_l, self.d_a, self.d_r = 100, 100, d_r
        self.dh_l, self.dh_a = 128, 128
        self.dh_sp, self.dh_li = 128, 128
        self.total_h_dim = self.dh_l + self.dh_a

        self.linear_in = nn.Linear(self.d_r, self.d_l)
        self.marn_cell_f = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a)
        self.marn_cell_b = MARN_cell(self.dh_l, self.dh_a, self.d_l, self.d_a)

        self.num_atts = 4
        output_dim = n_classes
        final_out = 2 * (self.total_h_dim + self.dh_l + self.dh_l) + self.dh_l + self.dh_a
        h_out = 32
        out_dropout = 0.5
        self.linear = nn.Linear(final_out, h_out)          # constructed and never used by the reference (:229)
        self.nn_out = nn.Sequential(nn.Linear(final_out, h_out), nn.ReLU(), nn.Dropout(out_dropout), nn.Linear(h_out, output_dim))
        self.dropout_rec = nn.Dropout(0.5)

        d_inner, n_head, d_k, d_v = 40, 8, 40, 40
        self.encoder_l = EncoderLayer(100, d_inner, n_head, d_k, d_v)
        self.encoder_a = EncoderLayer(100, d_inner, n_head, d_k, d_v)
        self.crossatt_l2a = CrossAttention2(self.d_l, self.dh_l, self.dh_l)
        self.crossatt_a2l = CrossAttention2(self.d_a, self.dh_a, self.dh_a)
        self.crossatt_l2a_1 = CrossAttention3(self.dh_l, self.d_l, self.d_l)
        self.crossatt_a2l_1 = CrossAttention3(self.dh_a, self.d_a, self.d_a)

        self.w = nn.Parameter(torch.ones(1))
        self.v = nn.Parameter(torch.ones(1))
        self.v1 = nn.Parameter(torch.ones(1))
        self.v2 = nn.Parameter(torch.ones(1))

        self.dims = ModelDims(d_r=d_r, d_a=self.d_a, D=self.d_l, H=self.dh_l, n_head=n_head, d_k=d_k, d_v=d_v, n_classes=n_classes)
        self.dropout_seed = 0x5EED
        self.dropout_enabled = True
        self._rng = None
        dead = [c + n for c in ("marn_cell_f.", "marn_cell_b.") for n in _CELL_DEAD]
        dead += [e + n for e in ("encoder_l.", "encoder_a.") for n in ("pos_ffn.fc.weight", "pos_ffn.fc.bias")]
        dead += ["linear.weight", "linear.bias"]
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
        """Train mode: the p of every nn.Dropout of the module tree and this step's generator words; eval mode: None.  The cell's one
        nn.Dropout serves h_s, h_l and h_a (:177,:184,:186); there is no ``fc`` site."""
        if not (self.training and self.dropout_enabled):
            return None
        el, ea = self.encoder_l, self.encoder_a
        cfg = DropCfg(
            p_enc_l=(el.slf_attn.attention.dropout.p, el.slf_attn.dropout.p, el.pos_ffn.dropout.p),
            p_enc_a=(ea.slf_attn.attention.dropout.p, ea.slf_attn.dropout.p, ea.pos_ffn.dropout.p),
            p_xattn=(self.crossatt_l2a.dropout.p, self.crossatt_a2l.dropout.p, self.crossatt_l2a_1.dropout.p, self.crossatt_a2l_1.dropout.p),
            p_fc=0.0, p_out=self.nn_out[2].p, p_rec=self.dropout_rec.p,
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

    def check_links(self) -> None:
        """Synchronising health check of every persistent / counter-linked launch issued on this device since the last check (the
        cell's own barriers, the linked GRU launches, label range): raises if the device's sticky fault word is set (mser.fault).
        ``ModelTrainer`` calls ``mser.fault.check`` itself once per epoch; kept as a method for callers of the bare model."""
        fault.check(self._store.data.device if self._store.data is not None else "cuda", "MARN1_onlysp")

    def forward(self, x, qmask, umask):
        require_gpu(x, qmask, umask)
        self._ensure_attached(x.device)
        hook = self._hook if torch.is_grad_enabled() else self._hook.detach()
        return _OnlyspFn.apply(self, hook, x, qmask, umask)
