"""MI355X-native mirror of the reference's ``models.DialogueRNN`` on the path the trainers use (reference file
model/DialogueRNN.py; SURVEY.md 8(f) row f2, BASELINE configs[3]): ``MatchingAttention`` (:24-77, the 'general' and 'general2'
types), ``DialogueRNNCell`` (:80-166), ``DialogueRNN`` (:169-198) and ``BiModel`` (:201-277) with the reference's constructor
signatures, parameter names, shapes, registration order and initialisers (``state_dict`` files interchange).  ``BiModel.forward``
is ONE autograd node whose forward / backward are the kernel sequences of ``mser.bimodel_fn`` (HIP only).  The cell and the
single-direction ``DialogueRNN`` hold the parameters in the reference's layout; their arithmetic runs inside ``BiModel`` (both
directions share every launch).  Of the reference's other variants only what BiModel constructs is mirrored: 'simple' / 'dot' /
'concat' attention, listener_state=False and the E2E / AVEC / DailyDialogue models are outside the hot path.
"""
import torch
import torch.nn as nn

from mser import ops
from mser.autograd import require_gpu
from mser.bimodel_fn import BiDims, BiDrop, bimodel_alpha_dir, bimodel_backward, bimodel_forward
from mser.flat import FlatStore


class MatchingAttention(nn.Module):
    """Reference :24-77.  Parameter holder for the types BiModel uses ('general': transform without bias; 'general2': with bias)."""

    def __init__(self, mem_dim, cand_dim, alpha_dim=None, att_type='general'):
        super(MatchingAttention, self).__init__()
        assert att_type != 'concat' or alpha_dim is not None
        assert att_type != 'dot' or mem_dim == cand_dim
        if att_type not in ('general', 'general2'):
            raise NotImplementedError(f"MatchingAttention att_type '{att_type}' is outside the accelerated path (BiModel uses 'general' "
                                      "inside the cell and 'general2' over the emotion states)")
        self.mem_dim = mem_dim
        self.cand_dim = cand_dim
        self.att_type = att_type
        self.transform = nn.Linear(cand_dim, mem_dim, bias=(att_type == 'general2'))

    def forward(self, M, x, mask=None):
        raise NotImplementedError("models.DialogueRNN.MatchingAttention runs inside BiModel (mser.bimodel_fn)")


class DialogueRNNCell(nn.Module):
    """Reference :80-166 (parameters; the step itself is csrc/dialogue.hip)."""

    def __init__(self, D_m, D_g, D_p, D_e, listener_state=False, context_attention='simple', D_a=100, dropout=0.5):
        super(DialogueRNNCell, self).__init__()
        if not listener_state or context_attention != 'general':
            raise NotImplementedError("DialogueRNNCell: the accelerated path is the trainers' configuration (listener_state=True, "
                                      "context_attention='general', model_trainer.py:42-47)")
        self.D_m = D_m
        self.D_g = D_g
        self.D_p = D_p
        self.D_e = D_e
        self.listener_state = listener_state
        self.g_cell = nn.GRUCell(D_m + D_p, D_g)
        self.p_cell = nn.GRUCell(D_m + D_g, D_p)
        self.e_cell = nn.GRUCell(D_p, D_e)
        if listener_state:
            self.l_cell = nn.GRUCell(D_m + D_p, D_p)
        self.dropout = nn.Dropout(dropout)
        self.attention = MatchingAttention(D_g, D_m, D_a, context_attention)

    def forward(self, U, qmask, g_hist, q0, e0):
        raise NotImplementedError("models.DialogueRNN.DialogueRNNCell runs inside BiModel (both directions share the launches)")


class DialogueRNN(nn.Module):
    """Reference :169-198."""

    def __init__(self, D_m, D_g, D_p, D_e, listener_state=False, context_attention='simple', D_a=100, dropout=0.5):
        super(DialogueRNN, self).__init__()
        self.D_m = D_m
        self.D_g = D_g
        self.D_p = D_p
        self.D_e = D_e
        self.dropout = nn.Dropout(dropout)
        self.dialogue_cell = DialogueRNNCell(D_m, D_g, D_p, D_e, listener_state, context_attention, D_a, dropout)

    def forward(self, U, qmask):
        raise NotImplementedError("models.DialogueRNN.DialogueRNN runs inside BiModel (both directions share the launches)")


class _BiModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, hook, U, qmask, umask):
        store = model._store
        lp, alpha, c = bimodel_forward(store.p, U, qmask, umask, model.dims, drop=model._drop_cfg(U.device))
        ctx.model, ctx.c = model, c
        model._last_ctx = c
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(alpha)
        return lp, alpha

    @staticmethod
    def backward(ctx, dlp, _dalpha):
        model, c = ctx.model, ctx.c
        if c is None:           # the saved activations are released by the first backward (as autograd does without retain_graph)
            raise RuntimeError("backward through this forward a second time: its saved activations have been freed; run the forward again")
        store = model._store
        if store.grads_were_reset():
            store.zero_grad()
        if dlp is None:
            dlp = torch.zeros_like(c.lp).view(c.L, c.B, -1)
        bimodel_backward(c, store.p, store.g, dlp)
        store.publish_grads()
        ctx.c = None
        return None, None, None, None, None


class BiModel(nn.Module):
    """Reference :201-277.  forward(U [L,B,D_m], qmask [L,B,2], umask [B,L], att2=True) ->
    (log_prob [L,B,n_classes], alpha: list of L tensors [B,L], alpha_f, alpha_b: lists of L-1 tensors [B,t], t = 1..L-1)."""

    def __init__(self, D_m, D_g, D_p, D_e, D_h, n_classes=7, listener_state=False, context_attention='simple', D_a=100,
                 dropout_rec=0.5, dropout=0.5):
        super(BiModel, self).__init__()
        self.D_m = D_m
        self.D_g = D_g
        self.D_p = D_p
        self.D_e = D_e
        self.D_h = D_h
        self.n_classes = n_classes
        self.dropout = nn.Dropout(dropout)
        self.dropout_rec = nn.Dropout(dropout + 0.15)
        self.dialog_rnn_f = DialogueRNN(D_m, D_g, D_p, D_e, listener_state, context_attention, D_a, dropout_rec)
        self.dialog_rnn_r = DialogueRNN(D_m, D_g, D_p, D_e, listener_state, context_attention, D_a, dropout_rec)
        self.linear = nn.Linear(2 * D_e, 2 * D_h)
        self.smax_fc = nn.Linear(2 * D_h, n_classes)
        self.matchatt = MatchingAttention(2 * D_e, 2 * D_e, att_type='general2')

        self.dims = BiDims(D_m=D_m, D_g=D_g, D_p=D_p, D_e=D_e, D_h=D_h, n_classes=n_classes)
        self.dropout_seed = 0x5EED
        self.dropout_enabled = True
        self._rng = None
        self._last_ctx = None
        self._store = FlatStore(self, dead=[])
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
        if not (self.training and self.dropout_enabled):
            return None
        cfg = BiDrop(p_cell=self.dialog_rnn_f.dialogue_cell.dropout.p, p_rec=self.dropout_rec.p, p_hid=self.dropout.p)
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

    def forward(self, U, qmask, umask, att2=True):
        require_gpu(U, qmask, umask)
        if not att2:
            raise NotImplementedError("BiModel.forward(att2=False): both reference trainers call att2=True (model_trainer_d.py:63)")
        self._ensure_attached(U.device)
        hook = self._hook if torch.is_grad_enabled() else self._hook.detach()
        lp, alpha = _BiModelFn.apply(self, hook, U, qmask, umask)
        Ln = U.shape[0]
        c = self._last_ctx
        alphas = [alpha[:, t, :] for t in range(Ln)]                                 # :260 alpha_[:,0,:] per position
        af, ab = bimodel_alpha_dir(c, 0), bimodel_alpha_dir(c, 1)
        alpha_f = [af[t, :, :t] for t in range(1, Ln)]                               # :196: steps 1 .. L-1 (step 0 has no history)
        alpha_b = [ab[t, :, :t] for t in range(1, Ln)]
        return lp, alphas, alpha_f, alpha_b

    def _reverse_seq(self, X, mask):
        """Reference :224-239."""
        require_gpu(X, mask)
        Ln, B = X.shape[0], X.shape[1]
        X2 = X.contiguous().float().view(Ln * B, -1)
        lens = torch.empty(B, device=X.device, dtype=torch.int32)
        rev = torch.empty(Ln, B, device=X.device, dtype=torch.int32)
        ops.build_reverse_index(mask.contiguous().float(), lens, rev)
        out = torch.empty_like(X2)
        ops.reverse_by_length(X2, rev, out, Ln, B)
        return out.view(X.shape)
