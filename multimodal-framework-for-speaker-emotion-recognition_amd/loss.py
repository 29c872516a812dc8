"""MI355X-native mirror of the reference's ``loss.MaskedLoss`` (reference loss.py:6-25).

``MaskedLoss(losser, weight=None)``; ``forward(pred [B*L,C], target [B*L], mask [B,L])``.  ``pred`` holds log-probabilities
(the model ends in log_softmax), for which NLLLoss and CrossEntropyLoss coincide (log_softmax is idempotent), so both
``losser`` choices of the reference trainer (model_trainer.py:74-77) map onto one fused HIP kernel pair.
"""
import torch
import torch.nn as nn

from mser import ops
from mser.autograd import require_gpu


class _MaskedNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask):
        pred = pred.contiguous()
        target = target.contiguous()
        mask = mask.contiguous().float().view(-1)
        out = torch.empty(2, device=pred.device)
        ops.masked_nll_fwd(pred, target, mask, out)
        ctx.save_for_backward(target, mask, out)
        ctx.shape = pred.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        target, mask, out = ctx.saved_tensors
        dpred = torch.empty(ctx.shape, device=out.device)
        ops.masked_nll_bwd(target, mask, out, g.contiguous().view(1), dpred)
        return dpred, None, None


class MaskedLoss(nn.Module):

    def __init__(self, losser, weight=None):
        super(MaskedLoss, self).__init__()
        if weight is not None:
            raise NotImplementedError("class-weighted MaskedLoss is not on the accelerated path (the reference trainer passes weight=None)")
        if losser not in (nn.NLLLoss, nn.CrossEntropyLoss):
            raise ValueError("MaskedLoss: losser must be nn.NLLLoss or nn.CrossEntropyLoss (model_trainer.py:74-77)")
        self.weight = weight
        self.is_ce = losser is nn.CrossEntropyLoss

    def forward(self, pred, target, mask):
        """pred -> batch*seq_len, n_classes (log-probs) ; target -> batch*seq_len ; mask -> batch, seq_len"""
        require_gpu(pred, target, mask)
        if target.dtype != torch.int64:
            raise RuntimeError("MaskedLoss: target must be int64")
        return _MaskedNLL.apply(pred, target, mask)
