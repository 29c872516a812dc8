"""MI355X-native mirror of the reference's ``loss.MaskedLoss`` (reference loss.py:6-25).

``MaskedLoss(losser, weight=None)``; ``forward(pred [B*L,C], target [B*L], mask [B,L])`` with ``losser`` one of the two the
reference trainer builds (model_trainer.py:74-77): ``nn.NLLLoss`` or ``nn.CrossEntropyLoss``, both with ``reduction='sum'`` on
``pred * mask``, divided by ``sum(mask)`` (or ``sum(weight[target] * mask)`` with class weights).  ``pred`` holds
log-probabilities (the model ends in log_softmax); CrossEntropyLoss re-applies log_softmax to ``pred * mask``: the identity on a
valid row, but a masked row contributes ``weight[y] * log(C)`` -- reproduced here, because it is what the reference reports with
its default ``--loss CrossEntropy`` (train.py:117) on padded batches.  One fused HIP kernel pair serves all four combinations.
"""
import torch
import torch.nn as nn

from mser import ops
from mser.autograd import require_gpu


class _MaskedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, weight, is_ce):
        pred = pred.contiguous()
        target = target.contiguous()
        mask = mask.contiguous().float().view(-1)
        out = torch.empty(2, device=pred.device)
        ops.masked_loss_fwd(pred, target, mask, weight, is_ce, out)
        ctx.save_for_backward(pred, target, mask, out)
        ctx.weight, ctx.is_ce = weight, is_ce
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, out = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        ops.masked_loss_bwd(pred, target, mask, ctx.weight, ctx.is_ce, out, g.contiguous().view(1), dpred)
        return dpred, None, None, None, None


class MaskedLoss(nn.Module):

    def __init__(self, losser, weight=None):
        super(MaskedLoss, self).__init__()
        if losser not in (nn.NLLLoss, nn.CrossEntropyLoss):
            raise ValueError("MaskedLoss: losser must be nn.NLLLoss or nn.CrossEntropyLoss (model_trainer.py:74-77)")
        self.weight = weight
        self.is_ce = losser is nn.CrossEntropyLoss
        self._w = None

    def _weight_on(self, device):
        if self.weight is None:
            return None
        if self._w is None or self._w.device != device:
            self._w = torch.as_tensor(self.weight, dtype=torch.float32).to(device).contiguous()
        return self._w

    def forward(self, pred, target, mask):
        """pred -> batch*seq_len, n_classes (log-probs) ; target -> batch*seq_len ; mask -> batch, seq_len"""
        require_gpu(pred, target, mask)
        if target.dtype != torch.int64:
            raise RuntimeError("MaskedLoss: target must be int64")
        return _MaskedLossFn.apply(pred, target, mask, self._weight_on(pred.device), self.is_ce)
