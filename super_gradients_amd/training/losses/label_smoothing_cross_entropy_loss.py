"""CrossEntropyLoss (reference: training/losses/label_smoothing_cross_entropy_loss.py:86-111 - nn.CrossEntropyLoss with
optional label smoothing, mean reduction) as one fused softmax-CE forward+backward kernel."""
import torch
from torch import nn

from ... import kernels as K
from ...common.registry import register_loss


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, smoothing):
        loss, dlogits = K.softmax_ce(logits, labels, smoothing)
        ctx.save_for_backward(dlogits)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return K.scale_by_device_scalar(dlogits, g.reshape(1).contiguous()), None, None


@register_loss(name="CrossEntropyLoss", deprecated_name="cross_entropy")
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, ignore_index: int = -100, reduction: str = "mean", label_smoothing: float = 0.0):
        super().__init__()
        if weight is not None or ignore_index != -100 or reduction != "mean":
            raise NotImplementedError("CrossEntropyLoss on the HIP path: no class weights / ignore_index, mean reduction")
        self.label_smoothing = label_smoothing

    def forward(self, input, target):
        return _CEFn.apply(input, target.to(input.device), float(self.label_smoothing))
