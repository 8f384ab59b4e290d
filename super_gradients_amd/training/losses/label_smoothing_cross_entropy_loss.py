"""CrossEntropyLoss (reference: training/losses/label_smoothing_cross_entropy_loss.py:32-111) as one fused softmax-CE forward + backward
kernel: F.cross_entropy with per-class weights / ignore_index when there is no smoothing, the reference's own smoothed form (weights
multiply the log-softmax, rows whose label is ignore_index >= 0 are masked, "mean" divides by the rows that are left) otherwise.
forward returns the loss tensor; the Trainer derives the logging item the reference's class returns next to it (`loss.unsqueeze(0)`)."""
import torch
from torch import nn

from ... import kernels as K
from ...common.registry import register_loss


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, cfg):
        smoothing, weight, ignore_index, reduction = cfg
        loss, dlogits, inv = K.softmax_ce(logits, labels, smoothing, weight, ignore_index, reduction)
        ctx.save_for_backward(dlogits, inv)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        dlogits, inv = ctx.saved_tensors
        return K.scale_by_device_scalar(dlogits, g.reshape(1).contiguous(), inv), None, None


@register_loss(name="CrossEntropyLoss", deprecated_name="cross_entropy")
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, ignore_index: int = -100, reduction: str = "mean", smooth_eps: float = None, smooth_dist=None,
                 from_logits: bool = True, label_smoothing: float = None):
        super().__init__()
        if reduction not in ("mean", "sum"):
            raise NotImplementedError("CrossEntropyLoss on the HIP path: reduction 'mean' or 'sum'")
        if smooth_dist is not None or not from_logits:
            raise NotImplementedError("CrossEntropyLoss on the HIP path: integer class targets on logits (no smooth_dist, from_logits=True)")
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32))
        self.ignore_index, self.reduction = int(ignore_index), reduction
        eps = smooth_eps if smooth_eps is not None else label_smoothing  # `label_smoothing`: the nn.CrossEntropyLoss spelling
        self.smooth_eps = float(eps or 0.0)

    @property
    def label_smoothing(self):
        return self.smooth_eps

    def forward(self, input, target):
        w = self.weight.to(input.device) if self.weight is not None else None
        return _CEFn.apply(input, target.to(input.device), (self.smooth_eps, w, self.ignore_index, self.reduction))
