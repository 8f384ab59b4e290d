"""PPYoloELoss on the HIP loss kernels.

Reference: training/losses/ppyolo_loss.py:641-992 - TaskAligned (use_static_assigner=False, :437-561) or ATSS (:258-434)
assignment, varifocal|focal classification loss (:1069-1084), GIoU (:564-638) and DFL (:994-1067) on the positives,
each divided by clip(sum of assigned scores, 1) and weighted 1.0 / 2.5 / 0.5; forward returns
(loss, stack[cls, iou, dfl, loss].detach()) and `component_names`.

One C-ABI call (sgx_ppyoloe_loss_fwd) does assignment, the four sums AND the gradients of the weighted sums with respect
to cls_logits / reg_distri: the loss is linear in the sums, so backward is a single scale by upstream/normaliser.  No host
synchronisation anywhere (the reference has three).  Under data parallelism the four sums are
all-reduced as ONE 16-byte collective (reference: four, ppyolo_loss.py:971-977), with the same `/= world_size` on the
score sum that cancels the gradient averaging.
"""
import warnings
from typing import Tuple

import torch
from torch import Tensor, nn

from ... import kernels as K
from ...common.registry import register_loss


class _PPYoloELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, distri, anchors, points, strides, targets, counts, cfg):
        static, vfl, w, world, sequential = cfg
        out = K.ppyoloe_loss_fwd(logits, distri, anchors, points, strides, targets, counts, static, vfl, w, sequential)
        sums = out["sums"]
        from ..utils.distributed_training_utils import collectives_active

        if collectives_active():  # (more than one rank; or the single-rank communicator of the RCCL test, where the sum is an identity)
            torch.distributed.all_reduce(sums, op=torch.distributed.ReduceOp.SUM)
        items, inv = K.ppyoloe_loss_finalize(sums, w, float(world))
        ctx.save_for_backward(out["g_logits"], out["g_distri"], inv)
        ctx.assignment = (out["label"], out["box"], out["score"])
        loss = items[3].clone()
        ctx.mark_non_differentiable(items)
        return loss, items

    @staticmethod
    def backward(ctx, g_loss, _g_items):
        g_logits, g_distri, inv = ctx.saved_tensors
        g = g_loss.reshape(1).contiguous()
        return K.scale_by_device_scalar(g_logits, inv, g), K.scale_by_device_scalar(g_distri, inv, g), None, None, None, None, None, None


@register_loss(name="PPYoloELoss", deprecated_name="ppyoloe_loss")
class PPYoloELoss(nn.Module):
    def __init__(self, num_classes: int, use_varifocal_loss: bool = True, use_static_assigner: bool = True, reg_max=None,
                 classification_loss_weight: float = 1.0, iou_loss_weight: float = 2.5, dfl_loss_weight: float = 0.5,
                 use_batched_assignment: bool = True):
        if reg_max is not None:
            warnings.warn("A reg_max argument is not needed for PPYoloE loss anymore. It is inferred from the model's outputs.", DeprecationWarning)
        super().__init__()
        self.use_varifocal_loss = use_varifocal_loss
        self.classification_loss_weight, self.iou_loss_weight, self.dfl_loss_weight = classification_loss_weight, iou_loss_weight, dfl_loss_weight
        self.use_static_assigner = use_static_assigner
        self.num_classes = num_classes
        self.reg_max = reg_max
        # The kernels implement assignment once, per (image, GT) workgroup.  The flag only selects the reference's
        # masking rule: batched = zero-padding mask (sum(coords) > 0, ppyolo_loss.py:754); sequential = pad_gt_mask None,
        # i.e. TAL keeps a GT iff its best candidate metric > 1e-9 (:224-226).  They coincide on the reference's own unit
        # test (tests/unit_tests/ppyoloe_unit_test.py:42-81) and differ when every candidate of a GT has a vanishing metric.
        self.use_batched_assignment = use_batched_assignment

    def forward(self, outputs, targets: Tensor) -> Tuple[Tensor, Tensor]:
        predictions = outputs[1] if (isinstance(outputs, tuple) and len(outputs) == 2) else outputs
        logits, distri, anchors, points, counts, strides = predictions
        if logits.shape[-1] != self.num_classes:
            raise ValueError(f"model predicts {logits.shape[-1]} classes, loss was built for {self.num_classes}")
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        cfg = (bool(self.use_static_assigner), bool(self.use_varifocal_loss),
               (float(self.classification_loss_weight), float(self.iou_loss_weight), float(self.dfl_loss_weight)), world, not self.use_batched_assignment)
        targets = targets.to(logits.device, non_blocking=True).float()
        loss, items = _PPYoloELossFn.apply(logits, distri, anchors, points, strides, targets, [int(c) for c in counts], cfg)
        return loss, items

    @property
    def component_names(self):
        return ["loss_cls", "loss_iou", "loss_dfl", "loss"]
