"""YoloNAS / YoloNAS_S / _M / _L (reference: yolo_nas/yolo_nas_variants.py:75-212) on the HIP kernels."""
import copy
from typing import Any, Tuple

import torch
from torch import nn

from ..... import kernels as K

from .....common.registry import register_model
from ....utils.utils import HpmStruct, get_param
from ...arch_params_factory import get_arch_params
from ..customizable_detector import CustomizableDetector


class YoloNASDecodingModule(nn.Module):
    """Pre-NMS decoding of the export / inference path (reference: yolo_nas_variants.py:25-72): keeps, per image, the
    `num_pre_nms_predictions` anchors with the highest class confidence, sorted by confidence, with their boxes and score rows.
    One launch of the post-prediction kernel (kernels.decode_topk) instead of max + topk + two flat gathers."""

    def __init__(self, num_pre_nms_predictions: int = 1000):
        super().__init__()
        self.num_pre_nms_predictions = num_pre_nms_predictions

    def infer_total_number_of_predictions(self, predictions: Any) -> int:
        pred_bboxes, _ = predictions[0]
        return pred_bboxes.size(1)

    def get_num_pre_nms_predictions(self) -> int:
        return self.num_pre_nms_predictions

    def forward(self, inputs: Tuple[Tuple[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, ...]]):
        pred_bboxes, pred_scores = inputs[0]
        boxes, scores, _ = K.decode_topk(pred_bboxes, pred_scores, self.num_pre_nms_predictions)
        return boxes, scores


class YoloNAS(CustomizableDetector):
    def __init__(self, backbone, heads, neck=None, num_classes: int = None, bn_eps=None, bn_momentum=None, inplace_act=True, in_channels: int = 3):
        super().__init__(backbone, heads, neck, num_classes, bn_eps, bn_momentum, inplace_act, in_channels)

    def get_post_prediction_callback(self, *, conf: float, iou: float, nms_top_k: int, max_predictions: int, multi_label_per_box: bool,
                                     class_agnostic_nms: bool):
        from ..pp_yolo_e.post_prediction_callback import PPYoloEPostPredictionCallback

        return PPYoloEPostPredictionCallback(score_threshold=conf, nms_threshold=iou, nms_top_k=nms_top_k, max_predictions=max_predictions,
                                             multi_label_per_box=multi_label_per_box, class_agnostic_nms=class_agnostic_nms)

    def supports_half_inference(self) -> bool:
        """predict(fp16=True) runs the fused model on the bf16 kernels of csrc/half.hip (every op of the deployment form has one)."""
        return True

    def get_decoding_module(self, num_pre_nms_predictions: int, **kwargs) -> YoloNASDecodingModule:
        return YoloNASDecodingModule(num_pre_nms_predictions)

    def get_input_shape_steps(self) -> Tuple[int, int]:
        return 32, 32

    def get_minimum_input_shape_size(self) -> Tuple[int, int]:
        return 32, 32

    @property
    def num_classes(self):
        return self.heads.num_classes


def _variant(default_name):
    def init(self, arch_params):
        merged = HpmStruct(**copy.deepcopy(get_arch_params(default_name)))
        merged.override(**(arch_params.to_dict() if hasattr(arch_params, "to_dict") else dict(arch_params or {})))
        YoloNAS.__init__(self, backbone=merged.backbone, neck=merged.neck, heads=merged.heads, num_classes=get_param(merged, "num_classes", None),
                         in_channels=get_param(merged, "in_channels", 3), bn_momentum=get_param(merged, "bn_momentum", None),
                         bn_eps=get_param(merged, "bn_eps", None), inplace_act=get_param(merged, "inplace_act", None))

    return init


@register_model("yolo_nas_s")
class YoloNAS_S(YoloNAS):
    __init__ = _variant("yolo_nas_s_arch_params")


@register_model("yolo_nas_m")
class YoloNAS_M(YoloNAS):
    __init__ = _variant("yolo_nas_m_arch_params")


@register_model("yolo_nas_l")
class YoloNAS_L(YoloNAS):
    __init__ = _variant("yolo_nas_l_arch_params")
