"""PPYoloE / PPYoloE_S / _M / _L / _X on the HIP kernels (reference: pp_yolo_e/pp_yolo_e.py:95-439).

CSPResNetBackbone -> PPYoloECSPPAN -> PPYOLOEHead as ONE autograd node (modules/engine.py:SgxNetwork): training returns the
reference's raw 6-tuple (cls_logits [B,L,C], reg_distri [B,L,68], anchors, anchor_points, num_anchors_list, stride_tensor), eval
returns ((pred_bboxes, pred_scores), raw) - exactly what PPYoloELoss / PPYoloEPostPredictionCallback consume.
"""
import copy
from typing import Tuple

import torch

from ..... import kernels as K
from .....common.registry import register_model
from .....modules.engine import SgxNetwork
from ....utils.utils import HpmStruct
from ...arch_params_factory import get_arch_params
from ..csp_resnet import CSPResNetBackbone
from ..predict_mixin import DetectionPredictMixin
from .pan import PPYoloECSPPAN
from .post_prediction_callback import PPYoloEPostPredictionCallback
from .pp_yolo_head import PPYOLOEHead


class PPYoloEDecodingModule(torch.nn.Module):
    """Pre-NMS decoding of the export / inference path (reference: pp_yolo_e.py:31-97): per image the `num_pre_nms_predictions` anchors
    with the highest class confidence, sorted by confidence, with their boxes and score rows (kernels.decode_topk)."""

    def __init__(self, num_pre_nms_predictions: int = 1000):
        super().__init__()
        self.num_pre_nms_predictions = num_pre_nms_predictions

    def get_num_pre_nms_predictions(self) -> int:
        return self.num_pre_nms_predictions

    def infer_total_number_of_predictions(self, predictions) -> int:
        pred_bboxes, _ = predictions[0]
        return pred_bboxes.size(1)

    def forward(self, inputs):
        pred_bboxes, pred_scores = inputs[0]
        boxes, scores, _ = K.decode_topk(pred_bboxes, pred_scores, self.num_pre_nms_predictions)
        return boxes, scores


class PPYoloE(DetectionPredictMixin, SgxNetwork):
    def __init__(self, arch_params):
        super().__init__()
        if isinstance(arch_params, HpmStruct):
            arch_params = arch_params.to_dict()
        arch_params = copy.deepcopy(dict(arch_params))
        self.backbone = CSPResNetBackbone(**arch_params["backbone"], depth_mult=arch_params["depth_mult"], width_mult=arch_params["width_mult"])
        self.neck = PPYoloECSPPAN(**arch_params["neck"], depth_mult=arch_params["depth_mult"], width_mult=arch_params["width_mult"])
        self.head = PPYOLOEHead(**arch_params["head"], width_mult=arch_params["width_mult"], num_classes=arch_params["num_classes"])
        self.in_channels = 3
        self._init_processing_params()

    def get_post_prediction_callback(self, *, conf: float, iou: float, nms_top_k: int, max_predictions: int, multi_label_per_box: bool,
                                     class_agnostic_nms: bool) -> PPYoloEPostPredictionCallback:
        return PPYoloEPostPredictionCallback(score_threshold=conf, nms_threshold=iou, nms_top_k=nms_top_k, max_predictions=max_predictions,
                                             multi_label_per_box=multi_label_per_box, class_agnostic_nms=class_agnostic_nms)

    def supports_half_inference(self) -> bool:
        """predict(fp16=True) runs the fused model on the bf16 kernels of csrc/half.hip (round 6: the squeeze-excitation means / gates and the
        nearest up-sampling of this family have bf16 forms; RepVGG blocks run as one fused 3x3 convolution each).  The bf16 kernels move 16-byte
        lanes - 8 channels: every convolution's input (the RGB stem's aside, which is padded 3 -> 8) must have a multiple of 8 channels.  The S
        and L widths do; M (width 0.75: 36-channel halves) and X (1.25: 60) do not - for them predict(fp16=True) keeps the fp32 path, loudly."""
        from .....modules.layers import ConvLayer

        return all(m.in_channels % 8 == 0 or m.in_channels == self.in_channels for m in self.modules() if isinstance(m, ConvLayer))

    def get_decoding_module(self, num_pre_nms_predictions: int, **kwargs) -> PPYoloEDecodingModule:
        return PPYoloEDecodingModule(num_pre_nms_predictions=num_pre_nms_predictions)

    def get_input_shape_steps(self) -> Tuple[int, int]:
        return 32, 32

    def get_minimum_input_shape_size(self) -> Tuple[int, int]:
        return 32, 32

    def get_input_channels(self) -> int:
        return self.backbone.get_input_channels()

    def get_finetune_lr_dict(self, lr: float):
        return {"head": lr, "default": 0}

    @property
    def num_classes(self):
        return self.head.num_classes

    def replace_head(self, new_num_classes=None, new_head=None):
        """pp_yolo_e.py:379-385; before the model is materialised in the HBM arenas."""
        if new_num_classes is None and new_head is None:
            raise ValueError("At least one of new_num_classes, new_head must be given to replace output layer.")
        if self._materialized:
            raise RuntimeError("replace_head must be called before the model is materialized in HBM (before the first forward)")
        if new_head is not None:
            self.head = new_head
        else:
            self.head.replace_num_classes(new_num_classes)

    def prep_model_for_conversion(self, input_size=None, **kwargs):
        """RepVGG blocks -> single 3x3 convs, anchors cached for `input_size` (reference :358-377)."""
        if input_size is not None:
            self.head.cache_anchors(input_size[-2:])
        return super().prep_model_for_conversion(input_size, **kwargs)

    # ---- SgxNetwork protocol -------------------------------------------------------------------------------------
    def _fwd(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected an NCHW batch with {self.in_channels} channels, got {tuple(x.shape)}")
        xh = K.input_to_nhwc(x)
        if self._half_inference and not self.training:
            xh = K.cast_bf16(xh, cpad=8)  # the bf16 batch the stem reads (3 -> 8 channels: one 16-byte lane load per pixel)
        boxes, scores, logits, distri, anchors, pts, counts, strides = self.head.fwd(self.neck.fwd(self.backbone.fwd(xh)))
        self._aux = (anchors, pts, list(counts), strides)
        self._out_shapes = (tuple(logits.shape), tuple(distri.shape))
        return (logits, distri) if boxes is None else (boxes, scores, logits, distri)

    def _differentiable_outputs(self, n):
        return [True, True] if n == 2 else [False, False, True, True]

    def _pack(self, flat):
        anchors, pts, counts, strides = self._aux
        if len(flat) == 2:
            return flat[0], flat[1], anchors, pts, list(counts), strides
        boxes, scores, logits, distri = flat
        return (boxes, scores), (logits, distri, anchors, pts, list(counts), strides)

    def _bwd(self, d_logits, d_distri):
        dev = self._device
        like_l, like_d = self._out_shapes
        if d_logits is None:
            d_logits = torch.zeros(like_l, device=dev)
        if d_distri is None:
            d_distri = torch.zeros(like_d, device=dev)
        ready = self._bucket_ready
        dps = self.head.bwd(d_logits.contiguous(), d_distri.contiguous())
        ready("head.")
        dcs = self.neck.bwd(*dps)
        ready("neck.")
        self.backbone.bwd(dcs, on_layer_done=lambda layer: ready(f"backbone.{layer}."))

    def gradient_buckets(self):
        """Arena ranges in backward-completion order (training/utils/distributed_training_utils.GradientAllReducer)."""
        return ["backbone.stem."] + [f"backbone.stages.{i}." for i in range(len(self.backbone.stages))] + ["neck.", "head."]


def _variant(default_name):
    def init(self, arch_params=None):
        if isinstance(arch_params, HpmStruct):
            arch_params = arch_params.to_dict()
        PPYoloE.__init__(self, get_arch_params(default_name, overriding_params=dict(arch_params or {})))

    return init


@register_model("ppyoloe_s")
class PPYoloE_S(PPYoloE):
    __init__ = _variant("ppyoloe_s_arch_params")


@register_model("ppyoloe_m")
class PPYoloE_M(PPYoloE):
    __init__ = _variant("ppyoloe_m_arch_params")


@register_model("ppyoloe_l")
class PPYoloE_L(PPYoloE):
    __init__ = _variant("ppyoloe_l_arch_params")


@register_model("ppyoloe_x")
class PPYoloE_X(PPYoloE):
    __init__ = _variant("ppyoloe_x_arch_params")
