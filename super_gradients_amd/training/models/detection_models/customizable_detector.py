"""CustomizableDetector on the HIP kernels: backbone -> neck -> heads built from `{TypeName: {kwargs}}` configs.

Reference: training/models/detection_models/customizable_detector.py:30-104 (constructor, forward, BN eps/momentum
override).  The model is one SgxNetwork: a single autograd node whose backward walks heads -> neck -> backbone with the
hand-written block backwards; the NCHW fp32 batch the reference's loaders deliver is re-laid to NHWC (channels padded
to 4) by one kernel at the entrance.
"""
from typing import Optional

import torch

from .... import kernels as K
from ....common.factories import DetectionModulesFactory
from ....modules.engine import SgxNetwork
from ....modules.layers import BatchNorm
from .predict_mixin import DetectionPredictMixin


class CustomizableDetector(DetectionPredictMixin, SgxNetwork):
    def __init__(self, backbone, heads, neck=None, num_classes: int = None, bn_eps: Optional[float] = None, bn_momentum: Optional[float] = None,
                 inplace_act: Optional[bool] = True, in_channels: int = 3):
        super().__init__()
        if neck is None:
            raise NotImplementedError("CustomizableDetector on the HIP path needs a neck (YOLO-NAS / PP-YOLOE style models)")
        self.heads_params = heads
        self.bn_eps, self.bn_momentum, self.inplace_act, self.in_channels = bn_eps, bn_momentum, inplace_act, in_channels
        f = DetectionModulesFactory()
        if num_classes is not None:
            self.heads_params = f.insert_module_param(self.heads_params, "num_classes", num_classes)
        self.backbone = f.get(f.insert_module_param(backbone, "in_channels", in_channels))
        self.neck = f.get(f.insert_module_param(neck, "in_channels", self.backbone.out_channels))
        self.heads = f.get(f.insert_module_param(self.heads_params, "in_channels", self.neck.out_channels))
        self._initialize_weights(bn_eps, bn_momentum, inplace_act)
        self._init_processing_params()

    def _initialize_weights(self, bn_eps=None, bn_momentum=None, inplace_act=True):
        for m in self.modules():
            if isinstance(m, BatchNorm):
                m.eps = bn_eps if bn_eps else m.eps
                m.momentum = bn_momentum if bn_momentum else m.momentum

    def replace_head(self, new_num_classes: Optional[int] = None, new_head=None):
        """customizable_detector.py:111-122 (the fine-tuning path: load the 80-class checkpoint, then ask for the new class count).
        Heads that support it (NDFLHeads) only get new class-prediction convs, initialised from the statistics of the old ones.
        Must happen before the model is materialised in the HBM arenas (i.e. before .materialize() / the first forward)."""
        if new_num_classes is None and new_head is None:
            raise ValueError("At least one of new_num_classes, new_head must be given to replace output layer.")
        if self._materialized:
            raise RuntimeError("replace_head must be called before the model is materialized in HBM (before the first forward)")
        if new_head is not None:
            self.heads = new_head
        elif hasattr(self.heads, "replace_num_classes"):
            from ....modules.head_replacement_utils import replace_num_classes_with_random_weights

            self.heads.replace_num_classes(new_num_classes, replace_num_classes_with_random_weights)
        else:
            f = DetectionModulesFactory()
            self.heads_params = f.insert_module_param(self.heads_params, "num_classes", new_num_classes)
            self.heads = f.get(f.insert_module_param(self.heads_params, "in_channels", self.neck.out_channels))
            self._initialize_weights(self.bn_eps, self.bn_momentum, self.inplace_act)

    def get_input_channels(self) -> int:
        if hasattr(self.backbone, "get_input_channels"):
            return self.backbone.get_input_channels()
        raise NotImplementedError(f"`{type(self.backbone).__name__}` does not support `replace_input_channels`")

    def replace_input_channels(self, in_channels: int, compute_new_weights_fn=None):
        """customizable_detector.py:124-129 (what models.get(num_input_channels=...) calls, model_factory.py:253-254)."""
        if self._materialized:
            raise RuntimeError("replace_input_channels must be called before the model is materialized in HBM (before the first forward)")
        if not hasattr(self.backbone, "replace_input_channels"):
            raise NotImplementedError(f"`{type(self.backbone).__name__}` does not support `replace_input_channels`")
        self.backbone.replace_input_channels(in_channels=in_channels, compute_new_weights_fn=compute_new_weights_fn)
        self.in_channels = self.get_input_channels()
        self._initialize_weights(self.bn_eps, self.bn_momentum, self.inplace_act)

    # ---- SgxNetwork protocol -------------------------------------------------------------------------------------
    def _input_layout(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected an NCHW batch with {self.in_channels} channels, got {tuple(x.shape)}")
        return K.input_to_nhwc(x)

    def _fwd(self, x, xh=None):
        if xh is None:
            xh = self._input_layout(x)
        if self._half_inference and not self.training:
            xh = K.cast_bf16(xh, cpad=8)  # the bf16 batch the first convolution reads (3 -> 8 channels: one 16-byte lane load per pixel)
        pre = getattr(self.neck, "pre", None) if self.training else None
        feats = self.backbone.fwd(xh, on_output=pre) if pre is not None else self.backbone.fwd(xh)
        p = self.neck.fwd(feats)
        boxes, scores, logits, distri, anchors, pts, counts, strides = self.heads.fwd(p)
        self._aux = (anchors, pts, list(counts), strides)  # constants of the feature-map sizes (cached in the heads)
        self._out_shapes = (tuple(logits.shape), tuple(distri.shape))
        return boxes, scores, logits, distri

    def _differentiable_outputs(self, n):
        return [False, False, True, True]

    def _pack(self, flat):
        boxes, scores, logits, distri = flat
        anchors, pts, counts, strides = self._aux
        return (boxes, scores), (logits, distri, anchors, pts, list(counts), strides)

    def forward(self, x):
        return super().forward(x)

    def _bwd(self, d_boxes, d_scores, d_logits, d_distri):
        dev = self._device
        like_l, like_d = self._out_shapes
        if d_logits is None:
            d_logits = torch.zeros(like_l, device=dev)
        if d_distri is None:
            d_distri = torch.zeros(like_d, device=dev)
        ready = self._bucket_ready
        dps = self.heads.bwd(d_logits.contiguous(), d_distri.contiguous())
        ready("heads.")
        # (a neck with join_bwd may leave gradients of the backbone's tensors in flight on the branch stream: the backbone's walk joins
        # before it reads the first of them)
        join = getattr(self.neck, "join_bwd", None)
        dcs = self.neck.bwd(*dps, **({"late_join": True} if join is not None else {}))
        ready("neck.")
        self.backbone.bwd(dict(zip(self.backbone.out_layers, dcs)), on_layer_done=lambda layer: ready(f"backbone.{layer}."),
                          **({"ext_ready": join} if join is not None else {}))

    def gradient_buckets(self):
        """Arena ranges in backward-completion order (see training/utils/distributed_training_utils.GradientAllReducer)."""
        return [f"backbone.{layer}." for layer in self.backbone._all_layers] + ["neck.", "heads."]
