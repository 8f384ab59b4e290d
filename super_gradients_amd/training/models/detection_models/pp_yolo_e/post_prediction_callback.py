"""PPYoloEPostPredictionCallback on the HIP NMS kernel.

Reference: pp_yolo_e/post_prediction_callback.py:10-123 - per image: score filter (multi-label: every (anchor, class)
above the threshold; single-label: best class per anchor), top-k (nms_top_k), torchvision nms (class-agnostic) or
batched_nms (per class), rows [x1,y1,x2,y2,conf,class], at most max_predictions of them.
Here the whole batch is ONE kernel launch (one workgroup per image, candidates ranked and suppressed in LDS); the
python loop over images, the nonzero/topk/gather tensors and the per-image kernel launches of the reference are gone.
Returns the reference's structure: a list of B tensors [Ni, 6] (device tensors; one small D2H copy of the B counts).
"""
from typing import Any, List, Tuple

import torch
from torch import Tensor

from ..... import kernels as K
from ....utils.detection_utils import DetectionPostPredictionCallback


class PPYoloEPostPredictionCallback(DetectionPostPredictionCallback):
    def __init__(self, *, score_threshold: float, nms_threshold: float, nms_top_k: int, max_predictions: int, multi_label_per_box: bool = True,
                 class_agnostic_nms: bool = False):
        super().__init__()
        self.score_threshold, self.nms_threshold = score_threshold, nms_threshold
        self.nms_top_k, self.max_predictions = nms_top_k, max_predictions
        self.multi_label_per_box, self.class_agnostic_nms = multi_label_per_box, class_agnostic_nms

    @torch.no_grad()
    def forward_batched(self, outputs: Any) -> Tuple[Tensor, Tensor, Tensor]:
        """Device-resident result without any host sync: (rows [B,max_predictions,6], counts [B] int32, num_candidates [B] int32)."""
        boxes, scores = self._get_decoded_predictions_from_model_output(outputs)
        out, cnt, _, ncand = K.nms(boxes, scores, self.score_threshold, self.nms_threshold, self.nms_top_k, self.max_predictions,
                                   multi_label=self.multi_label_per_box, class_mode=0 if self.class_agnostic_nms else 3)
        return out, cnt, ncand

    @torch.no_grad()
    def forward(self, outputs: Any, device: str = None) -> List[Tensor]:
        out, cnt, _ = self.forward_batched(outputs)
        counts = cnt.tolist()
        return [out[b, :n] for b, n in enumerate(counts)]

    def _get_decoded_predictions_from_model_output(self, outputs: Any) -> Tuple[Tensor, Tensor]:
        if isinstance(outputs, tuple) and len(outputs) == 2:
            if torch.is_tensor(outputs[0]) and torch.is_tensor(outputs[1]) and outputs[0].shape[1] == outputs[1].shape[1] and outputs[0].shape[2] == 4:
                return outputs
            return outputs[0]
        raise ValueError(f"Unsupported output format: {outputs}")
