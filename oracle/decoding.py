"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of the reference's pre-NMS decoding modules.

Follows YoloNASDecodingModule.forward (training/models/detection_models/yolo_nas/yolo_nas_variants.py:53-72) and
PPYoloEDecodingModule.forward (pp_yolo_e/pp_yolo_e.py:57-84) - the two bodies are the same arithmetic: class confidence
= max over classes, top-k anchors per image sorted by confidence, gather of the box and score rows.

Tie rule: torch.topk leaves the order of equal confidences unspecified (ATen CPU: whatever partial sort leaves; CUDA: radix select);
the restatement fixes it to "lower anchor index first" with a stable descending sort - identical to torch.topk whenever the
confidences are distinct, which is what tests/test_decoding.py pins against the reference's own modules.
"""
import torch


def decode_topk(pred_bboxes: torch.Tensor, pred_scores: torch.Tensor, k: int):
    conf, _ = torch.max(pred_scores, dim=2)                                  # [B, L]     (yolo_nas_variants.py:62)
    order = torch.sort(conf, dim=1, descending=True, stable=True).indices     # topk(sorted=True) with the tie rule fixed (:63)
    idx = order[:, :k]
    B, _, C = pred_scores.shape
    boxes = torch.gather(pred_bboxes, 1, idx[:, :, None].expand(B, k, pred_bboxes.shape[2]))   # (:65-69)
    scores = torch.gather(pred_scores, 1, idx[:, :, None].expand(B, k, C))                      # (:70)
    return boxes, scores, idx
