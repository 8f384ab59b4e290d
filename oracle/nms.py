"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement of the detection post-processing the reference runs after the model:
  PPYoloEPostPredictionCallback.forward      /root/reference/src/super_gradients/training/models/detection_models/pp_yolo_e/post_prediction_callback.py:42-98
  _filter_max_predictions                    .../post_prediction_callback.py:120-123
  torchvision.ops.nms / batched_nms          third-party, un-vendored (requirements.txt:12) -> oracle/nms.c restates
                                             torchvision/csrc/ops/cpu/nms_kernel.cpp; batched_nms follows
                                             torchvision/ops/boxes.py (coordinate-offset trick when boxes.numel() <= 4000
                                             on CPU, per-class loop + re-sort by score otherwise).
PARITY UNPINNED for the torchvision part (no reference test pins NMS results).
Tie rule fixed here: candidates are ordered by (score descending, original index ascending); torch.topk /
unstable sorts give the same order whenever scores are distinct.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle_nms.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_nms.restype = ctypes.c_int64
        _LIB.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    return _LIB


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    n = b.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    k = _lib().oracle_nms(b.ctypes.data, s.ctypes.data, n, ctypes.c_float(iou_threshold), keep.ctypes.data)
    return torch.from_numpy(keep[:k].copy())


def nms_python(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """Pure-numpy twin of oracle/nms.c for small cases (cross-check of the C build)."""
    n = len(scores)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(np.float32)
    sup = np.zeros(n, bool)
    keep = []
    for a, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if sup[j]:
                continue
            w = max(np.float32(0), min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]))
            h = max(np.float32(0), min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]))
            inter = np.float32(w) * np.float32(h)
            if inter / (area[i] + area[j] - inter) > np.float32(thr):
                sup[j] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float, force_vanilla: bool = False) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if boxes.numel() > 4000 or force_vanilla:  # CPU rule of torchvision/ops/boxes.py (_batched_nms_vanilla)
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for c in torch.unique(idxs):
            ci = torch.where(idxs == c)[0]
            keep_mask[ci[nms(boxes[ci], scores[ci], iou_threshold)]] = True
        ki = torch.where(keep_mask)[0]
        order = sorted(range(len(ki)), key=lambda t: (-float(scores[ki[t]]), int(ki[t])))
        return ki[torch.tensor(order, dtype=torch.long)] if len(order) else ki
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def post_prediction(pred_bboxes, pred_scores, *, score_threshold, nms_threshold, nms_top_k, max_predictions,
                    multi_label_per_box=True, class_agnostic_nms=False, force_vanilla=False):
    """[B,L,4], [B,L,C] -> list of [Ni,6] (x1,y1,x2,y2,conf,class) + list of candidate indices kept."""
    res = []
    for bx, sc in zip(pred_bboxes.float(), pred_scores.float()):
        if multi_label_per_box:
            i, j = (sc > score_threshold).nonzero(as_tuple=False).T
            conf, lab, bb = sc[i, j], j, bx[i]
        else:
            conf, lab = torch.max(sc, dim=1)
            m = conf >= score_threshold
            conf, lab, bb = conf[m], lab[m], bx[m]
        if conf.size(0) > nms_top_k:
            order = torch.tensor(sorted(range(conf.size(0)), key=lambda t: (-float(conf[t]), t))[:nms_top_k], dtype=torch.long)
            conf, lab, bb = conf[order], lab[order], bb[order]
        keep = nms(bb, conf, nms_threshold) if class_agnostic_nms else batched_nms(bb, conf, lab, nms_threshold, force_vanilla)
        out = torch.cat([bb[keep], conf[keep, None], lab[keep, None].float()], 1)
        res.append(out[:max_predictions])
    return res
