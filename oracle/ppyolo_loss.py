"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement in plain PyTorch fp32 of the reference's PPYoloELoss
(/root/reference/src/super_gradients/training/losses/ppyolo_loss.py).  Written per image
(the reference's "sequential" formulation, :854-942, which its own unit test proves equal to the
batched one, tests/unit_tests/ppyoloe_unit_test.py:42-81), with explicit index bookkeeping
instead of the reference's [B,n,L] one-hot tensors.  Autograd supplies the gradients the HIP
backward is checked against.  Pinned against the real reference by
tests/test_oracle_vs_reference.py and the fixtures made by oracle/make_golden.py.

Restated pieces (reference line numbers):
  cxcywh -> xyxy (x2 = x1 + w)                training/datasets/data_formats/bbox_formats/cxcywh.py:36-56
  target split per image / pad mask           ppyolo_loss.py:698-775
  distribution decode                         ppyolo_loss.py:1054-1061, bbox_utils.py:9-29
  batch_iou_similarity (eps 1e-9)             ppyolo_loss.py:17-35
  iou_similarity (eps 1e-10)                  ppyolo_loss.py:38-57
  check_points_inside_bboxes (eps 1e-9)       ppyolo_loss.py:178-211
  TaskAlignedAssigner                         ppyolo_loss.py:437-561
  ATSSAssigner                                ppyolo_loss.py:258-434
  GIoU loss                                   ppyolo_loss.py:564-638
  DFL                                         ppyolo_loss.py:994-1006, 1063-1067
  varifocal / focal                           ppyolo_loss.py:1069-1084
  normalisation and weights                   ppyolo_loss.py:971-988
"""
from typing import List, Tuple

import torch
import torch.nn.functional as F


def targets_to_xyxy(targets: torch.Tensor):
    cx, cy, w, h = targets[:, 2], targets[:, 3], targets[:, 4], targets[:, 5]
    x1 = cx - 0.5 * w
    y1 = cy - 0.5 * h
    return targets[:, 0], targets[:, 1].long(), torch.stack([x1, y1, x1 + w, y1 + h], -1)


def split_targets(targets: torch.Tensor, batch: int):
    """-> per image (labels [k], boxes [k,4], valid [k]) ; valid = sum(coords) > 0 (ppyolo_loss.py:754)."""
    idx, cls, box = targets_to_xyxy(targets)
    out = []
    for b in range(batch):
        m = idx == b
        bb = box[m]
        out.append((cls[m], bb, bb.sum(1) > 0))
    return out


def pair_iou(gt: torch.Tensor, boxes: torch.Tensor, eps: float):
    """gt [n,4], boxes [L,4] -> [n,L]; clip both areas at 0 (ppyolo_loss.py:17-57)."""
    g = gt[:, None, :]
    p = boxes[None, :, :]
    lt = torch.maximum(p[..., :2], g[..., :2])
    rb = torch.minimum(p[..., 2:], g[..., 2:])
    ov = (rb - lt).clip(0).prod(-1)
    a_g = (g[..., 2:] - g[..., :2]).clip(0).prod(-1)
    a_p = (p[..., 2:] - p[..., :2]).clip(0).prod(-1)
    return ov / (a_g + a_p - ov + eps)


def points_in_boxes(points: torch.Tensor, gt: torch.Tensor, eps: float = 1e-9):
    x, y = points[None, :, 0], points[None, :, 1]
    d = torch.stack([x - gt[:, None, 0], y - gt[:, None, 1], gt[:, None, 2] - x, gt[:, None, 3] - y], -1)
    return d.min(-1).values > eps


def decode_distribution(points_grid: torch.Tensor, distri: torch.Tensor):
    """distri [..., L, 4*(R+1)] -> xyxy in grid units."""
    shp = distri.shape[:-1]
    d = distri.reshape(*shp, 4, -1)
    r = d.shape[-1]
    ltrb = (torch.softmax(d, -1) * torch.linspace(0, r - 1, r)).sum(-1)
    return torch.cat([points_grid - ltrb[..., :2], points_grid + ltrb[..., 2:]], -1)


def _resolve(mask: torch.Tensor, ious: torch.Tensor):
    """mask [n,L] bool.  Anchors claimed by >1 GT go to argmax-IoU GT over ALL n rows (ppyolo_loss.py:527-538)."""
    count = mask.sum(0)
    multi = count > 1
    if multi.any():
        best = ious.argmax(0)  # first max
        onehot = F.one_hot(best, mask.shape[0]).T.bool()
        mask = torch.where(multi[None, :], onehot, mask)
        count = mask.sum(0)
    gt_index = mask.float().argmax(0)  # first 1, 0 if none
    return mask, count > 0, gt_index


def tal_assign_image(scores, boxes_px, points_px, labels, gts, valid, num_classes, topk=13, alpha=1.0, beta=6.0, eps=1e-9, sequential=False):
    """scores [L,C] (sigmoid), boxes_px [L,4], points_px [L,2], labels [n], gts [n,4], valid [n] bool.
    -> assigned label [L] (bg = num_classes), box [L,4], score [L] (value at the assigned class), gt index [L]."""
    L = scores.shape[0]
    n = gts.shape[0]
    if n == 0:
        return torch.full([L], num_classes), torch.zeros(L, 4), torch.zeros(L), torch.zeros(L, dtype=torch.long), torch.zeros(L, dtype=torch.bool)
    ious = pair_iou(gts, boxes_px, 1e-9)  # [n,L]
    cls_sc = scores[:, labels].T  # [n,L]
    metric = cls_sc.pow(alpha) * ious.pow(beta)
    inside = points_in_boxes(points_px, gts)
    k = min(topk, L)
    top_val, top_idx = torch.topk(metric * inside, k, dim=-1, largest=True)
    if sequential:
        # use_batched_assignment=False passes pad_gt_mask=None (ppyolo_loss.py:908-917): the zero-padding mask is replaced
        # by gather_topk_anchors' own gate, "best candidate metric > eps" (:224-226), and mask_positive is not masked (:525).
        valid = top_val.max(-1).values > eps
    in_top = torch.zeros(n, L, dtype=torch.bool)
    in_top.scatter_(1, top_idx, True)
    in_top &= valid[:, None]
    mask = in_top & inside & valid[:, None]
    mask, pos, gi = _resolve(mask, ious)
    a_label = torch.where(pos, labels[gi], torch.full_like(gi, num_classes))
    a_box = gts[gi]
    mm = metric * mask
    max_m = mm.max(-1, keepdim=True).values
    max_i = (ious * mask).max(-1, keepdim=True).values
    a_score = (mm / (max_m + eps) * max_i).max(0).values
    return a_label, a_box, a_score, gi, pos


def atss_assign_image(anchors, counts: List[int], pred_boxes_px, labels, gts, valid, num_classes, topk=9, sequential=False):
    L = anchors.shape[0]
    n = gts.shape[0]
    if sequential:  # pad_gt_mask=None (ppyolo_loss.py:896-905): no GT row is masked out (:294-295, :387-388, :401-402)
        valid = torch.ones_like(valid)
    if n == 0:
        return torch.full([L], num_classes), torch.zeros(L, 4), torch.zeros(L), torch.zeros(L, dtype=torch.long), torch.zeros(L, dtype=torch.bool)
    ious = pair_iou(gts, anchors, 1e-10)
    gc = torch.stack([(gts[:, 0] + gts[:, 2]) / 2, (gts[:, 1] + gts[:, 3]) / 2], -1)
    ac = torch.stack([(anchors[:, 0] + anchors[:, 2]) / 2, (anchors[:, 1] + anchors[:, 3]) / 2], -1)
    dist = torch.norm(gc[:, None, :] - ac[None, :, :], p=2, dim=-1)
    in_top = torch.zeros(n, L, dtype=torch.bool)
    cand = []
    off = 0
    for c in counts:
        _, idx = torch.topk(dist[:, off:off + c], topk, dim=-1, largest=False)
        idx = idx + off
        in_top.scatter_(1, idx, True)
        cand.append(idx)
        off += c
    cand = torch.cat(cand, -1)  # [n, levels*topk]
    in_top &= valid[:, None]
    iou_c = ious * in_top
    thr_src = torch.gather(iou_c, 1, cand)
    thr = thr_src.mean(-1, keepdim=True) + thr_src.std(-1, keepdim=True)
    in_top = in_top & (iou_c > thr)
    mask = in_top & points_in_boxes(ac, gts) & valid[:, None]
    mask, pos, gi = _resolve(mask, ious)
    a_label = torch.where(pos, labels[gi], torch.full_like(gi, num_classes))
    a_box = gts[gi]
    a_score = (pair_iou(gts, pred_boxes_px, 1e-9) * mask).max(0).values
    return a_label, a_box, a_score, gi, pos


def giou_loss(p: torch.Tensor, g: torch.Tensor, eps: float = 1e-10):
    x1, y1, x2, y2 = p.unbind(-1)
    x1g, y1g, x2g, y2g = g.unbind(-1)
    w = (torch.minimum(x2, x2g) - torch.maximum(x1, x1g)).clip(0)
    h = (torch.minimum(y2, y2g) - torch.maximum(y1, y1g)).clip(0)
    ov = w * h
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - ov + eps
    iou = ov / union
    hull = (torch.maximum(x2, x2g) - torch.minimum(x1, x1g)) * (torch.maximum(y2, y2g) - torch.minimum(y1, y1g)) + eps
    return 1 - (iou - (hull - union) / hull)


def dfl_loss(dist_logits: torch.Tensor, target: torch.Tensor):
    """dist_logits [P,4,R+1], target [P,4] in [0, R-0.01] -> [P]"""
    tl = target.long()
    wl = (tl + 1).float() - target
    lp = F.log_softmax(dist_logits, -1)
    left = -lp.gather(-1, tl[..., None]).squeeze(-1) * wl
    right = -lp.gather(-1, (tl + 1)[..., None]).squeeze(-1) * (1 - wl)
    return (left + right).mean(-1)


class PPYoloELossOracle:
    def __init__(self, num_classes, use_varifocal_loss=True, use_static_assigner=True,
                 classification_loss_weight=1.0, iou_loss_weight=2.5, dfl_loss_weight=0.5, use_batched_assignment=True):
        self.sequential = not use_batched_assignment
        self.nc = num_classes
        self.vfl = use_varifocal_loss
        self.static = use_static_assigner
        self.w = (classification_loss_weight, iou_loss_weight, dfl_loss_weight)

    def assign(self, predictions, targets):
        logits, distri, anchors, points, counts, strides = predictions
        B, L, _ = logits.shape
        pts_grid = points / strides
        boxes = decode_distribution(pts_grid, distri)
        per_img = split_targets(targets, B)
        out = []
        with torch.no_grad():
            for b, (lab, gts, valid) in enumerate(per_img):
                if self.static:
                    out.append(atss_assign_image(anchors, counts, boxes[b] * strides, lab, gts, valid, self.nc, sequential=self.sequential))
                else:
                    out.append(tal_assign_image(logits[b].sigmoid(), boxes[b] * strides, points, lab, gts, valid, self.nc, sequential=self.sequential))
        a_label = torch.stack([o[0] for o in out])
        a_box = torch.stack([o[1] for o in out])
        a_score = torch.stack([o[2] for o in out])
        return boxes, a_label, a_box, a_score

    def sums(self, predictions, targets) -> Tuple[torch.Tensor, ...]:
        logits, distri, anchors, points, counts, strides = predictions
        boxes, a_label, a_box, a_score = self.assign(predictions, targets)
        B, L, C = logits.shape
        pos = a_label != self.nc
        onehot = F.one_hot(a_label, C + 1)[..., :C].float()
        t = onehot * a_score[..., None]
        bce = F.binary_cross_entropy_with_logits(logits, t, reduction="none")
        p = logits.sigmoid()
        if self.vfl:
            wgt = 0.75 * p.pow(2.0) * (1 - onehot) + t * onehot
        else:
            wgt = (p - t).pow(2.0)
            if self.static:
                wgt = wgt * (0.25 * t + 0.75 * (1 - t))
        cls_sum = (wgt * bce).sum()
        score_sum = t.sum()
        if pos.any():
            wpos = a_score[pos]
            iou_sum = (giou_loss(boxes[pos], (a_box / strides)[pos]) * wpos).sum()
            pg = (points / strides).expand(B, L, 2)[pos]
            gb = (a_box / strides)[pos]
            R = distri.shape[-1] // 4 - 1
            ltrb = torch.cat([pg - gb[:, :2], gb[:, 2:] - pg], -1).clip(0, R - 0.01)
            dfl_sum = (dfl_loss(distri[pos].reshape(-1, 4, R + 1), ltrb) * wpos).sum()
        else:
            iou_sum = torch.zeros([])
            dfl_sum = distri.sum() * 0.0
        return cls_sum, iou_sum, dfl_sum, score_sum

    def __call__(self, outputs, targets):
        predictions = outputs[1] if (isinstance(outputs, tuple) and len(outputs) == 2) else outputs
        c, i, d, s = self.sums(predictions, targets)
        s = torch.clip(s, min=1.0)
        lc, li, ld = self.w[0] * c / s, self.w[1] * i / s, self.w[2] * d / s
        loss = lc + li + ld
        return loss, torch.stack([lc.detach(), li.detach(), ld.detach(), loss.detach()])
