"""TEST INFRASTRUCTURE ONLY (the parity oracle) - never imported by the product package.

CPU restatement in numpy loops of the reference's detection matching and AP computation
(/root/reference/src/super_gradients/training/utils/detection_utils.py):
  cxcywh2xyxy :725-736, change_bbox_bounds_for_image_size_inplace :174-184, box_iou :257-276, crowd_ioa :797-810,
  get_top_k_idx_per_cls :1342-1358, IoUMatching.compute_targets / compute_crowd_targets :902-1005,
  compute_img_detection_matching :1195-1290, compute_detection_metrics(_per_cls) :1361-1580.
Pinned against the real reference (tests/test_detection_metrics.py: live through oracle/ref_shim.py, and tests/golden/detection_metrics.pt).
"""
import numpy as np

F = np.float32


def _xyxy(rows, denorm, W, H):
    """rows [n,5] = (cls, cx, cy, w, h) -> boxes [n,4] float32 with the reference's operation order."""
    cx, cy, w, h = (rows[:, i].astype(F) for i in (1, 2, 3, 4))
    y1 = cy - h * F(0.5)
    x1 = cx - w * F(0.5)
    y2 = h + y1
    x2 = w + x1
    b = np.stack([x1, y1, x2, y2], 1).astype(F)
    if denorm:
        b[:, [0, 2]] *= F(W)
        b[:, [1, 3]] *= F(H)
    return b


def _iou(p, t):
    a1 = (p[2] - p[0]) * (p[3] - p[1])
    a2 = (t[2] - t[0]) * (t[3] - t[1])
    w = max(min(p[2], t[2]) - max(p[0], t[0]), F(0))
    h = max(min(p[3], t[3]) - max(p[1], t[1]), F(0))
    inter = F(w) * F(h)
    with np.errstate(divide="ignore", invalid="ignore"):
        return F(inter) / F(F(a1 + a2) - inter)


def match_image(preds, targets, crowd, H, W, thresholds, top_k=100, denormalize=False):
    """preds [n,6] (x1,y1,x2,y2,score,cls), targets / crowd [m,5] (cls,cx,cy,w,h) -> matched, ignore bool [n,nthr]."""
    thr = np.asarray(thresholds, dtype=F)
    n, nthr = len(preds), len(thr)
    matched = np.zeros((n, nthr), bool)
    ignore = np.ones((n, nthr), bool)
    if n == 0:
        return matched, ignore
    preds = np.asarray(preds, dtype=F).copy()
    score, cls = preds[:, 4], preds[:, 5]
    order = sorted(range(n), key=lambda i: (-float(score[i]), i))
    rank_in_cls = {}
    used = np.zeros(n, bool)
    for i in order:
        r = rank_in_cls.get(float(cls[i]), 0)
        rank_in_cls[float(cls[i])] = r + 1
        used[i] = r < top_k and score[i] != 0
    ignore[used] = False
    if len(targets) or len(crowd):
        preds[:, [0, 2]] = preds[:, [0, 2]].clip(0, W)
        preds[:, [1, 3]] = preds[:, [1, 3]].clip(0, H)
    tb = _xyxy(np.asarray(targets, dtype=F).reshape(-1, 5), denormalize, W, H)
    tc = np.asarray(targets, dtype=F).reshape(-1, 5)[:, 0]
    for j in range(nthr):
        taken = np.zeros(len(tb), bool)
        for i in order:
            if not used[i]:
                continue
            best, bi = F(-1), -1
            for t in range(len(tb)):
                if tc[t] != cls[i] or taken[t]:
                    continue
                v = _iou(preds[i, :4], tb[t])
                if v > best:
                    best, bi = v, t
            if bi >= 0 and best > thr[0] and best > thr[j]:
                taken[bi] = True
                matched[i, j] = True
    cb = _xyxy(np.asarray(crowd, dtype=F).reshape(-1, 5), denormalize, W, H)
    cc = np.asarray(crowd, dtype=F).reshape(-1, 5)[:, 0]
    if len(cb):
        for i in range(n):
            if not used[i]:
                continue
            p = preds[i, :4]
            area = (p[2] - p[0]) * (p[3] - p[1])
            best = None
            for t in range(len(cb)):
                v = F(0)
                if cc[t] == cls[i]:
                    w = max(min(p[2], cb[t, 2]) - max(p[0], cb[t, 0]), F(0))
                    h = max(min(p[3], cb[t, 3]) - max(p[1], cb[t, 1]), F(0))
                    with np.errstate(divide="ignore", invalid="ignore"):
                        v = F(F(w) * F(h)) / F(area)
                best = v if best is None or v > best else best
            ignore[i] |= best > thr
    return matched, ignore


def match_batch(pred_list, targets, crowd, H, W, thresholds, top_k=100, denormalize=False):
    """pred_list: list of [n_i,6] arrays (or None); targets / crowd flat [T,6] (img, cls, cx, cy, w, h)."""
    targets = np.asarray(targets, dtype=F).reshape(-1, 6)
    crowd = np.asarray(crowd, dtype=F).reshape(-1, 6)
    out = []
    for b, p in enumerate(pred_list):
        p = np.zeros((0, 6), F) if p is None else np.asarray(p, dtype=F)
        out.append(match_image(p, targets[targets[:, 0] == b, 1:], crowd[crowd[:, 0] == b, 1:], H, W, thresholds, top_k, denormalize))
    return out


def average_precision(matched, ignore, scores, pred_cls, target_cls, recall_thresholds=None, score_threshold=0.1):
    """-> dict(ap [ncls,nthr], precision, recall, f1, classes, best_score_threshold, best_per_class) in float32 like the reference."""
    matched, ignore = np.asarray(matched, bool), np.asarray(ignore, bool)
    scores, pred_cls, target_cls = np.asarray(scores, F), np.asarray(pred_cls), np.asarray(target_cls)
    import torch  # the reference builds both grids with torch.linspace (float32 steps): numpy's linspace rounds differently

    rt = torch.linspace(0, 1, 101).numpy() if recall_thresholds is None else np.asarray(recall_thresholds, F)
    grid = torch.linspace(0, 1, len(rt)).numpy()
    classes = np.unique(target_cls)
    nthr = matched.shape[1]
    ap = np.zeros((len(classes), nthr), F)
    prec, rec = np.zeros_like(ap), np.zeros_like(ap)
    f1c = np.zeros((len(classes), len(grid)), F)
    best_cls = np.zeros(len(classes), F)
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        tp, fp, sc = matched[sel], (~matched[sel]) & (~ignore[sel]), scores[sel]
        nt = int((target_cls == c).sum())
        if len(sc) == 0:
            continue
        # rank order among EQUAL scores is whatever torch.argsort(descending=True) (not stable) yields in the reference (:1502): use the same
        # routine, otherwise tied true/false positives swap places and the AP moves by one recall sample
        o = torch.argsort(torch.from_numpy(sc.copy()), descending=True).numpy()
        tp, fp, sc = tp[o], fp[o], sc[o]
        ctp, cfp = np.cumsum(tp, 0, dtype=F), np.cumsum(fp, 0, dtype=F)
        r = (ctp / F(nt)).astype(F)
        p = (ctp / (ctp + cfp + F(np.finfo(np.float64).eps))).astype(F)
        p = np.maximum.accumulate(p[::-1], 0)[::-1]
        k = int(np.searchsorted(-sc, -F(score_threshold), side="right"))
        if k > 0:
            rec[ci], prec[ci] = r[k - 1], p[k - 1]
        idx = np.searchsorted(-sc, -grid, side="right")
        rp = np.concatenate([np.zeros((1, nthr), F), r])[idx]
        pp = np.concatenate([np.zeros((1, nthr), F), p])[idx]
        f1c[ci] = (2 * rp * pp / (rp + pp + F(1e-16))).mean(1)
        best_cls[ci] = grid[int(np.argmax(f1c[ci]))]
        pz = np.concatenate([p, np.zeros((1, nthr), F)])
        for j in range(nthr):
            ap[ci, j] = pz[np.searchsorted(r[:, j], rt, side="left"), j].mean()
    f1 = 2 * prec * rec / (prec + rec + F(1e-16))
    best = grid[int(np.argmax(f1c.mean(0)))] if len(classes) else F(0)
    return dict(ap=ap, precision=prec, recall=rec, f1=f1, classes=classes, best_score_threshold=best, best_per_class=best_cls)
