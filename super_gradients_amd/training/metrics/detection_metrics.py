"""DetectionMetrics (Precision / Recall / F1 / mAP) with the prediction-to-target matching on the MI355X.

Reference: training/metrics/detection_metrics.py:26-330 (class, metric names, update/compute protocol, DDP state gathering) and
training/utils/detection_utils.py: IouThreshold :231-254, compute_detection_matching :1120-1290 (-> sgx_detection_match, one
launch per validation batch instead of a Python loop over images and candidate pairs), compute_detection_metrics(_per_cls)
:1361-1580 (once per epoch over the accumulated flags: host-side tensor arithmetic here as well).
"""
import collections
from enum import Enum
from typing import Dict, List, Optional, Tuple, Union

import torch

from ... import kernels as K
from ...common.registry import register_metric


class IouThreshold(tuple, Enum):
    MAP_05 = (0.5, 0.5)
    MAP_05_TO_095 = (0.5, 0.95)

    def is_range(self):
        return self[0] != self[1]

    def to_tensor(self):
        return self.from_bounds(self[0], self[1], step=0.05) if self.is_range() else torch.tensor([self[0]])

    @classmethod
    def from_bounds(cls, low: float, high: float, step: float = 0.05) -> torch.Tensor:
        return torch.linspace(low, high, int(round((high - low) / step)) + 1)


def _metrics_one_class(tps: torch.Tensor, fps: torch.Tensor, scores: torch.Tensor, n_targets: int, recall_thresholds: torch.Tensor,
                       score_threshold: float, score_grid: torch.Tensor):
    """One class: AP / precision / recall per IoU threshold, mean-F1 curve over the score grid and its arg-max
    (detection_utils.py:1449-1580).  tps/fps bool [n, nthr], scores [n]."""
    nthr = tps.shape[1]
    zeros = torch.zeros(nthr)
    if tps.shape[0] == 0:
        return zeros, zeros.clone(), zeros.clone(), torch.zeros(len(score_grid)), torch.tensor(0.0)
    order = torch.argsort(scores, descending=True)
    tps, fps, scores = tps[order], fps[order], scores[order].contiguous()
    ctp = torch.cumsum(tps, 0, dtype=torch.float)
    cfp = torch.cumsum(fps, 0, dtype=torch.float)
    rec = ctp / n_targets
    prec = ctp / (ctp + cfp + torch.finfo(torch.float64).eps)
    prec = prec.flip(0).cummax(0).values.flip(0)  # precision envelope: non-increasing in rank
    # precision / recall at the operating point: last prediction whose score is >= score_threshold
    k = int(torch.searchsorted(-scores, torch.tensor(-float(score_threshold)), right=True))
    recall_at, precision_at = (rec[k - 1], prec[k - 1]) if k > 0 else (zeros.clone(), zeros.clone())
    # F1 over the score grid -> best score threshold
    idx = torch.searchsorted(-scores, -score_grid, right=True)
    rec_p = torch.cat((torch.zeros(1, nthr), rec), 0)[idx]
    prec_p = torch.cat((torch.zeros(1, nthr), prec), 0)[idx]
    f1_curve = (2 * rec_p * prec_p / (rec_p + prec_p + 1e-16)).mean(1)
    best = score_grid[torch.argmax(f1_curve)]
    # AP: precision sampled at the recall thresholds (first rank whose recall reaches the threshold; 0 beyond the last)
    ridx = torch.searchsorted(rec.T.contiguous(), recall_thresholds.view(1, -1).repeat(nthr, 1), right=False).T
    ap = torch.gather(torch.cat((prec, torch.zeros(1, nthr)), 0), 0, ridx).mean(0)
    return ap, precision_at, recall_at, f1_curve, best


def compute_detection_metrics(preds_matched, preds_to_ignore, preds_scores, preds_cls, targets_cls, device="cpu", recall_thresholds=None,
                              score_threshold: Optional[float] = 0.1, calc_best_score_thresholds=None):
    """-> ap, precision, recall, f1 [n_present_classes, nthr], present classes, best score threshold, best per class
    (detection_utils.py:1361-1446).  Runs on the host: once per epoch, a few MB of flags."""
    preds_matched, preds_to_ignore = preds_matched.cpu().bool(), preds_to_ignore.cpu().bool()
    preds_scores, preds_cls, targets_cls = preds_scores.cpu().float(), preds_cls.cpu(), targets_cls.cpu()
    recall_thresholds = torch.linspace(0, 1, 101) if recall_thresholds is None else recall_thresholds.cpu().float()
    classes = torch.unique(targets_cls).long()
    nthr = preds_matched.shape[-1]
    grid = torch.linspace(0, 1, len(recall_thresholds))
    ap = torch.zeros(len(classes), nthr)
    precision, recall = torch.zeros_like(ap), torch.zeros_like(ap)
    f1_curves = torch.zeros(len(classes), len(grid))
    best_per_cls = torch.zeros(len(classes))
    fps_all = ~preds_matched & ~preds_to_ignore
    for i, c in enumerate(classes):
        sel = preds_cls == c
        ap[i], precision[i], recall[i], f1_curves[i], best_per_cls[i] = _metrics_one_class(
            preds_matched[sel], fps_all[sel], preds_scores[sel], int((targets_cls == c).sum()), recall_thresholds, score_threshold, grid)
    f1 = 2 * precision * recall / (precision + recall + 1e-16)
    best = grid[torch.argmax(f1_curves.mean(0))] if len(classes) else torch.tensor(0.0)
    return ap, precision, recall, f1, classes, best, best_per_cls


@register_metric("DetectionMetrics")
class DetectionMetrics:
    """Same constructor, update(preds, target, device, inputs, crowd_targets) / compute() / reset() protocol and metric names as the reference."""

    def __init__(self, num_cls: int, post_prediction_callback=None, normalize_targets: bool = False,
                 iou_thres: Union[IouThreshold, Tuple[float, float], float] = IouThreshold.MAP_05_TO_095, recall_thres: torch.Tensor = None,
                 score_thres: Optional[float] = 0.1, top_k_predictions: int = 100, dist_sync_on_step: bool = False, accumulate_on_cpu: bool = True,
                 calc_best_score_thresholds: bool = True, include_classwise_ap: bool = False, class_names: List[str] = None, state_dict_prefix: str = ""):
        if class_names is None:
            class_names = ["class_" + str(i) for i in range(num_cls)] if include_classwise_ap else None
        elif len(class_names) != num_cls:
            raise ValueError(f"Number of class names ({len(class_names)}) does not match number of classes ({num_cls})")
        self.num_cls, self.iou_thres, self.class_names = num_cls, iou_thres, (list(class_names) if class_names is not None else None)
        if isinstance(iou_thres, IouThreshold):
            self.iou_thresholds = iou_thres.to_tensor()
        elif isinstance(iou_thres, tuple):
            self.iou_thresholds = IouThreshold.from_bounds(*iou_thres)
        else:
            self.iou_thresholds = torch.tensor([iou_thres])
        r = self._get_range_str()
        self.map_str = "mAP" + r
        self.include_classwise_ap = include_classwise_ap
        self.precision_metric_key, self.recall_metric_key = f"{state_dict_prefix}Precision{r}", f"{state_dict_prefix}Recall{r}"
        self.f1_metric_key, self.map_metric_key = f"{state_dict_prefix}F1{r}", f"{state_dict_prefix}mAP{r}"
        gib = [(self.precision_metric_key, True), (self.recall_metric_key, True), (self.map_metric_key, True), (self.f1_metric_key, True)]
        if include_classwise_ap:
            self.per_class_ap_names = [f"{state_dict_prefix}AP{r}_{n}" for n in self.class_names]
            gib += [(k, True) for k in self.per_class_ap_names]
        self.greater_component_is_better = collections.OrderedDict(gib)
        self.component_names = list(self.greater_component_is_better.keys())
        self.calc_best_score_thresholds = calc_best_score_thresholds
        self.best_threshold_per_class_names = [f"Best_score_threshold_{n}" for n in (self.class_names or [])]
        if calc_best_score_thresholds:
            self.component_names.append("Best_score_threshold")
        if calc_best_score_thresholds and include_classwise_ap:
            self.component_names += self.best_threshold_per_class_names
        self.components = len(self.component_names)
        self.post_prediction_callback = post_prediction_callback
        self.denormalize_targets = not normalize_targets
        self.recall_thresholds = torch.linspace(0, 1, 101) if recall_thres is None else torch.tensor(recall_thres, dtype=torch.float32)
        self.score_threshold, self.top_k_predictions, self.accumulate_on_cpu = score_thres, top_k_predictions, accumulate_on_cpu
        self._state = []

    def _get_range_str(self):
        t = self.iou_thresholds
        return "@%.2f" % t[0] if len(t) == 1 else "@%.2f:%.2f" % (t[0], t[-1])

    def reset(self):
        self._state = []

    def to(self, device):
        return self

    @torch.no_grad()
    def update(self, preds, target: torch.Tensor, device: str = None, inputs: torch.Tensor = None, crowd_targets: Optional[torch.Tensor] = None) -> None:
        """preds: the model output (run through post_prediction_callback: device-resident NMS rows, no host sync) or an already
        post-processed list of [Ni,6] tensors; target [T,6] = (img, class, cx, cy, w, h); inputs: the image batch (for H, W)."""
        _, _, height, width = inputs.shape
        if self.post_prediction_callback is not None and hasattr(self.post_prediction_callback, "forward_batched"):
            rows, counts, _ = self.post_prediction_callback.forward_batched(preds)
        else:
            lst = self.post_prediction_callback(preds) if self.post_prediction_callback is not None else preds
            dev = target.device if target.is_cuda else (lst[0].device if len(lst) else target.device)
            pmax = max([int(p.shape[0]) for p in lst if p is not None] + [1])
            rows = torch.zeros(len(lst), pmax, 6, device=dev)
            counts = torch.zeros(len(lst), dtype=torch.int32, device=dev)
            for i, p in enumerate(lst):
                if p is not None and len(p):
                    rows[i, : p.shape[0]] = p.to(dev)
                    counts[i] = p.shape[0]
        crowd = torch.zeros(0, 6, device=rows.device) if crowd_targets is None else crowd_targets
        matched, ignore = K.detection_match(rows, counts, target.to(rows.device), crowd, self.iou_thresholds, height, width, self.top_k_predictions,
                                            self.denormalize_targets)
        keep = lambda t: t.cpu() if self.accumulate_on_cpu else t  # noqa: E731
        self._state.append((keep(matched), keep(ignore), keep(rows[..., 4]), keep(rows[..., 5]), keep(counts), keep(target[:, 1].detach().float())))

    def _flat_state(self):
        m, g, s, c, t = [], [], [], [], []
        for matched, ignore, scores, cls, counts, tcls in self._state:
            P = matched.shape[1]
            valid = (torch.arange(P, device=counts.device).view(1, -1) < counts.view(-1, 1).long())
            m.append(matched[valid].cpu())
            g.append(ignore[valid].cpu())
            s.append(scores[valid].cpu())
            c.append(cls[valid].cpu())
            t.append(tcls.cpu())
        return torch.cat(m), torch.cat(g), torch.cat(s), torch.cat(c), torch.cat(t)

    def compute(self) -> Dict[str, Union[float, torch.Tensor]]:
        mean_ap = mean_precision = mean_recall = mean_f1 = best = -1.0
        ap_cls, thr_cls = [0.0] * self.num_cls, [0.0] * self.num_cls
        state = self._state
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            gathered = [None] * torch.distributed.get_world_size()  # ragged state: object gather, as the reference does (:296-318)
            torch.distributed.all_gather_object(gathered, [tuple(t.cpu() for t in s) for s in state])
            state = [s for part in gathered for s in part]
        if len(state):
            keep, self._state = self._state, state
            flat = self._flat_state()
            self._state = keep
            ap, precision, recall, f1, classes, best, best_cls = compute_detection_metrics(*flat, recall_thresholds=self.recall_thresholds,
                                                                                           score_threshold=self.score_threshold)
            mean_precision, mean_recall, mean_f1, mean_ap = precision.mean(), recall.mean(), f1.mean(), ap.mean()
            for i, c in enumerate(classes):
                if 0 <= int(c) < self.num_cls:
                    ap_cls[int(c)], thr_cls[int(c)] = float(ap[i].mean()), float(best_cls[i])
        out = {self.precision_metric_key: float(mean_precision), self.recall_metric_key: float(mean_recall), self.map_metric_key: float(mean_ap),
               self.f1_metric_key: float(mean_f1)}
        if self.include_classwise_ap:
            out.update(zip(self.per_class_ap_names, ap_cls))
        if self.calc_best_score_thresholds:
            out["Best_score_threshold"] = float(best)
        if self.include_classwise_ap and self.calc_best_score_thresholds:
            out.update(zip(self.best_threshold_per_class_names, thr_cls))
        return out


def _variant(name, iou):
    def init(self, num_cls, post_prediction_callback=None, normalize_targets=False, recall_thres=None, score_thres=0.1, top_k_predictions=100,
             dist_sync_on_step=False, accumulate_on_cpu=True, calc_best_score_thresholds=True, include_classwise_ap=False, class_names=None):
        DetectionMetrics.__init__(self, num_cls, post_prediction_callback, normalize_targets, iou, recall_thres, score_thres, top_k_predictions,
                                  dist_sync_on_step, accumulate_on_cpu, calc_best_score_thresholds, include_classwise_ap, class_names)

    return register_metric(name)(type(name, (DetectionMetrics,), {"__init__": init}))


DetectionMetrics_050 = _variant("DetectionMetrics_050", IouThreshold.MAP_05)
DetectionMetrics_075 = _variant("DetectionMetrics_075", 0.75)
DetectionMetrics_050_095 = _variant("DetectionMetrics_050_095", IouThreshold.MAP_05_TO_095)
